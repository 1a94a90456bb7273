#!/bin/bash
# s_memtime anatomy of the 256x256 GEMM from an EXPERIMENTS build kept in a SEPARATE library (LLAMAREC_LIB).
# usage: gpu_gemm_stamps.sh "M shape" ["M shape" ...]     e.g. "32768 o" "1024 o"
L=$(pwd)/llamarec_amd/lib; C=$(pwd)/llamarec_amd/csrc
mkdir -p $L/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DLR_EXPERIMENTS $GEMM_DEFS -c $C/llama_gemm.hip -o $L/exp/gemm.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/exp/lib_exp_gemm.so $(ls $L/obj/*.o | grep -v llama_gemm.o) $L/exp/gemm.o || exit 1
export LLAMAREC_LIB=$L/exp/lib_exp_gemm.so
for a in "$@"; do python tools/gemm_stamps.py $a 2>&1 | grep -v amdgpu.ids || exit 1; done
