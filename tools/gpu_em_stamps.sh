#!/bin/bash
# s_memtime anatomy of em_layer_kernel from an EXPERIMENTS build. The build goes to a SEPARATE library selected with
# LLAMAREC_LIB (never install an experiment build as the product library).  usage: gpu_em_stamps.sh [workload users]...
L=$(pwd)/llamarec_amd/lib; C=$(pwd)/llamarec_amd/csrc
mkdir -p $L/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DLR_EXPERIMENTS -c $C/lru_encoder_mfma.hip -o $L/exp/enc.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/exp/lib_exp_enc.so $(ls $L/obj/*.o | grep -v lru_encoder_mfma.o) $L/exp/enc.o || exit 1
export LLAMAREC_LIB=$L/exp/lib_exp_enc.so
python tools/em_stamps.py ${1:-beauty} ${2:-22332} 2>&1 | grep -v amdgpu.ids | tail -8
[ -n "$3" ] && python tools/em_stamps.py $3 $4 2>&1 | grep -v amdgpu.ids | tail -8
