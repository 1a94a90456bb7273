#!/bin/bash
# s_memtime anatomy of the training step's score kernels from a -DTS_STAMP build kept in a separate library (LLAMAREC_LIB)
L=$(pwd)/llamarec_amd/lib; C=$(pwd)/llamarec_amd/csrc
mkdir -p $L/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DTS_STAMP -c $C/lru_train_scores.hip -o $L/exp/ts.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/exp/lib_exp_ts.so $(ls $L/obj/*.o | grep -v lru_train_scores.o) $L/exp/ts.o || exit 1
LLAMAREC_LIB=$L/exp/lib_exp_ts.so python tools/bench_train.py --only beauty --graph 0 --iters 2 2>&1 | grep -v amdgpu.ids | tail -${1:-24}
