#!/bin/bash
# LoRA-path tests with the current build, then the ranker training step and the attention microbench with the previous
# build (llamarec_amd/lib/libllamarec_old.so) and the current one, alternating: bash tools/gpu_attn_bwd_ab.sh <tag>
set -o pipefail
OUT=gpurun_out/${1:-attn_bwd}; mkdir -p $OUT
L=$(pwd)/llamarec_amd/lib
timeout -k 10 900 python -m pytest tests/test_gpu_llama_train.py tests/test_gpu_llama.py -m gpu -q -x > $OUT/tests.log 2>&1
rc=$?; tail -3 $OUT/tests.log
[ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping"; exit 1; }
for i in 1 2; do
  for which in old new; do
    lib=$L/libllamarec_mi355x.so; [ $which = old ] && lib=$L/libllamarec_old.so
    echo "== $which $i"
    LLAMAREC_LIB=$lib timeout -k 10 300 python tools/bench_rank_train.py --layers 8 --steps 3 2>&1 | grep "tokens/s" || exit 1
  done
done
