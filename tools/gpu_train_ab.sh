#!/bin/bash
# retriever training step: tests with the current build, then tools/bench_train.py with the previous build
# (llamarec_amd/lib/libllamarec_old.so) and the current one, alternating: bash tools/gpu_train_ab.sh <tag>
set -o pipefail
OUT=gpurun_out/${1:-train_ab}; mkdir -p $OUT
L=$(pwd)/llamarec_amd/lib
timeout -k 10 600 python -m pytest tests/test_gpu_lru_train.py -m gpu -q -x > $OUT/tests.log 2>&1
rc=$?; tail -3 $OUT/tests.log; [ $rc -eq 0 ] || { echo "pytest rc=$rc"; exit 1; }
for i in 1 2; do for which in old new; do
  lib=$L/libllamarec_mi355x.so; [ $which = old ] && lib=$L/libllamarec_old.so
  echo "== $which $i"; LLAMAREC_LIB=$lib timeout -k 10 300 python tools/bench_train.py 2>&1 | grep "fwd+bwd" || exit 1
done; done
