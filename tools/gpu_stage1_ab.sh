#!/bin/bash
# stage 1 (retrieve): tests with the current build, then tools/bench_stage1.py with the previous build
# (llamarec_amd/lib/libllamarec_old.so) and the current one, alternating: bash tools/gpu_stage1_ab.sh <tag>
set -o pipefail
OUT=gpurun_out/${1:-stage1_ab}; mkdir -p $OUT
L=$(pwd)/llamarec_amd/lib
timeout -k 10 800 python -m pytest tests/test_gpu_lru.py tests/test_gpu_edge_cases.py tests/test_gpu_entrypoints.py -m gpu -q -x > $OUT/tests.log 2>&1
rc=$?; tail -3 $OUT/tests.log; [ $rc -eq 0 ] || { echo "pytest rc=$rc"; exit 1; }
for i in 1 2; do for which in old new; do
  lib=$L/libllamarec_mi355x.so; [ $which = old ] && lib=$L/libllamarec_old.so
  echo "== $which $i"; LLAMAREC_LIB=$lib timeout -k 10 300 python tools/bench_stage1.py 2>&1 | grep "users/s" || exit 1
done; done
