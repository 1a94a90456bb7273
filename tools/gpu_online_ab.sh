#!/bin/bash
# online single-user path: GEMM / llama tests with the current build, then tools/bench_online.py with the previous build
# (llamarec_amd/lib/libllamarec_old.so) and the current one, at 460 and 1000 tokens
set -o pipefail
OUT=gpurun_out/${1:-online_ab}; mkdir -p $OUT
L=$(pwd)/llamarec_amd/lib
timeout -k 10 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x > $OUT/tests.log 2>&1
rc=$?; tail -3 $OUT/tests.log
[ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping"; exit 1; }
for i in 1 2; do
  echo "== old $i"; LLAMAREC_LIB=$L/libllamarec_old.so timeout -k 10 300 python tools/bench_online.py 2>&1 | grep "online path" || exit 1
  echo "== new $i"; timeout -k 10 300 python tools/bench_online.py 2>&1 | grep "online path" || exit 1

done
timeout -k 10 300 python tools/bench_online.py --tokens 1000 2>&1 | grep "online path"
LLAMAREC_LIB=$L/libllamarec_old.so timeout -k 10 300 python tools/bench_online.py --tokens 1000 2>&1 | grep "online path"
