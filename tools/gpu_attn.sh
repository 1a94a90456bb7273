#!/bin/bash
# attention iteration loop on the GPU box: parity tests that touch the attention kernel, then the microbench
TAG=${1:-attn}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_llama.py tests/test_gpu_llama_train.py tests/test_gpu_edge_cases.py -m gpu -q -x > $OUT/tests.log 2>&1
rc=$?
tail -6 $OUT/tests.log
echo "pytest rc=$rc"
[ $rc -le 1 ] || exit $rc
timeout -k 10 200 python tools/bench_attn.py > $OUT/bench_attn.log 2>&1 || { tail -5 $OUT/bench_attn.log; exit 1; }
cat $OUT/bench_attn.log
