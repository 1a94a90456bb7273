#!/bin/bash
# same-box A/B of two library builds on stage 1 (llamarec_amd/lib/libllamarec_old.so against the current one, selected with
# LLAMAREC_LIB; the product library is never overwritten): bit-exactness tests on the current build, then alternating
# tools/bench_stage1.py runs.   usage: bash tools/gpu_ab_stage1.sh <tag> [workloads...]
OUT=gpurun_out/${1:-abs1}; shift; mkdir -p $OUT
L=$(pwd)/llamarec_amd/lib
timeout -k 10 600 python -m pytest tests/test_gpu_lru.py tests/test_gpu_edge_cases.py -m gpu -q -x > $OUT/tests.log 2>&1; rc=$?
tail -2 $OUT/tests.log; [ $rc -eq 0 ] || exit 1
for i in 1 2 3; do for which in old new; do
  lib=$L/libllamarec_mi355x.so; [ $which = old ] && lib=$L/libllamarec_old.so
  echo "== $which $i"; LLAMAREC_LIB=$lib timeout -k 10 200 python tools/bench_stage1.py ${@:-synth-1m beauty} 2>&1 | grep -v amdgpu | cut -c1-120
done; done
