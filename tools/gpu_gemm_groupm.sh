#!/bin/bash
OUT=gpurun_out/${1:-groupm}
mkdir -p $OUT
for g in 8 4 16 2 32 8; do
  LR_GEMM_GROUP_M=$g timeout -k 10 200 python tools/bench_gemm.py 4 16384 > $OUT/g$g.log 2>&1 || exit 1
  echo "group_m=$g"; cat $OUT/g$g.log | grep -v amdgpu.ids
done
