#!/bin/bash
# tile-raster sweep (LR_GEMM_GROUP_M) of the 256-tile GEMM: bash tools/gpu_gemm_groupm.sh <tag> [M, default 32768]
OUT=gpurun_out/${1:-groupm}
M=${2:-32768}
mkdir -p $OUT
for g in 8 4 16 2 32 8; do
  LR_GEMM_GROUP_M=$g timeout -k 10 200 python tools/bench_gemm.py 4 $M > $OUT/g$g.log 2>&1 || exit 1
  echo "group_m=$g"; cat $OUT/g$g.log | grep -v amdgpu.ids
done
