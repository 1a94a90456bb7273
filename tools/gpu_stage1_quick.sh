#!/bin/bash
# stage-1 quick loop: bit-exactness tests, then Beauty retrieve time and the item kernels' share for chunk counts
OUT=gpurun_out/${1:-s1q}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_lru.py tests/test_gpu_edge_cases.py -m gpu -q -x > $OUT/tests.log 2>&1; rc=$?; tail -2 $OUT/tests.log; [ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping"; exit 1; }
for c in 0 1 4; do
  cd /tmp && export TMPDIR=/tmp && LR_TOPK_CHUNKS=$c rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt$c -- python3 $R/tools/bench_stage1.py beauty > $R/$OUT/kt$c.log 2>&1
  cd $R; echo "=== chunks=$c (0 = geometry model)"; grep "beauty" $OUT/kt$c.log | grep -v simple_timer; python tools/kstats.py $(find $OUT/kt$c -name '*kernel_stats.csv' | head -1) 12 12 2>/dev/null | grep "item_\|bound_\|cand_\|merge\|sort\|total"
done
