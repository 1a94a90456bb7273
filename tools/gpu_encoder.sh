#!/bin/bash
# encoder loop: bit-exactness tests (stage 1 + edge cases + entry points), then the Beauty / Games / ML-100k retrieve time
# with the per-kernel split of the Beauty call
OUT=gpurun_out/${1:-enc}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
set -o pipefail
timeout -k 10 420 python -m pytest tests/test_gpu_lru.py tests/test_gpu_edge_cases.py tests/test_gpu_entrypoints.py -m gpu -q -x > $OUT/tests.log 2>&1; rc=$?; tail -3 $OUT/tests.log; [ $rc -eq 0 ] || { echo "pytest rc=$rc: stopping"; exit 1; }
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt -- python3 $R/tools/bench_stage1.py beauty > $R/$OUT/kt.log 2>&1
cd $R; grep "beauty" $OUT/kt.log | grep -v simple_timer; python tools/kstats.py $(find $OUT/kt -name '*kernel_stats.csv' | head -1) 16 6 2>/dev/null
timeout -k 10 300 python tools/bench_stage1.py 2>&1 | grep -v simple_timer | tail -8
