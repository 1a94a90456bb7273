#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}   # set before any cd: a missing variable must not turn into /gpurun_out and /tools paths
# round 4, stage 1: bit-exactness tests of the retriever, its throughput at the BASELINE shapes, per-kernel breakdown at
# Synth-1M (4 096 users) and Beauty.   usage: bash tools/gpu_r4_stage1.sh <tag>
TAG=${1:-r4s1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_lru.py tests/test_gpu_edge_cases.py -m gpu -q -x > $OUT/tests.log 2>&1
rc=$?
tail -8 $OUT/tests.log
echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_stage1.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bench.log
cd /tmp && export TMPDIR=/tmp
for w in synth-1m beauty; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_$w -- python3 $R/tools/bench_stage1.py $w > $R/$OUT/kt_$w.log 2>&1 || exit 1
  f=$(find $R/$OUT/kt_$w -name '*kernel_stats.csv' | head -1)
  cp $f $R/$OUT/stage1_${w}_kernel_stats.csv
  head -14 $f | cut -c1-150
done
