#!/bin/bash
# stage-1 iteration loop on the GPU box: bit-exactness tests of the retriever, then its throughput with and without the
# bound pre-pass and over chunk counts, and the per-kernel breakdown
TAG=${1:-s1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_lru.py tests/test_gpu_edge_cases.py tests/test_gpu_entrypoints.py -m gpu -q -x > $OUT/tests.log 2>&1
rc=$?
tail -6 $OUT/tests.log
echo "pytest rc=$rc"
[ $rc -le 1 ] || exit $rc
echo "--- bound pre-pass ON (default)"; timeout -k 10 300 python tools/bench_stage1.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_on.log
echo "--- bound pre-pass OFF"; LR_TOPK_BOUND=0 timeout -k 10 300 python tools/bench_stage1.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_off.log
for c in 1 2 3 4 6 8; do echo "--- ON, chunks=$c"; LR_TOPK_CHUNKS=$c timeout -k 10 300 python tools/bench_stage1.py beauty 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench_chunks.log; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -- python3 $GRAFT_REPO_ROOT/tools/bench_stage1.py beauty > $GRAFT_REPO_ROOT/$OUT/kt.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/kstats.py $(find $OUT/kt -name '*kernel_stats.csv' | head -1) 14 1 2>/dev/null | tee $OUT/kstats.txt
