#!/bin/bash
# kernel stats of the ranker's LoRA step (run on the GPU box): rocprofv3 --kernel-trace --stats of tools/bench_rank_train.py
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/kt_rank; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/bench_rank_train.py --layers 32 --steps 3 $RT_ARGS > $OUT/run.log 2>&1 || exit 1
cd $R; python3 tools/kstats.py "$(find $OUT/kt -name '*kernel_stats.csv' | head -1)" 26 4 | tee $OUT/summary.txt; grep "^layers=" $OUT/run.log
