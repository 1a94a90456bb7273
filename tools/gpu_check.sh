#!/bin/bash
# One gpurun call: GPU test suite, then (only if pytest ended normally: rc 0 = green, 1 = some test failed) bench runs.
# Everything is written under gpurun_out/<tag>/.   usage: bash tools/gpu_check.sh <tag> [quick]
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
rc=$?
tail -15 $OUT/tests.log
echo "pytest rc=$rc"
[ $rc -le 1 ] || exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench default failed"; tail -5 $OUT/bench_default.err; exit 1; }
tail -c 300 $OUT/bench_default.json
[ "$2" = "quick" ] && exit 0
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-shared-prefix --no-cpu-baseline --no-other-shapes > $OUT/bench_noprefix.json 2> $OUT/bench_noprefix.err || exit 1
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-shared-prefix --users-per-step 16 --no-cpu-baseline --no-other-shapes > $OUT/bench_ref_batches.json 2> $OUT/bench_ref_batches.err || exit 1
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --token-budget 32768 --no-cpu-baseline --no-other-shapes > $OUT/bench_32k.json 2> $OUT/bench_32k.err || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 2 --token-budget 8192 --no-cpu-baseline --no-other-shapes > $OUT/bench_8k.json 2> $OUT/bench_8k.err || exit 1
echo done
