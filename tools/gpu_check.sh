#!/bin/bash
# One gpurun call: GPU test suite, then (only if pytest ended normally: rc 0 = green, 1 = some test failed) the bench
# variants and the GEMM row-count sweep. Everything is written under gpurun_out/<tag>/.
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
rc=$?
tail -5 $OUT/tests.log
echo "pytest rc=$rc"
[ $rc -le 1 ] || exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench default failed"; tail -5 $OUT/bench_default.err; exit 1; }
tail -c 600 $OUT/bench_default.json
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-shared-prefix --no-cpu-baseline --no-other-shapes > $OUT/bench_noprefix.json 2> $OUT/bench_noprefix.err || exit 1
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-shared-prefix --users-per-step 16 --no-cpu-baseline --no-other-shapes > $OUT/bench_ref_batches.json 2> $OUT/bench_ref_batches.err || exit 1
python bench.py --gpus 2 --steps 2 > $OUT/bench_gpus2.out 2> $OUT/bench_gpus2.err; echo "bench --gpus 2 on this box: rc=$?" | tee -a $OUT/bench_gpus2.err
timeout -k 10 300 python tools/bench_gemm.py 4 8192,12288,14800,16384,20480,24576,32768 > $OUT/gemm_sweep.log 2>&1 || exit 1
cat $OUT/gemm_sweep.log
(rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|TCC_EA0|TCC_HIT|TCC_MISS|TCC_REQ" | head -80) > $OUT/counters.txt
echo done
