#!/bin/bash
# Round profile refresh, run on the GPU box from the repo root:
#   bash tools/profile_round.sh r01
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> profiles/<tag>_bench_kernel_stats.csv
# 2. three separate --pmc passes (FETCH_SIZE; WRITE_SIZE; MFMA busy + clock)        -> profiles/<tag>_pmc_summary.json
# 3. kernel stats of the retriever training step (Beauty shape)          -> profiles/<tag>_train_beauty_kernel_stats.csv
# 4. the same for the stage-1 roofline point (synth-1M item GEMM + top-K) -> profiles/<tag>_stage1_*.csv
# 5. kernel stats of the ranker LoRA training step (Llama-2-7b shapes)    -> profiles/<tag>_rank_train_kernel_stats.csv
# The program sits directly after `--` (no env/bash hop) and --pmc is never combined with a trace domain.
set -e
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$ROOT/profiles"
export TMPDIR=/tmp
cd /tmp

PMC_BENCH="--steps 1 --warmup 1 --layers 2 --no-cpu-baseline --no-profile --no-other-shapes"
echo "[profile] kernel trace of bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-other-shapes > "$OUT/bench_kt.log" 2>&1
echo "[profile] pmc FETCH_SIZE (2-layer slice: the per-launch GEMM numbers do not depend on depth)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $PMC_BENCH > "$OUT/bench_pmc_fetch.log" 2>&1
echo "[profile] pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $PMC_BENCH > "$OUT/bench_pmc_write.log" 2>&1
echo "[profile] pmc MFMA busy / clock"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $PMC_BENCH > "$OUT/bench_pmc_sq.log" 2>&1
echo "[profile] stage-1 kernel trace (synth-1M, Beauty, ML-100k)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_s1" -- python3 "$ROOT/tools/prof_stage1.py" > "$OUT/stage1_kt.log" 2>&1
echo "[profile] stage-1 pmc FETCH_SIZE (synth-1M)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_s1_fetch" -- python3 "$ROOT/tools/prof_stage1.py" --only=synth-1m > "$OUT/stage1_pmc_fetch.log" 2>&1
echo "[profile] stage-1 pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_s1_write" -- python3 "$ROOT/tools/prof_stage1.py" --only=synth-1m > "$OUT/stage1_pmc_write.log" 2>&1
echo "[profile] stage-1 pmc MFMA busy"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_s1_sq" -- python3 "$ROOT/tools/prof_stage1.py" --only=synth-1m > "$OUT/stage1_pmc_sq.log" 2>&1

echo "[profile] retriever training step kernel trace (Beauty shape)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_train" -- python3 "$ROOT/tools/bench_train.py" --only beauty --graph 0 > "$OUT/train_kt.log" 2>&1
echo "[profile] ranker LoRA training step kernel trace (Llama-2-7b shapes, 16 prompts)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_rank_train" -- python3 "$ROOT/tools/bench_rank_train.py" --layers 32 --steps 3 > "$OUT/rank_train_kt.log" 2>&1

cd "$ROOT"
f() { find "$1" -name "$2" | head -1; }
cp "$(f $OUT/kt '*kernel_stats.csv')" profiles/${TAG}_bench_steps5_kernel_stats.csv
cp "$(f $OUT/kt_s1 '*kernel_stats.csv')" profiles/${TAG}_stage1_kernel_stats.csv
cp "$(f $OUT/kt_train '*kernel_stats.csv')" profiles/${TAG}_train_beauty_kernel_stats.csv
cp "$(f $OUT/kt_rank_train '*kernel_stats.csv')" profiles/${TAG}_rank_train_kernel_stats.csv
{ python3 tools/kstats.py profiles/${TAG}_rank_train_kernel_stats.csv 24 4; echo "# tools/bench_rank_train.py --layers 32 --steps 3 (1 warm-up + 3 timed passes; lt_transpose_kernel and the at::native initialisers are setup)"; grep "^layers=" "$OUT/rank_train_kt.log"; } > profiles/${TAG}_rank_train_summary.txt
python3 tools/summarize_pmc.py profiles/${TAG}_pmc_summary.json a=$(f $OUT/pmc_fetch '*counter_collection.csv') b=$(f $OUT/pmc_write '*counter_collection.csv') c=$(f $OUT/pmc_sq '*counter_collection.csv') > "$OUT/pmc_summary.txt"
python3 tools/summarize_pmc.py profiles/${TAG}_stage1_pmc_summary.json a=$(f $OUT/pmc_s1_fetch '*counter_collection.csv') b=$(f $OUT/pmc_s1_write '*counter_collection.csv') c=$(f $OUT/pmc_s1_sq '*counter_collection.csv') > "$OUT/pmc_s1_summary.txt"
cp profiles/${TAG}_*.csv profiles/${TAG}_*.json "$OUT/"
tail -2 "$OUT/bench_kt.log"
cat "$OUT/pmc_summary.txt" "$OUT/pmc_s1_summary.txt"
