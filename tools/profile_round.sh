#!/bin/bash
# Round profile refresh, run on the GPU box from the repo root:
#   bash tools/profile_round.sh r02
# 1. rocprofv3 --kernel-trace --stats of the default bench command          -> profiles/<tag>_bench_steps5_kernel_stats.csv
# 2. four separate --pmc passes over a 2-layer slice of the same command (the per-launch GEMM numbers do not depend on
#    depth): FETCH_SIZE; WRITE_SIZE; MFMA busy + clock; L2 hit / miss + the share of L2-side reads that leave for DRAM
#                                                                            -> profiles/<tag>_pmc_summary.json
# 3. the same for the stage-1 roofline point (synth-1M item GEMM + top-K)    -> profiles/<tag>_stage1_*
# 4. kernel stats of the retriever training step and of the ranker LoRA step -> profiles/<tag>_train_*, <tag>_rank_train_*
# Every PMC pass is `rocprofv3 --pmc <counters> --kernel-trace` (the one combination the pool allows: counters are never
# mixed with --sys-trace / --runtime-trace / hip / hsa / memory-copy / marker domains). PMC passes serialise kernels and
# run 2-5 % slower (MI355X_MICROARCH.md, DVFS give-back (2)): durations quoted in DESIGN.md come from pass 1 and from
# bench.py's own HIP events, never from a PMC pass. The program sits directly after `--` (no env / bash hop).
set -e
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$ROOT/profiles"
export TMPDIR=/tmp
cd /tmp

PMC_BENCH="--steps 1 --warmup 1 --layers 2 --no-cpu-baseline --no-profile --no-other-shapes"
echo "[profile] kernel trace of bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-other-shapes > "$OUT/bench_kt.log" 2>&1
echo "[profile] pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $PMC_BENCH > "$OUT/bench_pmc_fetch.log" 2>&1
echo "[profile] pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $PMC_BENCH > "$OUT/bench_pmc_write.log" 2>&1
echo "[profile] pmc MFMA busy / clock"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $PMC_BENCH > "$OUT/bench_pmc_sq.log" 2>&1
echo "[profile] pmc L2 hit/miss, EA read requests and those destined for DRAM (no MALL / Infinity-Cache counter is exposed by rocprofv3 -L on gfx950)"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d "$OUT/pmc_tcc" -- python3 "$ROOT/bench.py" $PMC_BENCH > "$OUT/bench_pmc_tcc.log" 2>&1 || echo "[profile] TCC pass failed (counter set not accepted): see $OUT/bench_pmc_tcc.log"
echo "[profile] stage-1 kernel trace (synth-1M, Beauty, Games, ML-100k)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_s1" -- python3 "$ROOT/tools/prof_stage1.py" > "$OUT/stage1_kt.log" 2>&1
echo "[profile] stage-1 kernel trace, Beauty only (22 332 users in one call)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_s1b" -- python3 "$ROOT/tools/bench_stage1.py" beauty > "$OUT/stage1_beauty_kt.log" 2>&1
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
cp "$(f $OUT/kt_s1b '*kernel_stats.csv')" profiles/${TAG}_stage1_beauty_kernel_stats.csv
cp "$(f $OUT/kt_train '*kernel_stats.csv')" profiles/${TAG}_train_beauty_kernel_stats.csv
cp "$(f $OUT/kt_rank_train '*kernel_stats.csv')" profiles/${TAG}_rank_train_kernel_stats.csv
{ python3 tools/kstats.py profiles/${TAG}_bench_steps5_kernel_stats.csv 16 6; echo "# python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-shapes (6 passes of the step incl. the warm-up)"; grep '^{"metric"' "$OUT/bench_kt.log" | tail -1 | head -c 2500; echo; } > profiles/${TAG}_bench_summary.txt
{ python3 tools/kstats.py profiles/${TAG}_stage1_beauty_kernel_stats.csv 16 6; echo "# python3 tools/bench_stage1.py beauty (1 warm-up + 5 timed calls of 22 332 users)"; grep beauty "$OUT/stage1_beauty_kt.log" | grep -v simple_timer; } > profiles/${TAG}_stage1_beauty_summary.txt
{ python3 tools/kstats.py profiles/${TAG}_rank_train_kernel_stats.csv 24 4; echo "# tools/bench_rank_train.py --layers 32 --steps 3 (1 warm-up + 3 timed passes; lt_transpose_kernel and the at::native initialisers are setup)"; grep "^layers=" "$OUT/rank_train_kt.log"; } > profiles/${TAG}_rank_train_summary.txt
TCC=$(f $OUT/pmc_tcc '*counter_collection.csv')
python3 tools/summarize_pmc.py profiles/${TAG}_pmc_summary.json a=$(f $OUT/pmc_fetch '*counter_collection.csv') b=$(f $OUT/pmc_write '*counter_collection.csv') c=$(f $OUT/pmc_sq '*counter_collection.csv') ${TCC:+d=$TCC} > "$OUT/pmc_summary.txt"
python3 tools/summarize_pmc.py profiles/${TAG}_stage1_pmc_summary.json a=$(f $OUT/pmc_s1_fetch '*counter_collection.csv') b=$(f $OUT/pmc_s1_write '*counter_collection.csv') c=$(f $OUT/pmc_s1_sq '*counter_collection.csv') > "$OUT/pmc_s1_summary.txt"
cp profiles/${TAG}_*.csv profiles/${TAG}_*.json profiles/${TAG}_*.txt "$OUT/" 2>/dev/null || true
tail -2 "$OUT/bench_kt.log" | head -c 600; echo
cat "$OUT/pmc_summary.txt" "$OUT/pmc_s1_summary.txt"
