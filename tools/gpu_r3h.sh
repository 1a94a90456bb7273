#!/bin/bash
# round 3: refresh the stage-1 and training profiles after the grouped bound path / training GEMM changes; smoke()
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r03b
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_s1" -- python3 "$ROOT/tools/prof_stage1.py" > "$OUT/stage1_kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_s1_fetch" -- python3 "$ROOT/tools/prof_stage1.py" --only=synth-1m > "$OUT/stage1_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_s1_write" -- python3 "$ROOT/tools/prof_stage1.py" --only=synth-1m > "$OUT/stage1_pmc_write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_s1_sq" -- python3 "$ROOT/tools/prof_stage1.py" --only=synth-1m > "$OUT/stage1_pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_train" -- python3 "$ROOT/tools/bench_train.py" --only beauty --graph 0 > "$OUT/train_kt.log" 2>&1
cd "$ROOT"
f() { find "$1" -name "$2" | head -1; }
cp "$(f $OUT/kt_s1 '*kernel_stats.csv')" $OUT/r03_stage1_kernel_stats.csv
cp "$(f $OUT/kt_train '*kernel_stats.csv')" $OUT/r03_train_beauty_kernel_stats.csv
python3 tools/summarize_pmc.py $OUT/r03_stage1_pmc_summary.json a=$(f $OUT/pmc_s1_fetch '*counter_collection.csv') b=$(f $OUT/pmc_s1_write '*counter_collection.csv') c=$(f $OUT/pmc_s1_sq '*counter_collection.csv') > "$OUT/pmc_s1_summary.txt"
cat $OUT/pmc_s1_summary.txt | cut -c1-250
python3 tools/kstats.py $OUT/r03_stage1_kernel_stats.csv 12 1 | head -16
grep "TFLOP" $OUT/train_kt.log | tail -1
