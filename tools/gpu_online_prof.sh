#!/bin/bash
# Kernel trace of the online single-user path (tools/bench_online.py, latency mode): bash tools/gpu_online_prof.sh <tag>
set -e -o pipefail
TAG=${1:-online}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/bench_online.py > "$OUT/online.log" 2>&1
grep "online path" "$OUT/online.log"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$ROOT/tools/bench_online.py" --iters 5 > "$OUT/online_kt.log" 2>&1
cd "$ROOT"
python3 tools/kstats.py $(find "$OUT/kt" -name "*kernel_stats.csv" | head -1) 8 > "$OUT/online_kstats.txt" 2>&1 || true
head -40 "$OUT/online_kstats.txt"
