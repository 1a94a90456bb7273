#!/bin/bash
# Smoke every bench / tool entry once (short runs) -- used before a round ends to catch a tool broken by a refactor.
set -e
for w in beauty games synth-1m; do
  echo "[bench.py --workload $w]"
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-other-shapes 2>&1 | tail -1 | cut -c1-160
done
echo "[bench_online]";  timeout -k 10 200 python tools/bench_online.py 2>&1 | tail -2
echo "[bench_stage1]";  timeout -k 10 200 python tools/bench_stage1.py 2>&1 | tail -4
echo "[bench_train]";   timeout -k 10 200 python tools/bench_train.py --graph 0 2>&1 | tail -4
echo "[bench_attn]";    timeout -k 10 200 python tools/bench_attn.py 2>&1 | tail -3
