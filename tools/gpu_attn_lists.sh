#!/bin/bash
# attention variant 4 (item lists) against variant 2 (one tile per workgroup): bit-identity tests, then the stand-alone timings
# (uniform batches + the ragged Beauty token-budget step) with the lists in couples (default) and as single tiles.
set -e
mkdir -p gpurun_out/attn_lists
timeout -k 10 600 python -m pytest tests/test_gpu_llama.py -x -q -m gpu -k "attention or shared_prefix" > gpurun_out/attn_lists/tests.log 2>&1
tail -3 gpurun_out/attn_lists/tests.log
timeout -k 10 300 python tools/bench_attn.py 2,4,2,4 > gpurun_out/attn_lists/bench_attn.log 2>&1
cat gpurun_out/attn_lists/bench_attn.log
echo "== LR_ATTN_LIST_MODE=1 (single tiles)"
LR_ATTN_LIST_MODE=1 timeout -k 10 300 python tools/bench_attn.py 4,4 > gpurun_out/attn_lists/bench_attn_mode1.log 2>&1
cat gpurun_out/attn_lists/bench_attn_mode1.log
