#!/bin/bash
# tests of the LoRA training step, then the 7B-shape bench with and without the side-stream overlap
timeout -k 10 400 python -m pytest tests/test_gpu_llama_train.py -m gpu -q -x > gpurun_out/lt_tests.log 2>&1
tail -3 gpurun_out/lt_tests.log
for c in 1 0; do
  echo "LR_LORA_OVERLAP=$c"
  LR_LORA_OVERLAP=$c timeout -k 10 300 python tools/bench_rank_train.py --layers 32 2>&1 | tail -1
done
