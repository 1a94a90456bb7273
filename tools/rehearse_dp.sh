#!/bin/bash
# Two-rank rehearsals on ONE GPU (gloo, every rank on cuda:0): the multi-GPU bench path and data-parallel LoRA
# fine-tuning through the real entry points. The 8-GPU RCCL runs are the driver's.
set -e
export MASTER_ADDR=127.0.0.1
OUT=gpurun_out/rehearse
rm -rf $OUT && mkdir -p $OUT
echo "[rehearse] bench.py, 2 ranks"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 3 --warmup 1 --layers 4 --dist-backend gloo --share-gpu --no-cpu-baseline > $OUT/bench2.log 2>&1
tail -1 $OUT/bench2.log | cut -c1-300
echo "[rehearse] train_retriever (1 rank) -> retrieved.pkl"
timeout -k 10 300 python train_retriever.py --dataset_code synthetic --synthetic --export_root $OUT/lru \
  --max_train_iterations 30 --val_iterations 10 > $OUT/retr.log 2>&1
echo "[rehearse] train_ranker, 2 ranks, LoRA fine-tuning + test"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
  train_ranker.py --dataset_code synthetic --synthetic --llm_retrieved_path $OUT/lru --export_root $OUT/llm \
  --lora_max_steps 6 --lora_val_iterations 3 --warmup_steps 2 --lora_micro_batch_size 4 --train_batch_size 8 \
  --lora_max_val_samples 16 --llm_max_history 5 --dist_backend gloo --share_gpu > $OUT/rank2.log 2>&1
grep -E "LoRA fine-tuning|Ranking Performance|eval " $OUT/rank2.log | cut -c1-200
ls $OUT/llm
