#!/usr/bin/env python
"""Ranker LoRA training step on Llama-2-7b shapes (random init, synthetic prompts): time per micro-batch
(forward + backward), per optimizer step, tokens/s and model FLOP/s. SURVEY.md 8(f) #4.

  python tools/bench_rank_train.py [--layers 32] [--batch 16] [--tokens 460] [--steps 5] [--accum 1]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--batch", type=int, default=16)       # config.py:90-97: lora_micro_batch_size 16 (8 on beauty)
    ap.add_argument("--tokens", type=int, default=460)     # mean prompt length of the ML-100k workload
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--accum", type=int, default=1)
    ap.add_argument("--dropout", type=float, default=0.05)
    a = ap.parse_args()
    from llamarec_amd.llm import LLAMA2_7B, LlamaRanker
    from llamarec_amd.rank_train import LoraTrainEngine

    cfg = dict(LLAMA2_7B, num_hidden_layers=a.layers)
    t0 = time.time()
    ranker = LlamaRanker.random_init(cfg, seed=1)
    eng = LoraTrainEngine(ranker, dropout=a.dropout)
    # non-zero B so that every kernel of the backward sees real numbers
    init = eng.peft_init(3)
    for k in init:
        if k.endswith("lora_B"):
            init[k] = torch.randn(init[k].shape) * 0.01
    eng.load(init)
    torch.cuda.synchronize()
    print(f"setup {time.time() - t0:.1f} s, HBM in use {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
    rng = np.random.default_rng(0)
    lens = np.clip(rng.normal(a.tokens, a.tokens * 0.15, size=a.batch).astype(int), 16, 1536)
    seqs = [np.concatenate([[1], rng.integers(3, 32000, size=n - 2), [2]]).astype(np.int32) for n in lens]
    labels = []
    for s in seqs:
        l = s.copy()
        l[:-2] = -100
        labels.append(l)
    n = int(sum(lens))

    def step():
        for i in range(a.accum):
            eng.loss_and_grads(seqs, labels, grad_scale=1.0 / a.accum, accumulate=i > 0)
        eng.apply(2e-4, 1.0)

    step()
    torch.cuda.synchronize()
    print(f"workspace {eng._ws.numel() / 2**30:.1f} GiB, loss {float(eng._out[0]):.4f}", flush=True)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    fb = ap_ms = 0.0
    for _ in range(a.steps):
        e0.record()
        for i in range(a.accum):
            eng.loss_and_grads(seqs, labels, grad_scale=1.0 / a.accum, accumulate=i > 0)
        e1.record()
        eng.apply(2e-4, 1.0)
        e2.record()
        torch.cuda.synchronize()
        fb += e0.elapsed_time(e1)
        ap_ms += e1.elapsed_time(e2)
    fb /= a.steps * a.accum
    ap_ms /= a.steps
    d, f, L = cfg["hidden_size"], cfg["intermediate_size"], a.layers
    lin = 2 * (d * 3 * d + d * d + 3 * d * f)                    # forward flops per token and layer
    attn = sum(2 * 2 * d * (T * (T + 1) / 2) for T in lens)      # forward attention flops per layer
    flops = L * (2 * n * lin + 3.5 * attn)                       # forward + data gradients (no weight gradients: frozen)
    print(f"layers={L} B={a.batch} tokens={n}: fwd+bwd {fb:.1f} ms/micro-batch, clip+AdamW {ap_ms:.3f} ms, "
          f"{n / fb * 1e3:.0f} tokens/s, {a.batch / fb * 1e3:.1f} samples/s, {flops / fb / 1e9:.0f} TFLOP/s (bf16)")


if __name__ == "__main__":
    main()
