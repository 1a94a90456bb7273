"""Latency of the online single-user path (reference demo/inference.py:46-76):
retrieve top-20 for one history (no mask) -> one prompt -> Llama-2-7b prefill -> verbalizer -> top-10.
Synthetic ids and random bf16 weights at Llama-2-7b shapes; prints per-stage and total latency.
usage: bench_online.py [--tokens 460] [--iters 20] [--layers 32] [--batch 1]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from llamarec_amd import metrics as M  # noqa: E402
from llamarec_amd.llm import LlamaRanker  # noqa: E402
from llamarec_amd.lru import LRURec, init_lru_state_dict  # noqa: E402
from llamarec_amd.synth import WORKLOADS  # noqa: E402

LLAMA2_7B = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=4096,
                 rms_norm_eps=1e-5, rope_theta=10000.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=460)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--gemm-variant", type=int, default=5, help="5 = latency mode (split-K), 0 = throughput default")
    a = ap.parse_args()
    w = WORKLOADS["ml-100k"]
    rng = np.random.default_rng(0)
    retr = LRURec.from_state_dict(init_lru_state_dict(w["V"], seed=42))
    cfg = dict(LLAMA2_7B, num_hidden_layers=a.layers)
    rank = LlamaRanker.random_init(cfg, seed=42).set_variants(a.gemm_variant, 0)
    labels = np.arange(319, 339, dtype=np.int32)  # 20 distinct token ids
    hist = rng.integers(1, w["V"] + 1, size=(a.batch, 30)).astype(np.int64)
    prompts = [np.concatenate([[1], rng.integers(3, 32000, size=a.tokens - 1)]).astype(np.int32) for _ in range(a.batch)]

    def once():
        t0 = time.perf_counter()
        idx, _ = retr.retrieve_topk(hist, 20, exclude_history=False)
        cands = idx.cpu().numpy()
        t1 = time.perf_counter()
        scores = rank.prefill_verbalize(prompts, labels)
        order = M.rank_classes(scores).cpu().numpy()
        t2 = time.perf_counter()
        return (t1 - t0) * 1e3, (t2 - t1) * 1e3, cands, order

    for _ in range(3):
        once()
    r, p = [], []
    for _ in range(a.iters):
        x, y, _, _ = once()
        r.append(x)
        p.append(y)
    flops = a.batch * (a.tokens * 12.95e9 * a.layers / 32 + a.tokens ** 2 * 2.62e5 * a.layers / 32)
    print(f"online path (gemm variant {a.gemm_variant}), batch {a.batch}, {a.tokens} prompt tokens, {a.layers} layers: retrieve {np.median(r):.3f} ms, "
          f"prefill+verbalize+rank {np.median(p):.3f} ms (min {np.min(p):.3f}), total {np.median(r) + np.median(p):.3f} ms; "
          f"prefill {flops / np.median(p) / 1e9:.1f} TFLOP/s; weight-stream floor {13.5e9 * a.layers / 32 / 8e12 * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
