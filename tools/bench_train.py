"""Retriever training-step timing at the reference's batch shapes
(config.py:103-111: batch 64 x L 50 for Beauty/Games, 16 x 200 for ML-100k), synthetic full-length rows.
usage: bench_train.py [--iters 20] [--only beauty]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from llamarec_amd.lru import init_lru_state_dict  # noqa: E402
from llamarec_amd.synth import WORKLOADS  # noqa: E402
from llamarec_amd.train import LRUTrainEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--deterministic", type=int, default=0)   # lr_lru_train_set_deterministic
a = ap.parse_args()
for name, B in (("beauty", 64), ("games", 64), ("ml-100k", 16), ("synth-1m", 64)):
    if a.only and name != a.only:
        continue
    w = WORKLOADS[name]
    V, L = w["V"], w["L"]
    rng = np.random.default_rng(0)
    seq = rng.integers(1, V + 1, size=(B, L + 1))
    tokens, labels = seq[:, :-1].copy(), seq[:, 1:].copy()
    short = rng.integers(2, L, size=B // 2)            # half of the rows left-padded like short users
    for i, n in enumerate(short):
        tokens[i, : L - n] = 0
        labels[i, : L - n - 1] = 0
    eng = LRUTrainEngine(init_lru_state_dict(V, seed=1), seed=3, use_graph=bool(a.graph))
    if a.deterministic:
        eng.set_deterministic(True)
    t = torch.from_numpy(tokens).cuda()
    l = torch.from_numpy(labels).cuda()
    for _ in range(3):
        eng.train_step(t, l)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        eng.loss_and_grads(t, l)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(a.iters):
        eng.apply()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    R = B * L
    flops = 3 * 2.0 * R * (V + 1) * 64 + 3 * R * 2 * 2.64e5 / 2    # three V-sized GEMMs + fwd/bwd of the blocks
    ms_fb, ms_opt = (t1 - t0) / a.iters * 1e3, (t2 - t1) / a.iters * 1e3
    print(f"graph={a.graph} {name:9s} B={B} L={L} V={V}: fwd+bwd {ms_fb:.3f} ms, clip+AdamW {ms_opt:.3f} ms, "
          f"{B / (ms_fb + ms_opt) * 1e3:.0f} sequences/s, {flops / ms_fb / 1e9:.2f} TFLOP/s (f32), loss {float(eng._out[0]):.3f}")
