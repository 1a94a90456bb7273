"""Timing of the compatibility path `LRURec.forward` / lr_lru_scores_last (materialised [B, V+1] scores)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd.lru import LRURec, init_lru_state_dict
from llamarec_amd.synth import WORKLOADS, synth_users
for name, U in (("ml-100k", 610), ("beauty", 4096), ("games", 4096)):
    w = WORKLOADS[name]
    hist, labels, n, T = synth_users(name, U)
    model = LRURec.from_state_dict(init_lru_state_dict(w["V"], seed=42))
    ids = torch.from_numpy(hist).cuda()
    for excl in (False, True):
        model.scores_last(ids, excl); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): s = model.scores_last(ids, excl)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"{name:8s} U={U} V={w['V']} exclude={excl}: {ms:.3f} ms, {U * (w['V'] + 1) * 4 / ms / 1e6:.1f} GB/s of scores, {2 * 64 * U * (w['V'] + 1) / ms / 1e9:.2f} TF/s")
