"""Experiment: the bench's timed loop with ONE step in flight (as bench.py) against TWO (alternate steps on two HIP
streams, each with its own workspaces; weights shared) -- does a second stream fill the ragged ends of the GEMM launches
and the attention kernel's idle slots? usage: bench_two_streams.py [--steps 12] [--layers 32] [--token-budget 32768]"""
import argparse, copy, ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import bench as BN
from llamarec_amd.llm import LLAMA2_7B, LlamaRanker
from llamarec_amd.lru import LRURec, init_lru_state_dict
from llamarec_amd.pipeline import TwoStagePipeline
from llamarec_amd.synth import WORKLOADS

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--token-budget", type=int, default=32768)
a = ap.parse_args()
dev = torch.device("cuda:0")
w = WORKLOADS["beauty"]
steps, hist, labels, T = BN.build_steps("beauty", 0, a.steps, a.token_budget, 0, dev, True)
retr = LRURec.from_state_dict(init_lru_state_dict(w["V"], seed=42), device=dev)
rank = LlamaRanker.random_init(dict(LLAMA2_7B, num_hidden_layers=a.layers), seed=42, device=dev)
label_ids = list(range(319, 339))


def view(obj):          # same handle and weights, own workspace; never destroys the handle
    v = copy.copy(obj)
    v._ws = None
    return v


def run(n_streams):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    views = [(view(retr), view(rank)) for _ in range(n_streams)]
    pipes = [TwoStagePipeline(r, k, label_ids, device=dev, shared_prefix=True) for r, k in views]
    def go(i):
        s = steps[i % len(steps)]
        with torch.cuda.stream(streams[i % n_streams]):
            pipes[i % n_streams].step(s["hist"], s["labels"], s["ids"], s["cu_dev"], s["cu"], s["prefix"])
    for i in range(2 * n_streams): go(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps): go(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    users = sum(steps[i % len(steps)]["users"] for i in range(a.steps))
    hists = sum(p.hist_rerank for p in pipes).cpu().numpy()
    for r, k in views:   # the originals own the handles
        r._h = ctypes.c_void_p(None); k._h = ctypes.c_void_p(None)
    return users / dt, dt / a.steps * 1e3, hists

for rep in range(2):
    for ns in (1, 2):
        ups, ms, h = run(ns)
        print(f"{ns} step(s) in flight: {ups:.2f} users/s, {ms:.2f} ms/step, rerank histogram checksum {int((h * np.arange(len(h))).sum())}", flush=True)
