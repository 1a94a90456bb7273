"""Does the retriever's training step run at the same speed whatever torch stream the engine gets? (bench.py saw its third training
engine's captured graph take 2-3 x the GPU time of the previous one's: docs/EXPERIMENTS.md.) K dummy streams are created first."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from llamarec_amd.lru import init_lru_state_dict
from llamarec_amd.train import LRUTrainEngine
from llamarec_amd.synth import WORKLOADS
wb = WORKLOADS["beauty"]
rng = np.random.default_rng(0)
seq = rng.integers(1, wb["V"] + 1, size=(64, wb["L"] + 1))
toks, labs = torch.from_numpy(seq[:, :-1].copy()).cuda(), torch.from_numpy(seq[:, 1:].copy()).cuda()
keep = []
if len(sys.argv) > 1 and sys.argv[1] in ("lowprio", "lowprio_alive", "normal_destroyed"):   # what the LoRA engine does: a lowest-priority non-blocking stream, later destroyed
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    lo, hi = ctypes.c_int(), ctypes.c_int()
    assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
    side = ctypes.c_void_p()
    prio = 0 if sys.argv[1] == "normal_destroyed" else lo.value
    assert hip.hipStreamCreateWithPriority(ctypes.byref(side), 1, prio) == 0
    x = torch.ones(1 << 20, device="cuda")
    torch.cuda.synchronize()
    print(f"priority range least {lo.value} greatest {hi.value}; created a stream of priority {prio} ({sys.argv[1]})", flush=True)
    if sys.argv[1] != "lowprio_alive":
        assert hip.hipStreamDestroy(side) == 0
for k in range(12):
    e = LRUTrainEngine(init_lru_state_dict(wb["V"], seed=1), seed=3, use_graph=True)
    for _ in range(3):
        e.train_step(toks, labs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        e.train_step(toks, labs)
    torch.cuda.synchronize()
    print(f"engine {k}: stream {e._stream.cuda_stream:#x}: train_step {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms", flush=True)
    del e
    keep.append(torch.cuda.Stream())   # one more stream taken from the pool between engines
