"""Wall time of the retriever's training step with and without deterministic mode, as bench.py drives it (train_step = loss_and_grads +
apply, interleaved) and as tools/bench_train.py does (the two halves in separate loops)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from llamarec_amd.lru import init_lru_state_dict
from llamarec_amd.train import LRUTrainEngine
from llamarec_amd.synth import WORKLOADS
wb = WORKLOADS["beauty"]
rng = np.random.default_rng(0)
Bt, Lt = 64, wb["L"]
seq = rng.integers(1, wb["V"] + 1, size=(Bt, Lt + 1))
toks_h, labs_h = seq[:, :-1].copy(), seq[:, 1:].copy()
if len(sys.argv) > 1 and sys.argv[1] == "padded":   # bench.py's batch: half of the rows left-padded like short users
    for i, n in enumerate(rng.integers(2, Lt, size=Bt // 2)):
        toks_h[i, : Lt - n] = 0
        labs_h[i, : Lt - n - 1] = 0
toks, labs = torch.from_numpy(toks_h).cuda(), torch.from_numpy(labs_h).cuda()
for det in ((1, 1, 1) if (len(sys.argv) > 2 and sys.argv[2] == "detdet") else (0, 1, 0, 1)):
    e = LRUTrainEngine(init_lru_state_dict(wb["V"], seed=1), seed=3, use_graph=True)
    if det:
        e.set_deterministic(True)
    for _ in range(3):
        e.train_step(toks, labs)
    torch.cuda.synchronize()
    def timed(fn, n=50):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    a = timed(lambda: e.train_step(toks, labs))
    b = timed(lambda: e.loss_and_grads(toks, labs))
    c = timed(lambda: e.apply())
    print(f"deterministic={det}: train_step {a:.3f} ms; loss_and_grads alone {b:.3f} ms; apply alone {c:.3f} ms", flush=True)
    del e
