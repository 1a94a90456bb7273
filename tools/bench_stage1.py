"""Stage-1 (retrieve) throughput at the BASELINE shapes, one call over many users (parity at these shapes:
tests/test_gpu_lru.py::test_baseline_shapes_sample_vs_oracle and ::test_full_size_catalog_properties)."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd.lru import LRURec, init_lru_state_dict
from llamarec_amd.synth import WORKLOADS, synth_users

ALL = (("ml-100k", 610), ("beauty", 22332), ("games", 15264), ("synth-1m", 4096))
for name, U in [x for x in ALL if len(sys.argv) < 2 or x[0] in sys.argv[1:]]:
    w = WORKLOADS[name]
    t0 = time.time()
    hist, labels, n, T = synth_users(name, U)
    sd = init_lru_state_dict(w["V"], seed=42)
    model = LRURec.from_state_dict(sd)
    ids = torch.from_numpy(hist).cuda()
    model.retrieve_topk(ids, 50, True); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); idx, sc = model.retrieve_topk(ids, 50, True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    flops = 2.0 * 64 * (w["V"] + 1) * U
    table_bytes = (w["V"] + 1) * 65 * 4
    print(f"{name:9s} U={U:6d} V={w['V']:8d} L={w['L']:3d} mean_hist={n.mean():6.1f}: {ms:8.3f} ms  {U/ms*1e3:10.0f} users/s  "
          f"item-GEMM {flops/ms/1e9:7.2f} TF/s(f32)  table {table_bytes/1e6:7.1f} MB  (setup {time.time()-t0:.1f}s)", flush=True)
