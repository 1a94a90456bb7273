"""Diagnostic: where the item top-K tile loop spends its cycles (s_memtime stamps, LR_TOPK_STAMPS=1)."""
import ctypes as C, os, sys
os.environ["LR_TOPK_STAMPS"] = "1"
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd.lru import LRURec, init_lru_state_dict
from llamarec_amd.synth import WORKLOADS, synth_users
from llamarec_amd._lib import check, lib
name, U = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("synth-1m", 4096)
w = WORKLOADS[name]
hist, labels, n, T = synth_users(name, U)
model = LRURec.from_state_dict(init_lru_state_dict(w["V"], seed=42))
ids = torch.from_numpy(hist).cuda()
for _ in range(2): model.retrieve_topk(ids, 50, True)
torch.cuda.synchronize()
l = lib(); l.lr_debug_topk_stamps.argtypes = [C.c_void_p, C.c_int]
out = np.zeros(16 * 8, np.uint64)
check(l.lr_debug_topk_stamps(out.ctypes.data, out.size), "stamps")
s = out.reshape(16, 8).astype(np.float64)
n_tiles = (w["V"] + 1 + 31) // 32
np.set_printoptions(precision=0, suppress=True, linewidth=200)
print("s_memtime ticks summed over the tile loop, wave 0 of 16 workgroups")
print("cols: frag reads | thr/bias/mask | chain+filter+inserts | make_room | store(+vmcnt) | barrier (first col also holds the loop-entry offset) | compactions (count) | inserts (count)")
tpc = (n_tiles + 7) // 8 if name == "synth-1m" else n_tiles
if os.environ.get("LR_TOPK_CHUNKS"): tpc = (n_tiles + int(os.environ["LR_TOPK_CHUNKS"]) - 1) // int(os.environ["LR_TOPK_CHUNKS"])
print("per tile (assuming", tpc, "tiles per chunk):")
print(s / tpc)
print("mean:", s.mean(0) / tpc, " total:", s.mean(0).sum() / tpc)
print("totals per workgroup (ticks): loop", s[:, :6].sum(1).mean(), " final", s[:, 6].mean(), " compactions", s[:, 6 if False else 6].mean() * 0, " inserts/tile", s[:, 7].mean() / tpc)
