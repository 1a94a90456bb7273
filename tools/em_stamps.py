"""Diagnostic: where em_layer_kernel spends its cycles (s_memtime stamps; EXPERIMENTS build, LR_EM_STAMPS=1)."""
import ctypes as C, os, sys
os.environ["LR_EM_STAMPS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd.lru import LRURec, init_lru_state_dict
from llamarec_amd.synth import WORKLOADS, synth_users
from llamarec_amd._lib import check, lib
name, U = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("beauty", 22332)
w = WORKLOADS[name]
hist, labels, n, T = synth_users(name, U)
model = LRURec.from_state_dict(init_lru_state_dict(w["V"], seed=42))
ids = torch.from_numpy(hist).cuda()
for _ in range(3): model.retrieve_topk(ids, 50, True)
torch.cuda.synchronize()
l = lib(); l.lr_debug_em_stamps.argtypes = [C.c_void_p, C.c_int]
out = np.zeros(4 * 16 * 12, np.uint64)
check(l.lr_debug_em_stamps(out.ctypes.data, out.size), "stamps")
s = out.reshape(4, 16, 12).astype(np.float64)
np.set_printoptions(precision=0, suppress=True, linewidth=220)
print("s_memtime ticks (100 MHz? no: shader clock) summed over the super tiles of wave 0, mean of 16 workgroups; last launch of each mode")
print("cols: 0 prologue | 1 top barrier | 2 stage | 3 barrier | 4 phase A | 5 barrier | 6 recurrence | 7 barrier | 8 phase B chain | 9 hand-over | 10 barrier | 11 LayerNorm+store")
print("em_pipe_kernel (modes 0, 1 unless LR_EM_PIPE=0): B wave 0: 0 LayerNorm | 1 chain part 1 | 2 wait | 3 chain part 2 + hand-over | 4 wait | 10 prologue;  A wave 4: 5 phase A | 6 wait | 7 loads + recurrence + staging | 8 wait | 11 prologue")
for m, nm in enumerate(["LRU layer", "FFN", "LRU last block", "out_proj last rows"]):
    print(nm, s[m].mean(0), " total", s[m].mean(0).sum())
