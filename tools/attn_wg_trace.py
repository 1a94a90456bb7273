"""Diagnostic: per-workgroup timeline of one flash-attention launch (s_memrealtime at entry / prologue done / key-block
loop done / exit, plus the CU the workgroup ran on). Answers what a query tile costs besides its key blocks: prologue,
epilogue, and the gap between two workgroups on one CU slot. Needs the experiment build (see tools/attn_stamps.py).
usage: attn_wg_trace.py B T [variant = 2 | 4]"""
import ctypes as C, os, sys
os.environ["LR_ATTN_STAMPS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr
B, T = int(sys.argv[1]), int(sys.argv[2])
VAR = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nh, hd = 32, 128
g = torch.Generator(device="cuda"); g.manual_seed(0)
qkv = (torch.randn(B * T, 3 * nh * hd, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty(B * T, nh * hd, dtype=torch.bfloat16, device="cuda")
cu_h = np.arange(B + 1, dtype=np.int32) * T
cu = torch.from_numpy(cu_h).cuda()
l = lib(); l.lr_debug_attn_wg_trace.argtypes = [C.c_void_p, C.c_int]
call = lambda: check(l.lr_attention_varlen(qkv.data_ptr(), out.data_ptr(), cu.data_ptr(), cu_h.ctypes.data, B, nh, nh, hd, VAR, stream_ptr()), "attn")
for _ in range(3): call()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); call(); t1.record(); torch.cuda.synchronize()
print("variant %d B=%d T=%d stamped launch: %.1f us" % (VAR, B, T, t0.elapsed_time(t1) * 1e3))
n_wg = min(8192, B * nh * ((T + 127) // 128))
s = np.zeros(n_wg * 8, np.uint64)
check(l.lr_debug_attn_wg_trace(s.ctypes.data, s.size), "trace")
s = s.reshape(n_wg, 8)
s = s[s[:, 3] > 0]
t = s[:, :4].astype(np.float64) * 0.01   # us
t -= t[:, 0].min()
hw, xcc, nb = s[:, 4].astype(np.int64), s[:, 5].astype(np.int64) & 0xf, s[:, 6].astype(np.int64)
cu_id = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)   # xcc | SE | SH | CU
pro, loop, epi, life = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
print("workgroups traced %d on %d CUs; launch span %.1f us" % (len(s), len(np.unique(cu_id)), t[:, 3].max()))
print("per workgroup, us: prologue mean %.2f (median %.2f, p90 %.2f) | loop mean %.2f = %.2f per key block | epilogue mean %.2f (p90 %.2f) | life %.2f"
      % (pro.mean(), np.median(pro), np.percentile(pro, 90), loop.mean(), loop.sum() / nb.sum(), epi.mean(), np.percentile(epi, 90), life.mean()))
for k in sorted(set(nb.tolist())):
    m = nb == k
    print("  %2d key blocks: %4d wgs  prologue %.2f  loop %.2f (%.2f/block)  epilogue %.2f" % (k, m.sum(), pro[m].mean(), loop[m].mean(), loop[m].mean() / k, epi[m].mean()))
# per CU: two slots; occupancy = sum of lives / (2 * span), idle = time with fewer than 2 resident
span = t[:, 3].max()
occ, first, last = [], [], []
for c in np.unique(cu_id):
    m = cu_id == c
    occ.append(life[m].sum() / (2 * span)); first.append(t[m, 0].min()); last.append(t[m, 3].max())
print("per CU: resident-workgroup occupancy of its 2 slots mean %.3f (min %.3f); first entry %.1f..%.1f us; last exit %.1f..%.1f us"
      % (np.mean(occ), np.min(occ), np.min(first), np.max(first), np.min(last), np.max(last)))
# gap between an exit and the next entry on the same CU (dispatch latency), from the merged event list
gaps = []
for c in np.unique(cu_id)[:64]:
    m = cu_id == c
    ev = sorted([(x, +1) for x in t[m, 0]] + [(x, -1) for x in t[m, 3]])
    res, t_prev = 0, 0.0
    idle = {0: 0.0, 1: 0.0, 2: 0.0}
    for x, d in ev:
        idle[min(res, 2)] += x - t_prev; t_prev = x; res += d
    gaps.append((idle[0], idle[1], idle[2]))
gaps = np.array(gaps)
print("per CU (first 64): time with 0 / 1 / 2+ resident workgroups: %.1f / %.1f / %.1f us" % tuple(gaps.mean(0)))
