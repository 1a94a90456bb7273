"""Diagnostic: where do the ping-pong GEMM's phases spend their cycles (s_memtime stamps, variant 9)."""
import ctypes as C, sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr
M, N, K = 14800, 12288, 4096
g = torch.Generator(device="cuda"); g.manual_seed(0)
A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
Cm = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
l = lib(); l.lr_debug_gemm_stamps.argtypes = [C.c_void_p, C.c_int]
for _ in range(3):
    check(l.lr_gemm_bf16_nt(A.data_ptr(), B.data_ptr(), Cm.data_ptr(), M, N, K, 9, stream_ptr()), "gemm9")
torch.cuda.synchronize()
out = np.zeros(8 * 2 * 24, np.uint64)
check(l.lr_debug_gemm_stamps(out.ctypes.data, out.size), "stamps")
s = out.reshape(8, 2, 4, 6).astype(np.float64) / (K // 64)   # [wg][group][phase][load, barA, mfma, barB] cycles per K tile
np.set_printoptions(precision=0, suppress=True, linewidth=200)
print("cycles per K tile, mean over 8 workgroups; columns = reads landed | DMA issue | vmcnt wait | wait@barrier A | MFMA issue | wait@barrier B")
for grp in range(2):
    print("group", grp)
    print(s[:, grp].mean(0))
    print(" per-tile total:", s[:, grp].mean(0).sum())
