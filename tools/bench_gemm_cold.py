"""Latency-mode GEMM (variant 5, M = 460) against where its weights are: 'hot' (the same weight matrix every call: L2 /
Infinity Cache), 'cold' (a ring of weight matrices larger than the 256 MB Infinity Cache: every tile a first HBM read), and
'warmed' (cold ring, but a streaming read of the matrix -- a stand-in for a prefetch kernel one product ahead -- runs just
before the GEMM, outside the timed region). Answers what weight prefetch into the Infinity Cache could buy the online path."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr

M = int(sys.argv[1]) if len(sys.argv) > 1 else 460
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]
g = torch.Generator(device="cuda"); g.manual_seed(0)
ws = torch.empty((64 << 20) + 4096, dtype=torch.uint8, device="cuda")
l = lib()
for name, N, K in shapes:
    ring = max(2, int(np.ceil(600e6 / (N * K * 2))))           # > 2 x the Infinity Cache
    A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    Bs = [(torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(ring)]
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    call = lambda B: check(l.lr_gemm_bf16_nt_ws(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, 5, ws.data_ptr(), ws.numel(), stream_ptr()), "gemm")
    res = {}
    for mode in ("hot", "cold", "warmed"):
        ts = []
        for it in range(3 * ring + 3):
            B = Bs[0] if mode == "hot" else Bs[it % ring]
            if mode == "warmed":
                B.view(torch.int16).max()                         # streaming read of the whole matrix
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call(B); e1.record(); torch.cuda.synchronize()
            if it >= 3: ts.append(e0.elapsed_time(e1) * 1e3)
        res[mode] = float(np.median(ts))
    print(f"{name:8s} M={M} N={N} K={K} ({N * K * 2 / 1e6:.0f} MB, ring of {ring}): hot {res['hot']:.1f} us | cold {res['cold']:.1f} us | "
          f"cold + streamed just before {res['warmed']:.1f} us", flush=True)
