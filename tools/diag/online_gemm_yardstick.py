"""The online path's four products at M = 460 rows: latency mode (gemm variant 5: split-K over 256 x 256 tiles) against torch.matmul
(hipBLASLt's MT256x256x64 stream-K kernel: a yardstick only) -- what an even cut of gate-up's 172 tiles over 256 CUs is worth."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from llamarec_amd._lib import check, lib, stream_ptr
M = int(sys.argv[1]) if len(sys.argv) > 1 else 460
g = torch.Generator(device="cuda"); g.manual_seed(0)
ws = torch.empty((64 << 20) + 4096, dtype=torch.uint8, device="cuda")
for name, N, K in (("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)):
    A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = {}
    for which in ("variant 5", "variant 4", "hipblaslt"):
        def once():
            if which == "hipblaslt":
                torch.matmul(A, B.T, out=C)
            else:
                check(lib().lr_gemm_bf16_nt_ws(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, int(which[-1]), ws.data_ptr(), ws.numel(), stream_ptr()), "gemm")
        for _ in range(5):
            once()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                once()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[which] = float(np.median(ts))
    print(f"M={M} {name:8s} N={N} K={K}: latency mode {res['variant 5']:.1f} us, one tile per workgroup {res['variant 4']:.1f} us, hipBLASLt {res['hipblaslt']:.1f} us", flush=True)
