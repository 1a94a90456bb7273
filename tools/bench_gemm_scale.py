"""Does the GEMM rate depend on the VALUES of the A operand? (gate-up shape, 32 768 rows.) Scale, per-row scale, zeros:
magnitudes do not matter (1419-1436 TF/s), all-zero operands run 22 % faster (1745 TF/s): the clock is power-limited on
real data, and 0.70 of peak is what this instruction stream issues when it is not."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr
M, N, K = 32768, 22016, 4096
g = torch.Generator(device="cuda"); g.manual_seed(0)
ws = torch.empty((64 << 20) + 4096, dtype=torch.uint8, device="cuda")
B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
base = torch.randn(M, K, generator=g, device="cuda")
for name, A in (("x1", base), ("x8", base * 8), ("x64", base * 64), ("x0.05", base * 0.05), ("rowscaled", base * torch.exp(torch.randn(M, 1, generator=g, device="cuda") * 1.5)), ("zeros", base * 0)):
    A = A.to(torch.bfloat16)
    ts = []
    for r in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            check(lib().lr_gemm_bf16_nt_ws(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, 4, ws.data_ptr(), ws.numel(), stream_ptr()), "gemm")
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1) / 3)
    t = float(np.median(ts)); print(f"A {name:10s}: {t:.3f} ms {2.0*M*N*K/t/1e9:.0f} TF/s", flush=True)
