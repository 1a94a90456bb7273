"""GEMM rate of the 256 x 256 kernel against N (K = 4096, M = 32768 rows): is the qkv product's lower rate a property of its
width? Usage: python tools/bench_gemm_nsweep.py [M]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
K = 4096
g = torch.Generator(device="cuda"); g.manual_seed(0)
ws = torch.empty((64 << 20) + 4096, dtype=torch.uint8, device="cuda")
A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
for N in (4096, 8192, 12288, 16384, 22016, 24576, 32768):
    B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ts = []
    for r in range(6):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            check(lib().lr_gemm_bf16_nt_ws(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, 4, ws.data_ptr(), ws.numel(), stream_ptr()), "gemm")
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1) / 3)
    t = float(np.median(ts)); fl = 2.0 * M * N * K
    tiles = (M // 256) * (N // 256)
    print(f"N={N:6d} tiles={tiles:6d} rounds={tiles/256:6.2f}: {t:.3f} ms  {fl/t/1e9:.0f} TF/s (min {fl/min(ts)/1e9:.0f})", flush=True)
    del B, C
