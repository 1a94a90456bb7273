"""A/B the bf16 GEMM variants on the Llama-2-7b prefill shapes (interleaved rounds, one process)."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr

def run(variants, M=16384, rounds=5, check_equal=True):
    shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    ws = torch.empty((64 << 20) + 4096, dtype=torch.uint8, device="cuda")   # split-K planes / stream-K slots + flags
    for name, N, K in shapes:
        A = (torch.randn(M, K, generator=g, device="cuda") ).to(torch.bfloat16)
        B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
        outs = {}
        times = {v: [] for v in variants}
        for r in range(rounds + 1):
            for v in variants:
                C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    check(lib().lr_gemm_bf16_nt_ws(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, v, ws.data_ptr(),
                                                   ws.numel(), stream_ptr()), "gemm")
                e1.record(); torch.cuda.synchronize()
                if r: times[v].append(e0.elapsed_time(e1) / 3)
                outs[v] = C
        fl = 2.0 * M * N * K
        line = f"{name:8s} M={M} N={N} K={K}: "
        for v in variants:
            t = np.median(times[v]); line += f" v{v}: {t:.3f} ms {fl/t/1e9:.0f} TF/s (min {fl/min(times[v])/1e9:.0f}) |"
        if check_equal and len(variants) > 1:
            ref = outs[variants[0]].float()
            for v in variants[1:]:
                d = (outs[v].float() - ref).abs().max().item()
                line += f" maxdiff v{v}-v{variants[0]}={d:.4g}"
        print(line, flush=True)

if __name__ == "__main__":
    # python tools/bench_gemm.py <variants, e.g. 4 or 1,4> <M, or M1,M2,...> [yardstick]
    vs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4]
    Ms = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] != "yardstick" else [16384]
    for M in Ms:
        run(vs, M=M)


def yardstick(M=16384, rounds=5):
    """torch.matmul (hipBLASLt) on the same shapes/data -- a yardstick only, never the product."""
    shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for name, N, K in shapes:
        A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
        B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
        ts = []
        for r in range(rounds + 1):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                C = A @ B.T
            e1.record(); torch.cuda.synchronize()
            if r: ts.append(e0.elapsed_time(e1) / 3)
        fl = 2.0 * M * N * K
        print(f"yardstick torch.matmul {name:8s}: {np.median(ts):.3f} ms {fl/np.median(ts)/1e9:.0f} TF/s", flush=True)


if __name__ == "__main__" and "yardstick" in sys.argv[2:]:
    for M in ([int(x) for x in sys.argv[2].split(",")] if sys.argv[2] != "yardstick" else [16384]):
        yardstick(M)
