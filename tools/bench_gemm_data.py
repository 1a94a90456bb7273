"""How much of the bf16 GEMM's rate is set by what the operands hold (power -> clock), not by the instruction stream: the product
kernel (variant 4) and torch.matmul (hipBLASLt, a yardstick only) on M = 32 768 rows of the gate-up and o shapes with operands of
different bit activity, each held for >= 100 back-to-back launches so that the clock settles. Round 5: a timing-only arm of the
folded-RMSNorm A/B whose activations had gone non-finite ran the SAME kernels at 1 635 TF/s against 1 430 (gpurun_out/foldab)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr


def fill(kind, shape, g, scale):
    if kind == "normal":
        return (torch.randn(shape, generator=g, device="cuda") * scale).to(torch.bfloat16)
    if kind == "zeros":
        return torch.zeros(shape, dtype=torch.bfloat16, device="cuda")
    if kind == "ones":
        return torch.full(shape, scale, dtype=torch.bfloat16, device="cuda")
    if kind == "nan":
        return torch.full(shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    if kind == "half_zero":      # every other K element zero
        x = (torch.randn(shape, generator=g, device="cuda") * scale).to(torch.bfloat16)
        x[:, ::2] = 0
        return x
    if kind == "small_ints":     # values in {-2 .. 2}: few mantissa bits set
        return torch.randint(-2, 3, shape, generator=g, device="cuda").to(torch.bfloat16) * scale
    raise ValueError(kind)


def main():
    M, launches = 32768, 100
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    ws = torch.empty((64 << 20) + 4096, dtype=torch.uint8, device="cuda")
    for name, N, K in (("gate_up", 22016, 4096), ("o", 4096, 4096)):
        C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        for kind in ("normal", "small_ints", "half_zero", "ones", "zeros", "nan", "normal"):
            A, B = fill(kind, (M, K), g, 1.0), fill(kind, (N, K), g, 0.02)
            res = []
            for which in ("product", "hipblaslt"):
                def once():
                    if which == "product":
                        check(lib().lr_gemm_bf16_nt_ws(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, 4, ws.data_ptr(), ws.numel(), stream_ptr()), "gemm")
                    else:
                        torch.matmul(A, B.T, out=C)
                for _ in range(10):
                    once()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(launches):
                    once()
                e1.record(); torch.cuda.synchronize()
                res.append(2.0 * M * N * K * launches / (e0.elapsed_time(e1) * 1e-3) / 1e12)
            print(f"{name:8s} N={N} K={K} operands {kind:10s}: product {res[0]:7.0f} TF/s   hipBLASLt {res[1]:7.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
