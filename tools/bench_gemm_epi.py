"""The four projection GEMMs of a Llama-2-7b layer WITH the epilogue each one carries in the prefill (qkv: RoPE, o: residual,
gate/up: SwiGLU, down: residual) against the same product with a plain store -- interleaved rounds in one process, random
bf16 operands (MI355X_MICROARCH.md: never rank on zeros). Tells what each fused epilogue costs on top of the main loop.

  python tools/bench_gemm_epi.py [M=32768] [rounds=5] [extra variants, e.g. 6]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from llamarec_amd._lib import check, lib, stream_ptr

EPI = {"store": 0, "residual": 1, "swiglu": 2, "rope": 3}


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    extra_variants = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else []
    shapes = [("qkv", 12288, 4096, "rope"), ("o", 4096, 4096, "residual"), ("gate_up", 22016, 4096, "swiglu"),
              ("down", 4096, 11008, "residual")]
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    L = lib()
    cs = torch.empty(L.lr_rope_table_bytes(4096, 128) // 4, dtype=torch.float32, device="cuda")
    check(L.lr_rope_table(cs.data_ptr(), 4096, 128, 10000.0, stream_ptr()), "rope table")
    # token positions of packed Beauty-like prompts (~740 tokens each)
    pos = torch.cat([torch.arange(740, dtype=torch.int32)] * (M // 740 + 1))[:M].cuda()
    for name, N, K, epi in shapes:
        A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
        B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
        R = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16) if epi == "residual" else None
        C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        arms = [("store", 4), (epi, 4)] + [(epi, v) for v in extra_variants]
        times = {a: [] for a in arms}
        outs = {}
        for r in range(rounds + 1):
            for e, v in arms:
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    check(L.lr_gemm_bf16_nt_epi(A.data_ptr(), B.data_ptr(), C.data_ptr(), R.data_ptr() if (R is not None and e == "residual") else None,
                                                M, N, K, EPI[e], v, pos.data_ptr(), cs.data_ptr(), 4096, 128, 8192 if e == "rope" else 0,
                                                None, 0, stream_ptr()), "gemm")
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[(e, v)].append(e0.elapsed_time(e1) / 3)
                elif e == epi:
                    outs[v] = C[:, : (N // 2 if e == "swiglu" else N)].clone()
        fl = 2.0 * M * N * K
        ts, te = np.median(times[("store", 4)]), np.median(times[(epi, 4)])
        line = (f"{name:8s} M={M} N={N} K={K}: store {ts:.3f} ms {fl / ts / 1e9:.0f} TF/s | {epi} {te:.3f} ms {fl / te / 1e9:.0f} TF/s "
                f"({(te / ts - 1) * 100:+.1f} % time)")
        for v in extra_variants:
            tv = np.median(times[(epi, v)])
            same = torch.equal(outs[v].view(torch.int16), outs[4].view(torch.int16))
            line += f" | variant {v}: {tv:.3f} ms {fl / tv / 1e9:.0f} TF/s ({(tv / te - 1) * 100:+.1f} % vs v4, bits equal: {same})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
