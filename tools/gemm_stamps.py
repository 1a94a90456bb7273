"""Diagnostic: where a workgroup of the 256x256x64 GEMM spends its time (s_memtime / s_memrealtime stamps) and how the
workgroups of a launch line up in time. Needs the experiment build: `make -C llamarec_amd/csrc -j8 EXPERIMENTS=1` (box-local;
the product library holds no stamping code); LR_GEMM_STAMPS=1 selects the stamped instantiation.

  python tools/gemm_stamps.py [M=32768] [shape=qkv|o|gate_up|down|all]
"""
import ctypes as C
import os
import sys

os.environ["LR_GEMM_STAMPS"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from llamarec_amd._lib import check, lib, stream_ptr

EPI = {"store": 0, "residual": 1, "swiglu": 2, "rope": 3}
SHAPES = {"qkv": (12288, 4096, "rope"), "o": (4096, 4096, "residual"), "gate_up": (22016, 4096, "swiglu"),
          "down": (4096, 11008, "residual")}


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    L = lib()
    L.lr_debug_gemm_stamps.argtypes = [C.c_void_p, C.c_int]
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    cs = torch.empty(L.lr_rope_table_bytes(4096, 128) // 4, dtype=torch.float32, device="cuda")
    check(L.lr_rope_table(cs.data_ptr(), 4096, 128, 10000.0, stream_ptr()), "rope table")
    pos = torch.cat([torch.arange(740, dtype=torch.int32)] * (M // 740 + 1))[:M].cuda()
    np.set_printoptions(precision=0, suppress=True, linewidth=220)
    for name, (N, K, epi) in SHAPES.items():
        if which not in ("all", name):
            continue
        A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
        B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
        R = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16) if epi == "residual" else None
        Cc = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")

        def run():
            check(L.lr_gemm_bf16_nt_epi(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), R.data_ptr() if R is not None else None, M, N, K,
                                        EPI[epi], 4, pos.data_ptr(), cs.data_ptr(), 4096, 128, 8192 if epi == "rope" else 0, None, 0,
                                        stream_ptr()), "gemm")

        for _ in range(12):                      # settle the clock under load first
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        nwg = min(16384, ((M + 255) // 256) * (N // 256))
        s = np.zeros(nwg * 2 * 8, np.uint64)
        check(L.lr_debug_gemm_stamps(s.ctypes.data, nwg), "stamps")
        s = s.reshape(nwg, 2, 8).astype(np.float64)
        t0, t1, t2, t3, t4, r0, r1, xcc = (s[:, :, i] for i in range(8))
        clock = (t4 - t0) / np.maximum(r1 - r0, 1) * 0.1      # GHz: shader cycles per 100 MHz tick
        start = (r0 - r0.min()) * 0.01                          # us since the first workgroup started
        end = (r1 - r0.min()) * 0.01
        print(f"== {name}: M={M} N={N} K={K} {epi}; stamped launch {ms * 1e3:.0f} us = {2.0 * M * N * K / ms / 1e9:.0f} TF/s; "
              f"{nwg} workgroups, in-kernel clock {np.median(clock):.3f} GHz")
        for grp in (0, 1):
            pro, loop, epi_issue, epi_drain = (t1 - t0)[:, grp], (t2 - t1)[:, grp], (t3 - t2)[:, grp], (t4 - t3)[:, grp]
            tot = (t4 - t0)[:, grp]
            print(f"  wave group {grp}: cycles median  prologue {np.median(pro):7.0f} | K loop {np.median(loop):8.0f} "
                  f"({np.median(loop) / (K // 64):.0f} per K tile; 2048 = MFMA-bound) | epilogue issue {np.median(epi_issue):6.0f} | "
                  f"store drain {np.median(epi_drain):6.0f} | total {np.median(tot):8.0f}  -> loop share {np.median(loop / tot):.3f}")
        # occupancy of the launch in time: how many workgroups are in their K loop at each instant
        dur = (end - start)[:, 0]
        print(f"  workgroup wall time us: median {np.median(dur):.1f}  p5 {np.percentile(dur, 5):.1f}  p95 {np.percentile(dur, 95):.1f}; "
              f"launch span {end.max():.0f} us; sum(workgroup time) / (256 CUs x span) = {dur.sum() / (256 * end.max()):.3f}")
        # how synchronised are the epilogues? real-time interval of each workgroup's epilogue (K loop end .. last store
        # drained, wave group 1 = the later one), and for each the number of OTHER workgroups inside theirs at its midpoint
        ck = np.maximum(clock[:, 1], 1e-3) * 1e3               # cycles per us
        es = start[:, 1] + (t2 - t0)[:, 1] / ck
        ee = start[:, 1] + (t4 - t0)[:, 1] / ck
        mid = 0.5 * (es + ee)
        order = np.argsort(es)
        es_s, ee_s = es[order], np.sort(ee)
        conc = np.searchsorted(es_s, mid, side="right") - np.searchsorted(ee_s, mid, side="left") - 1
        print(f"  epilogue: median {np.median(ee - es):.2f} us; other workgroups in their epilogue at its midpoint: median "
              f"{np.median(conc):.0f}  p10 {np.percentile(conc, 10):.0f}  p90 {np.percentile(conc, 90):.0f} (of {min(nwg, 256) - 1})")
        rounds = np.sort(start[:, 0])
        print("  workgroup start times (us), every 256th:", rounds[::256][:16])


if __name__ == "__main__":
    main()
