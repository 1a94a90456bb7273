"""Stand-alone timing of the varlen flash-attention kernel (32 heads x 128) at the bench's shapes:
   python tools/bench_attn.py            -> 32 x 460 tokens (ML-100k step), 16 x 740 (Beauty reference batch), and a
                                            token-budget Beauty step (about 23 prompts of 565..1125 tokens, 16 384 rows)
   python tools/bench_attn.py B T [VAR]  -> B prompts of T tokens"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr

nh, hd = 32, 128


def run(lens, var=0, name=""):
    lens = np.asarray(lens, dtype=np.int64)
    B, n = len(lens), int(lens.sum())
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    qkv = (torch.randn(n, 3 * nh * hd, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty(n, nh * hd, dtype=torch.bfloat16, device="cuda")
    cu_h = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cu = torch.from_numpy(cu_h).cuda(); l = lib()
    wsb = l.lr_attention_workspace_bytes(n, B, nh)
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    call = lambda: check(l.lr_attention_varlen_ws(qkv.data_ptr(), out.data_ptr(), None, cu.data_ptr(), cu_h.ctypes.data, B, nh, nh, hd, var,
                                                  ws.data_ptr(), wsb, stream_ptr()), "attn")
    for _ in range(3): call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10): call()
        t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) * 100)
    us = float(np.median(ts))
    fl = float((4.0 * nh * hd * (lens * (lens + 1) / 2)).sum())
    print("attention variant %d %s B=%d tokens=%d: %.1f us (min %.1f), %.0f TF/s (causal flops)" % (var, name, B, n, us, min(ts), fl / us / 1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "beauty":      # the token-budget Beauty step alone: python tools/bench_attn.py beauty VAR
        from llamarec_amd.packing import token_budget_steps
        from llamarec_amd.synth import synth_users
        T = synth_users("beauty", 100)[3]
        run(T[token_budget_steps(T)[0]], int(sys.argv[2]), name="beauty token-budget step")
    elif len(sys.argv) > 2:
        run([int(sys.argv[2])] * int(sys.argv[1]), int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    else:
        from llamarec_amd.packing import token_budget_steps
        from llamarec_amd.synth import synth_users
        variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
        T = synth_users("beauty", 100)[3]
        for v in variants:
            run([460] * 32, v, name="ml-100k step (32 x 460)")
            run([740] * 16, v, name="beauty reference batch (16 x 740)")
            run(T[token_budget_steps(T)[0]], v, name="beauty token-budget step")
            run([1000] * 16, v, name="16 x 1000")
            run([8192] * 4, v, name="4 x 8192")
