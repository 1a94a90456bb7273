"""First-light check of attention variant 3 (llama_attn256.hip) against a float64 numpy reference and variant 2."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr


def ref(qkv, cu, nh, nkv, hd):
    n = qkv.shape[0]
    out = np.zeros((n, nh * hd)); lse = np.zeros((n, nh))
    for b in range(len(cu) - 1):
        s, e = cu[b], cu[b + 1]; T = e - s
        q = qkv[s:e, :nh * hd].reshape(T, nh, hd).astype(np.float64)
        k = qkv[s:e, nh * hd:(nh + nkv) * hd].reshape(T, nkv, hd).astype(np.float64)
        v = qkv[s:e, (nh + nkv) * hd:].reshape(T, nkv, hd).astype(np.float64)
        mask = np.tril(np.ones((T, T), bool))
        for h in range(nh):
            sc = (q[:, h] @ k[:, h // (nh // nkv)].T) / np.sqrt(hd)
            sc = np.where(mask, sc, -np.inf)
            m = sc.max(-1, keepdims=True); p = np.exp(sc - m)
            out[s:e, h * hd:(h + 1) * hd] = (p @ v[:, h // (nh // nkv)]) / p.sum(-1, keepdims=True)
            lse[s:e, h] = (m + np.log(p.sum(-1, keepdims=True)))[:, 0]
    return out, lse


def run(lens, nh, nkv, hd, var, scale=1.0, want_lse=True, seed=0):
    cu_h = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32); n = int(cu_h[-1])
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    qkv = (torch.randn(n, (nh + 2 * nkv) * hd, generator=g, device="cuda") * scale).to(torch.bfloat16)
    out = torch.full((n, nh * hd), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.full((n, nh), float("nan"), dtype=torch.float32, device="cuda")
    cu = torch.from_numpy(cu_h).cuda(); l = lib()
    wsb = l.lr_attention_workspace_bytes(n, len(lens), nh)
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    check(l.lr_attention_varlen_ws(qkv.data_ptr(), out.data_ptr(), lse.data_ptr() if want_lse else None, cu.data_ptr(), cu_h.ctypes.data,
                                   len(lens), nh, nkv, hd, var, ws.data_ptr(), wsb, stream_ptr()), "attn")
    torch.cuda.synchronize()
    return qkv.float().cpu().numpy(), out.float().cpu().numpy(), lse.cpu().numpy(), cu_h


if __name__ == "__main__":
    ok = True
    for (lens, nh, nkv, scale) in [([64], 8, 8, 1.0), ([1, 63, 64, 65, 128, 129, 300, 2, 256, 257, 600], 8, 8, 1.0),
                                   ([255, 511, 513, 1125, 740], 8, 2, 1.0), ([700, 33], 3, 1, 3.0), ([2048], 8, 8, 0.5)]:
        for var in (3, 3, 2):
            qkv, out, lse, cu = run(lens, nh, nkv, 128, var, scale)
            r, rl = ref(qkv, cu, nh, nkv, 128)
            e = np.abs(out - r); el = np.abs(lse - rl)
            bad = ~np.isfinite(out)
            print("variant %d lens %s nh %d nkv %d scale %.1f: max err %.4g (rows nan: %d) lse err %.3g" %
                  (var, lens, nh, nkv, scale, np.nanmax(e), int(bad.any(1).sum()), np.nanmax(el)), flush=True)
            if var == 3 and (bad.any() or e.max() > 2e-2 * max(1.0, scale) or el.max() > 2e-2):
                ok = False
                rows = np.where((e.max(1) > 2e-2 * max(1.0, scale)) | bad.any(1))[0]
                for b in range(len(cu) - 1):
                    rb = rows[(rows >= cu[b]) & (rows < cu[b + 1])] - cu[b]
                    if len(rb):
                        heads = sorted(set(int(hh) for r_ in rb for hh in np.where(e[cu[b] + r_].reshape(nh, -1).max(1) > 2e-2 * max(1.0, scale))[0]))
                        print("   segment %d (T=%d): %d bad rows, local %d..%d, heads %s" % (b, cu[b + 1] - cu[b], len(rb), rb.min(), rb.max(), heads), flush=True)
    print("OK" if ok else "FAILED")
