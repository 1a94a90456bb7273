"""Reads the per-segment cycle stamps of the diagnostic build of attention variant 3 (LLAMAREC_LIB=.../abl/libllamarec_stamps.so)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, stream_ptr, LIB_PATH
l = C.CDLL(LIB_PATH)
nh, hd = 32, 128
lens = [int(x) for x in sys.argv[1:]] or [8192] * 4
B, n = len(lens), sum(lens)
qkv = (torch.randn(n, 3 * nh * hd, device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty(n, nh * hd, dtype=torch.bfloat16, device="cuda")
cu_h = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32); cu = torch.from_numpy(cu_h).cuda()
l.lr_attention_workspace_bytes.restype = C.c_size_t
wsb = l.lr_attention_workspace_bytes(n, B, nh); ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
for _ in range(3):
    rc = l.lr_attention_varlen_ws(C.c_void_p(qkv.data_ptr()), C.c_void_p(out.data_ptr()), None, C.c_void_p(cu.data_ptr()), C.c_void_p(cu_h.ctypes.data),
                                  B, nh, nh, hd, 3, C.c_void_p(ws.data_ptr()), C.c_size_t(wsb), C.c_void_p(stream_ptr()))
    assert rc == 0
torch.cuda.synchronize()
st2 = np.zeros(64 * 24, dtype=np.uint64)
assert l.lr_debug_attn256_stamps(st2.ctypes.data_as(C.c_void_p), 64 * 24) == 0
st2u = st2.reshape(64, 2, 12)
st2 = st2u.astype(np.float64)
for w, name in ((0, "steady-state blocks"), (1, "first blocks (not diagonal)")):
    st = st2[:, w]
    blocks = np.maximum(st[:, 9], 1)
    per = st[:, :9] / blocks[:, None]
    print("wave 3, %s per workgroup: median %d" % (name, np.median(st[:, 9])))
    print("  cycles per 8-gap segment (median over 64 workgroups):", " ".join("%.0f" % x for x in np.median(per[:, :8], 0)))
    print("  waits + barrier: %.0f   block total: %.0f   (64 MFMAs = 2048 cycles)" % (np.median(per[:, 8]), np.median(st[:, 10] / blocks)))
    if w == 1:
        raw = st2u[:, 1, 11]
        print("  first blocks behind an epilogue: vmcnt(24) wait %.0f, barrier %.0f (per first block)"
              % (np.median((raw & np.uint64(0xffffffff)).astype(np.float64) / blocks), np.median((raw >> np.uint64(32)).astype(np.float64) / blocks)))

ph = np.zeros(64 * 16, dtype=np.uint64)
assert l.lr_debug_attn256_phases(ph.ctypes.data_as(C.c_void_p), 64 * 16) == 0
ph = ph.reshape(64, 2, 8).astype(np.float64)
for w, name in ((0, "wave 0"), (1, "wave 3")):
    tiles = np.maximum(ph[:, w, 5], 1)
    m = np.median(ph[:, w, :5] / tiles[:, None], 0)
    tot = np.median(ph[:, w, :5].sum(1))
    print("%s: tiles %d; cycles per tile: prologue %.0f, key blocks %.0f, drain+staging-only blocks %.0f, epilogue %.0f, ticket+rendezvous %.0f; kernel total %.0f"
          % (name, np.median(ph[:, w, 5]), m[0], m[1], m[2], m[3], m[4], tot))

bk = np.zeros(64 * 16, dtype=np.uint64)
assert l.lr_debug_attn256_blocks(bk.ctypes.data_as(C.c_void_p), 64 * 16) == 0
bk = bk.reshape(64, 8, 2).astype(np.float64)
names = ["first", "second", "steady", "third-last", "second-last", "last"]
print("wave 3, cycles per key block (incl. its barrier) by position in the tile: " +
      ", ".join("%s %.0f (x%.1f per tile)" % (names[i], np.median(bk[:, i, 0] / np.maximum(bk[:, i, 1], 1)), np.median(bk[:, i, 1]) / max(1.0, np.median(ph[:, 1, 5])))
                for i in range(6)))
nf = np.maximum(bk[:, 0, 1], 1)
fl = bk.reshape(64, 16)
print("wave 3, a tile's first block (not diagonal), cycles: entry code %.0f, into the first statement %.0f, behind the last statement (LDS-read wait) %.0f, "
      "to the block-end wait %.0f" % tuple(np.median(fl[:, 12 + i] / nf) for i in range(4)))
