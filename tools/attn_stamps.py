"""Diagnostic: where the flash-attention key-block loop spends its cycles (s_memtime stamps). Needs the experiment
build of the library: `make -C llamarec_amd/csrc clean && make -C llamarec_amd/csrc -j8 EXPERIMENTS=1` (the product
library holds no stamping code); LR_ATTN_STAMPS=1 then selects the stamped instantiation."""
import ctypes as C, os, sys
os.environ["LR_ATTN_STAMPS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr
B, T, nh, hd = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 460, 32, 128
g = torch.Generator(device="cuda"); g.manual_seed(0)
qkv = (torch.randn(B * T, 3 * nh * hd, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty(B * T, nh * hd, dtype=torch.bfloat16, device="cuda")
cu_h = np.arange(B + 1, dtype=np.int32) * T
cu = torch.from_numpy(cu_h).cuda()
l = lib(); l.lr_debug_attn_stamps.argtypes = [C.c_void_p, C.c_int]
for _ in range(3):
    check(l.lr_attention_varlen(qkv.data_ptr(), out.data_ptr(), cu.data_ptr(), cu_h.ctypes.data, B, nh, nh, hd, 2, stream_ptr()), "attn")
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10):
    check(l.lr_attention_varlen(qkv.data_ptr(), out.data_ptr(), cu.data_ptr(), cu_h.ctypes.data, B, nh, nh, hd, 2, stream_ptr()), "attn")
t1.record(); torch.cuda.synchronize()
print("stamped kernel: %.1f us per launch" % (t0.elapsed_time(t1) * 100))
s = np.zeros(64 * 8, np.uint64)
check(l.lr_debug_attn_stamps(s.ctypes.data, s.size), "stamps")
s = s.reshape(64, 8).astype(np.float64)
np.set_printoptions(precision=0, suppress=True, linewidth=200)
print("per workgroup (wave 0): prologue | DMA issue | S=KQ^T | softmax | PV | barrier+DMA wait | epilogue | key blocks")
print(s[:12])
nb = s[:, 7:8]
print("per key block, mean over workgroups:", (s[:, 1:6] / nb).mean(0), " prologue", s[:, 0].mean(), "epilogue", s[:, 6].mean())
