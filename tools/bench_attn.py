"""Stand-alone timing of the varlen flash-attention kernel at the bench's shape (32 prompts x 460 tokens, 32 heads x 128)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd._lib import check, lib, stream_ptr
B, T, nh, hd = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 460, 32, 128
VAR = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator(device="cuda"); g.manual_seed(0)
qkv = (torch.randn(B * T, 3 * nh * hd, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty(B * T, nh * hd, dtype=torch.bfloat16, device="cuda")
cu_h = np.arange(B + 1, dtype=np.int32) * T
cu = torch.from_numpy(cu_h).cuda(); l = lib()
run = lambda: check(l.lr_attention_varlen(qkv.data_ptr(), out.data_ptr(), cu.data_ptr(), cu_h.ctypes.data, B, nh, nh, hd, VAR, stream_ptr()), "attn")
for _ in range(3): run()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20): run()
t1.record(); torch.cuda.synchronize()
us = t0.elapsed_time(t1) * 50
print("attention variant %d B=%d T=%d: %.1f us, %.0f TF/s (causal flops)" % (VAR, B, T, us, B * 4 * nh * hd * (T * (T + 1) / 2) / us / 1e6))
