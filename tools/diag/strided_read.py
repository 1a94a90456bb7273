"""Does a column block of a wide row-major matrix read slower than a contiguous one? (The LoRA step's d B reductions read 4 096 of
dqkv's 12 288 columns.) torch copies, timing only."""
import torch
n = 7057
for ld, off in ((4096, 0), (12288, 0), (12288, 8192), (12288 + 64, 0), (16384, 0)):
    x = torch.randn(n, ld, device="cuda").to(torch.bfloat16)
    y = torch.empty(n, 4096, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        y.copy_(x[:, off:off + 4096])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y.copy_(x[:, off:off + 4096])
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    print(f"row stride {ld} elements, columns {off}..{off + 4096}: {t:.1f} us per copy, {2 * n * 4096 * 2 / t / 1e6:.2f} TB/s (read + write)")
