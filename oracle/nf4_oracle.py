"""numpy ORACLE of the load-time NF4 round trip (llamarec_amd/csrc/llama_nf4.hip). TEST INFRASTRUCTURE ONLY.

Restates the published algorithm behind the reference's `BitsAndBytesConfig(load_in_4bit=True,
bnb_4bit_quant_type="nf4", bnb_4bit_use_double_quant=True, bnb_4bit_compute_dtype=bfloat16)`
(train_ranker.py:49-56, setup_demo.py:67-74): QLoRA (Dettmers et al. 2023) 4-bit NormalFloat with double
quantisation as implemented by bitsandbytes 0.43.1 (environment.yml:327; functional.quantize_4bit /
dequantize_4bit / create_dynamic_map). bitsandbytes is CUDA-only and absent here: **parity with it is unpinned**;
the reference holds no fixture at this boundary.
"""
from __future__ import annotations

import numpy as np

NF4 = np.array([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941, 0.7229568362236023,
                1.0], dtype=np.float32)
# thresholds of the published decision tree = midpoints between neighbouring levels
NF4_MID = np.array([-0.8480964004993439, -0.6106329262256622, -0.4599952697753906, -0.33967943489551544,
                    -0.23460740596055984, -0.13791173323988914, -0.045525018125772476, 0.03979014977812767,
                    0.1202552504837513, 0.2035212516784668, 0.2920137718319893, 0.3893125355243683, 0.5016634166240692,
                    0.6427869200706482, 0.8614784181118011], dtype=np.float32)


def dynamic_map():
    """create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8): 256 sorted levels in [-1, 1]."""
    import torch   # the published generator is written with torch.linspace in fp32; same calls here

    data = []
    for i in range(7):
        items = 2 ** i + 1
        boundaries = torch.linspace(0.1, 1, items)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-6 + i)) * means).tolist()
        data += (-(10 ** (-6 + i)) * means).tolist()
    data += [0, 1.0]
    assert len(data) == 256
    data.sort()
    return torch.tensor(data, dtype=torch.float32).numpy()


def _bf16_round(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def roundtrip(w, double_quant=True, code8=None):
    """w: bf16-valued float32 array (any shape, row-major blocks of 64). Returns the bf16-valued dequantised array."""
    shape = w.shape
    x = np.ascontiguousarray(w, dtype=np.float32).ravel()
    n = x.size
    nb = (n + 63) // 64
    xp = np.zeros(nb * 64, np.float32)
    xp[:n] = x
    blocks = xp.reshape(nb, 64)
    absmax = np.abs(blocks).max(axis=1).astype(np.float32)
    absmax_q = absmax
    if double_quant:
        code8 = dynamic_map() if code8 is None else np.asarray(code8, np.float32)
        fixed = (absmax.astype(np.float64) * float(1 << 40)).astype(np.uint64)          # exact, order-independent
        offset = np.float32(float(int(fixed.astype(object).sum())) / float(1 << 40) / float(nb))
        a = (absmax - offset).astype(np.float32)
        nb2 = (nb + 255) // 256
        ap = np.zeros(nb2 * 256, np.float32)
        ap[:nb] = a
        a2 = np.abs(ap.reshape(nb2, 256)).max(axis=1).astype(np.float32)
        mids = (np.float32(0.5) * (code8[:-1] + code8[1:])).astype(np.float32)
        q = np.zeros_like(ap)
        for j in range(nb2):
            if a2[j] > 0:
                v = (ap[j * 256:(j + 1) * 256] * (np.float32(1.0) / a2[j])).astype(np.float32)
                idx = np.searchsorted(mids, v, side="left")                                 # first mid >= v
                q[j * 256:(j + 1) * 256] = (code8[idx] * a2[j]).astype(np.float32)
        absmax_q = (q[:nb] + offset).astype(np.float32)
    out = np.zeros_like(blocks)
    nz = absmax > 0
    inv = np.zeros_like(absmax)
    inv[nz] = (np.float32(1.0) / absmax[nz]).astype(np.float32)
    v = (blocks * inv[:, None]).astype(np.float32)
    idx = (v[:, :, None] > NF4_MID[None, None, :]).sum(axis=2)
    out = (NF4[idx] * absmax_q[:, None]).astype(np.float32)
    out[~nz] = 0
    return _bf16_round(out).ravel()[:n].reshape(shape)
