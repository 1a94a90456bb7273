"""NF4 + double-quantisation round trip (llamarec_amd/csrc/llama_nf4.hip): oracle properties and the 8-bit code table on
CPU, HIP kernel against the numpy oracle bit for bit and the ranker wiring on GPU. bitsandbytes is absent: the published
algorithm is restated, parity with bitsandbytes itself is unpinned."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from llamarec_amd.synth import bf16_round
from oracle import nf4_oracle as N


def test_dynamic_map_of_the_library_equals_the_published_generator():
    from llamarec_amd._lib import check, lib

    out = (C.c_float * 256)()
    check(lib().lr_nf4_dynamic_map(out), "lr_nf4_dynamic_map")
    table = np.array(out[:], dtype=np.float32)
    ref = N.dynamic_map()
    assert np.array_equal(table, ref)
    assert table[0] < -0.99 and table[-1] == 1.0 and (np.diff(table) > 0).all() and (table == 0).sum() == 1


def test_oracle_properties():
    rng = np.random.default_rng(1)
    w = bf16_round((rng.standard_normal((257, 96)) * 0.02).astype(np.float32))
    w[3, :64] = 0                                              # an all-zero block stays zero
    for dq in (False, True):
        r = N.roundtrip(w, double_quant=dq)
        assert r.shape == w.shape and np.array_equal(r, bf16_round(r))
        assert np.all(r[3, :64] == 0)
        rel = np.linalg.norm(r - w) / np.linalg.norm(w)
        assert 0.05 < rel < 0.13                               # NF4's quantisation noise on normal weights (~9 %)
        assert np.all(np.sign(r[w != 0]) * np.sign(w[w != 0]) >= 0)
    # at most 16 distinct values per block, the block's extreme element is reproduced exactly (level +-1)
    r = N.roundtrip(w, double_quant=False)
    blocks_w, blocks_r = w.ravel()[: 64 * 100].reshape(100, 64), r.ravel()[: 64 * 100].reshape(100, 64)
    assert max(len(np.unique(b)) for b in blocks_r) <= 16
    i = np.abs(blocks_w).argmax(axis=1)
    assert np.array_equal(np.abs(blocks_r[np.arange(100), i]), np.abs(blocks_w[np.arange(100), i]))
    # idempotent without double quantisation
    assert np.array_equal(N.roundtrip(r, double_quant=False), r)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dq", [((300, 128), 1), ((300, 128), 0), ((1024, 1024), 1), ((7, 100), 1), ((1, 3), 1)])
def test_hip_roundtrip_equals_oracle_bit_for_bit(shape, dq):
    import torch

    from llamarec_amd._lib import check, lib, stream_ptr

    rng = np.random.default_rng(shape[0] + dq)
    w = bf16_round((rng.standard_normal(shape) * 0.02).astype(np.float32))
    if shape[0] >= 300:
        w[5, :64] = 0
        w[9] *= 40.0                                           # a row of outliers: very different absmax values
        w = bf16_round(w)
    ref = N.roundtrip(w, double_quant=bool(dq))
    dev = torch.device("cuda:0")
    t = torch.from_numpy(w).to(torch.bfloat16).to(dev)
    out = torch.empty_like(t)
    nbytes = lib().lr_nf4_scratch_bytes(t.numel())
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().lr_nf4_roundtrip_bf16(t.data_ptr(), t.numel(), dq, out.data_ptr(), scratch.data_ptr(), nbytes,
                                      stream_ptr()), "lr_nf4_roundtrip_bf16")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), float(np.abs(got - ref).max())


@pytest.mark.gpu
def test_ranker_loaded_in_4bit_equals_ranker_of_round_tripped_weights(golden_dir):
    """`from_state_dict(nf4=True)`: the seven Linears of every layer go through the round trip, lm_head / embeddings /
    norms do not, and an adapter is merged after it."""
    from llamarec_amd.llm import LlamaRanker
    from llamarec_amd.synth import synth_llama_state

    z = np.load(os.path.join(golden_dir, "llama_lora_train_tiny_hd128.npz"))
    cfg = json.loads(str(z["config"]))
    sd = synth_llama_state(cfg, 7)
    lin = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
    sd_rt = {k: (N.roundtrip(v) if any(k.endswith(n + ".weight") for n in lin) else v) for k, v in sd.items()}
    names = [str(n) for n in z["param_names"]]
    weights = {}
    for n in names:
        _, l, proj, ab = n.split(".")
        weights[f"model.layers.{l}.self_attn.{proj}.{ab}.weight"] = z["init/" + n]
    lora = dict(r=int(z["lora_r"]), alpha=float(z["lora_alpha"]), weights=weights)
    seqs = [np.arange(3, 3 + n, dtype=np.int32) for n in (40, 129, 7)]
    label_ids = np.arange(10, 30, dtype=np.int32)
    a = LlamaRanker.from_state_dict(sd, cfg, lora=lora, nf4=True).prefill_verbalize(seqs, label_ids).cpu().numpy()
    b = LlamaRanker.from_state_dict(sd_rt, cfg, lora=lora).prefill_verbalize(seqs, label_ids).cpu().numpy()
    c = LlamaRanker.from_state_dict(sd, cfg, lora=lora).prefill_verbalize(seqs, label_ids).cpu().numpy()
    assert np.array_equal(a, b)
    assert np.abs(a - c).max() > 1e-3                           # and it is not the unquantised model
