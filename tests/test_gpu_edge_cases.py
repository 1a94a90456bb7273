"""GPU: edge cases of the C ABI -- empty and ragged inputs, extreme sizes, error reporting."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def lru():
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from oracle import lru_oracle as O

    sd = init_lru_state_dict(777, seed=5)
    return LRURec.from_state_dict(sd), O.LruOracle(sd)


def test_lru_ragged_and_degenerate_histories(lru):
    model, orc = lru
    rng = np.random.default_rng(1)
    for L in (1, 2, 16, 17, 33, 200, 257):
        ids = np.zeros((7, L), np.int64)
        lens = [0, 1, L, L // 2, min(L, 3), L - 1 if L > 1 else 1, L]
        for b, n in enumerate(lens):
            if n:
                ids[b, L - n:] = rng.integers(1, 778, size=n)  # repeats allowed (collisions in the history)
        idx, sc = model.retrieve_topk(ids, 50, True)
        oi, os_ = orc.retrieve_topk(ids, 50, True)
        assert np.array_equal(idx.cpu().numpy(), oi), L
        assert np.array_equal(_bits(sc.cpu().numpy()), _bits(os_)), L


def test_lru_empty_batch_and_k_limits(lru):
    from llamarec_amd._lib import LlamaRecError

    model, orc = lru
    idx, sc = model.retrieve_topk(np.zeros((0, 10), np.int64), 20, True)
    assert idx.shape == (0, 20)
    ids = np.arange(1, 11, dtype=np.int64)[None, :]
    i1, _ = model.retrieve_topk(ids, 1, True)
    i64, _ = model.retrieve_topk(ids, 64, True)
    o64, _ = orc.retrieve_topk(ids, 64, True)
    assert np.array_equal(i64.cpu().numpy(), o64) and i1[0, 0] == i64[0, 0]
    with pytest.raises(ValueError):
        model.retrieve_topk(ids, 65, True)
    with pytest.raises(ValueError):
        model.retrieve_topk(np.zeros(5, np.int64), 5, True)  # not [B, L]


def test_lru_many_users_one_call(lru):
    """More users than one workgroup tile and uneven tail; determinism across calls."""
    model, orc = lru
    rng = np.random.default_rng(2)
    ids = np.zeros((1000, 50), np.int64)
    for b in range(1000):
        n = int(rng.integers(1, 51))
        ids[b, 50 - n:] = rng.choice(777, size=n, replace=False) + 1
    a, _ = model.retrieve_topk(ids, 20, True)
    b_, _ = model.retrieve_topk(ids, 20, True)
    oi, _ = orc.retrieve_topk(ids, 20, True)
    assert torch.equal(a, b_) and np.array_equal(a.cpu().numpy(), oi)


def test_abi_error_codes(lru):
    from llamarec_amd._lib import lib, stream_ptr

    model, _ = lru
    l = lib()
    ids = torch.ones((2, 5), dtype=torch.int64, device="cuda")
    out = torch.empty((2, 5), dtype=torch.int32, device="cuda")
    ws = torch.empty(16, dtype=torch.uint8, device="cuda")  # far too small
    assert l.lr_lru_workspace_bytes(model._h, 2, 5, 5) > 16
    rc = l.lr_lru_retrieve_topk(model._h, ids.data_ptr(), 2, 5, 5, 1, out.data_ptr(), None, ws.data_ptr(), 16, stream_ptr())
    assert rc == -4 and b"workspace" in l.lr_last_error()
    rc = l.lr_lru_retrieve_topk(model._h, ids.data_ptr(), 2, 5, 99, 1, out.data_ptr(), None, ws.data_ptr(), 16, stream_ptr())
    assert rc == -1
    rc = l.lr_lru_retrieve_topk(model._h, None, 2, 5, 5, 1, out.data_ptr(), None, ws.data_ptr(), 16, stream_ptr())
    assert rc == -1 and b"null" in l.lr_last_error()
    h = C.c_void_p()
    rc = l.lr_lru_create(ws.data_ptr(), 16, 777, 2, C.byref(h))
    assert rc == -1 and b"layout needs" in l.lr_last_error()


@pytest.fixture(scope="module")
def tiny_llama():
    from llamarec_amd.llm import LlamaRanker
    from llamarec_amd.synth import synth_llama_state

    cfg = dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               num_key_value_heads=2, max_position_embeddings=1536, rms_norm_eps=1e-5, rope_theta=10000.0)
    sd = synth_llama_state(cfg, 21)
    return cfg, sd, LlamaRanker.from_state_dict(sd, cfg)


def test_llama_prompt_length_extremes(tiny_llama):
    """T = 1 and T = llm_max_text_len = 1536 (config.py:236) in one packed batch, MFMA kernels
    (12 query blocks, 24 key blocks) vs the oracle."""
    from oracle import llama_oracle as LO

    cfg, sd, model = tiny_llama
    rng = np.random.default_rng(3)
    seqs = [np.array([1]), np.concatenate([[1], rng.integers(3, 512, size=1535)]), np.concatenate([[1], rng.integers(3, 512, size=126)]),
            np.concatenate([[1], rng.integers(3, 512, size=127)]), np.concatenate([[1], rng.integers(3, 512, size=128)])]
    label_ids = list(range(100, 120))
    got = model.prefill_verbalize(seqs, label_ids).cpu().numpy()
    ref = LO.prefill_verbalize(sd, cfg, seqs, label_ids, "bf16")
    assert np.abs(got - ref).max() < 4e-2
    gen = model.set_variants(1, 1).prefill_verbalize(seqs, label_ids).cpu().numpy()
    model.set_variants(0, 0)
    assert np.abs(gen - ref).max() < 4e-2


def test_llama_errors(tiny_llama):
    from llamarec_amd._lib import LlamaRecError

    cfg, sd, model = tiny_llama
    with pytest.raises(LlamaRecError, match="max_positions"):
        model.prefill_verbalize([np.ones(1537, np.int32)], list(range(20)))
    with pytest.raises(ValueError, match="empty"):
        model.prefill_verbalize([np.ones(3, np.int32), np.zeros(0, np.int32)], list(range(20)))
    # token ids outside the vocabulary are clamped to id 0 rather than faulting
    out = model.prefill_verbalize([np.array([1, 99999, -5, 7])], list(range(20)))
    assert torch.isfinite(out).all()
    # a label word outside the vocabulary gives a NaN score for that class only (no wild read of lm_head)
    bad = model.prefill_verbalize([np.array([1, 5, 7])], [3, cfg["vocab_size"], -1, 4])
    assert torch.isnan(bad[0, 1]) and torch.isnan(bad[0, 2]) and torch.isfinite(bad[0, [0, 3]]).all()


def test_rank_classes_ties_and_histogram_absent_labels():
    from llamarec_amd import metrics as M

    s = torch.tensor([[0.5, 0.5, 1.0, -1.0], [0.0, 0.0, 0.0, 0.0]], device="cuda")
    assert M.rank_classes(s).cpu().tolist() == [[2, 0, 1, 3], [0, 1, 2, 3]]
    ranked = torch.tensor([[3, 1, 2], [5, 6, 7]], dtype=torch.int32, device="cuda")
    hist = M.rank_histogram(ranked, torch.tensor([2, 9], device="cuda"))
    assert hist.cpu().tolist() == [0, 0, 1, 1]  # label 9 is not ranked -> last bin
