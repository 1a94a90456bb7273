"""CPU: the C-ABI library is built, loads, and exports every symbol include/*.h declares."""
import ctypes
import os
import re

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "llamarec_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from llamarec_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    l = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(l, s), f"{s} declared in include/llamarec_mi355x.h but not exported"
    assert set(syms) == set(_lib.PROTOTYPES), set(syms) ^ set(_lib.PROTOTYPES)
    assert b"gfx950" in _lib.lib().lr_version()


def test_pack_is_pure_cpu_and_layout_is_consistent(golden_dir):
    """lr_lru_pack runs without a GPU; spot-check the transposes against the state_dict."""
    from llamarec_amd.lru import pack_state_dict

    z = np.load(os.path.join(golden_dir, "lru_v300.npz"))
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    img, v, nb = pack_state_dict(sd)
    assert (v, nb) == (300, 2)
    rows_padded = 320
    assert np.array_equal(img[: 301 * 64].reshape(301, 64), sd["embedding.token.weight"])
    assert not img[301 * 64: rows_padded * 64].any()
    # first block: lambda/gamma derived from params_log, |lambda| < 1
    # bf16 copy of the table + its two norms (the top-K bound pre-pass) sit between the bias and the embedding LayerNorm
    e16 = img[rows_padded * 64 + rows_padded: rows_padded * 64 + rows_padded + rows_padded * 32].view(np.uint16).reshape(rows_padded, 64)
    from llamarec_amd.synth import f32_to_bf16_bits

    # ... stored in MFMA A-fragment order: [tile][step][lane][8] = E[32 tile + (lane & 31)][32 (lane >> 5) + 8 step + j]
    frag = e16.reshape(rows_padded // 32, 4, 64, 8)
    want = np.zeros((rows_padded, 64), np.uint16)
    want[:301] = f32_to_bf16_bits(sd["embedding.token.weight"])
    want = want.reshape(rows_padded // 32, 32, 2, 4, 8)            # [tile][row][half][step][j]
    assert np.array_equal(frag.reshape(-1, 4, 2, 32, 8), want.transpose(0, 3, 2, 1, 4))   # lane = 32 half + row
    stats = img[rows_padded * 64 + rows_padded + rows_padded * 32:][:2]
    assert stats[0] >= np.linalg.norm(sd["embedding.token.weight"].astype(np.float64), axis=1).max() > 0.99 * stats[0]
    assert stats[1] >= np.abs(sd["model.bias"]).max() >= 0.99 * stats[1]
    off = rows_padded * 64 + rows_padded + rows_padded * 32 + 64 + 64 + 64
    lam_re, lam_im, gamma = img[off:off + 128], img[off + 128:off + 256], img[off + 256:off + 384]
    pl = sd["model.lru_blocks.0.lru_layer.params_log"].astype(np.float64)
    lam = np.exp(-np.exp(pl[0]) + 1j * np.exp(pl[1]))
    assert np.allclose(lam_re, lam.real, atol=1e-7) and np.allclose(lam_im, lam.imag, atol=1e-7)
    assert np.allclose(gamma, np.exp(pl[2]), rtol=1e-6)
    in_wt = img[off + 384: off + 384 + 64 * 256].reshape(64, 256)
    w = sd["model.lru_blocks.0.lru_layer.in_proj.weight"]
    assert np.array_equal(in_wt[:, :128], w.real.T) and np.array_equal(in_wt[:, 128:], w.imag.T)


def test_error_reporting_without_gpu():
    from llamarec_amd import _lib

    l = _lib.lib()
    assert l.lr_lru_packed_bytes(10, 0) == 0
    hist = np.zeros(5, np.int64)
    ks = np.array([9], np.int32)
    sums = np.zeros(3)
    rc = l.lr_metrics_from_histogram(hist.ctypes.data, 4, ks.ctypes.data, 1, sums.ctypes.data)
    assert rc == -1 and b"outside" in l.lr_last_error()
    # argument checks of device entry points run before anything touches a GPU
    rc = l.lr_gemm_bf16_nt_residual_rmsnorm(None, None, None, None, 8, 256, 64, 5, None, None, 1e-5, 1, None, None, 0, None)
    assert rc == -1 and b"null pointer" in l.lr_last_error()
    rc = l.lr_gemm_bf16_nt_epi(None, None, None, None, 8, 256, 64, 0, 4, None, None, 0, 0, 0, None, 0, None)
    assert rc == -1 and b"null pointer" in l.lr_last_error()


def test_metrics_from_histogram_matches_oracle(golden_dir):
    from llamarec_amd import metrics as M
    from oracle import lru_oracle as O

    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    ranked, labels = z["ranked"], z["labels"]
    hist = np.zeros(ranked.shape[1] + 1, np.int64)
    for u in range(len(labels)):
        pos = np.nonzero(ranked[u] == labels[u])[0]
        hist[pos[0] if len(pos) else ranked.shape[1]] += 1
    ks = [1, 5, 10, 20, 50]
    assert np.allclose(M.metric_sums_from_histogram(hist, ks), O.rank_metric_sums(ranked, labels, ks), atol=1e-12)
