"""Edge-case calls into the host-only C entry points and the C oracle (CPU, no GPU): what tools/sanitize_cpu.sh drives under
AddressSanitizer / UBSan -- empty and ragged inputs, maximum ranks, catalogs smaller than K, all-pad histories -- and a
plain correctness test otherwise."""
import ctypes as C

import numpy as np

from llamarec_amd import _lib
from llamarec_amd.lru import init_lru_state_dict
from oracle import lru_oracle as O


def test_common_prefix_len_edges():
    l = _lib.lib()

    def cpl(seqs):
        ids = np.concatenate([np.asarray(s, np.int32) for s in seqs]) if seqs else np.zeros(0, np.int32)
        cu = np.zeros(len(seqs) + 1, np.int32)
        cu[1:] = np.cumsum([len(s) for s in seqs])
        return l.lr_common_prefix_len(ids.ctypes.data, cu.ctypes.data, len(seqs))

    assert cpl([[1, 2, 3]]) == 0                                   # one prompt: nothing to share
    assert cpl([[1, 2, 3], [1, 2, 3]]) == 2                        # capped: every prompt keeps a token of its own
    assert cpl([[1, 2, 3, 4], [1, 2, 9]]) == 2
    assert cpl([[1], [1, 2]]) == 0                                 # shortest prompt has one token
    assert cpl([[5, 6, 7], [4, 6, 7]]) == 0
    assert cpl([[1, 2, 3, 4, 5]] * 7 + [[1, 2, 3, 9, 9, 9]]) == 3
    assert l.lr_common_prefix_len(None, None, 3) == 0


def test_metrics_from_histogram_edges():
    l = _lib.lib()
    for kmax in (1, 20, 50):
        hist = np.arange(kmax + 1, dtype=np.int64)
        ks = np.array([1, kmax], np.int32)
        sums = np.zeros((2, 3))
        assert l.lr_metrics_from_histogram(hist.ctypes.data, kmax, ks.ctypes.data, 2, sums.ctypes.data) == 0
        assert sums[1, 0] == hist[:kmax].sum() and sums[0, 0] == hist[0]
    sums = np.zeros((1, 3))
    ks = np.array([0], np.int32)
    assert l.lr_metrics_from_histogram(hist.ctypes.data, 50, ks.ctypes.data, 1, sums.ctypes.data) != 0   # k < 1
    assert l.lr_metrics_from_histogram(None, 50, ks.ctypes.data, 1, sums.ctypes.data) != 0


def test_pack_and_oracle_edges():
    """lr_lru_pack through the oracle's own packing call, then the oracle on ragged / empty / tiny inputs."""
    for V, K in ((3, 50), (31, 20), (32, 5), (500, 50)):
        sd = init_lru_state_dict(V, seed=V)
        orc = O.LruOracle(sd)
        L = 7
        ids = np.zeros((5, L), np.int64)
        ids[1, -1] = 1
        ids[2, -3:] = [1, 2, 3]
        ids[3, :] = (np.arange(L) % V) + 1
        ids[4, -2:] = [V, V]
        top, scores = orc.retrieve_topk(ids, K, True)
        assert top.shape == (5, K)
        live = top[top > 0]
        assert live.min() >= 1 and live.max() <= V
        top2, _ = orc.retrieve_topk(ids, K, False)
        assert top2.shape == (5, K)
        assert orc.retrieve_topk(ids[:0], K, True)[0].shape == (0, K)
    # metric sums on ranked lists with labels absent, first, last
    ranked = np.array([[3, 1, 2], [1, 2, 3], [2, 3, 1]], np.int32)
    s = O.rank_metric_sums(ranked, np.array([9, 1, 1], np.int64), [3, 1])
    assert s[0, 0] == 2 and s[1, 0] == 1


def test_pack_helpers_round_trip():
    l = _lib.lib()
    rng = np.random.default_rng(0)
    inter, hidden = 32, 8
    gate = rng.integers(0, 65535, size=(inter, hidden)).astype(np.uint16)
    up = rng.integers(0, 65535, size=(inter, hidden)).astype(np.uint16)
    out = np.zeros((2 * inter, hidden), np.uint16)
    assert l.lr_llama_pack_gate_up(gate.ctypes.data, up.ctypes.data, inter, hidden, out.ctypes.data) == 0
    assert np.array_equal(out[:16], gate[:16]) and np.array_equal(out[16:32], up[:16]) and np.array_equal(out[32:48], gate[16:])
    assert l.lr_llama_pack_gate_up(gate.ctypes.data, up.ctypes.data, 24, hidden, out.ctypes.data) != 0   # not a multiple of 16
    nh, nkv, hd = 2, 1, 8
    q = rng.integers(0, 65535, size=(nh * hd, hidden)).astype(np.uint16)
    k = rng.integers(0, 65535, size=(nkv * hd, hidden)).astype(np.uint16)
    v = rng.integers(0, 65535, size=(nkv * hd, hidden)).astype(np.uint16)
    out = np.zeros(((nh + 2 * nkv) * hd, hidden), np.uint16)
    assert l.lr_llama_pack_qkv(q.ctypes.data, k.ctypes.data, v.ctypes.data, nh, nkv, hd, hidden, out.ctypes.data) == 0
    assert np.array_equal(out[0], q[0]) and np.array_equal(out[1], q[hd // 2]) and np.array_equal(out[-nkv * hd:], v)
    code = (C.c_float * 256)()
    assert l.lr_nf4_dynamic_map(code) == 0 and code[255] == 1.0


def test_sanitizer_job_is_clean():
    """tools/sanitize_cpu.sh: ASan + UBSan builds of the oracle and of the library's host side, driven by the CPU tests that
    reach them (this file's other tests included), must finish without a finding."""
    import os
    import subprocess

    import pytest

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("LLAMAREC_LIB", "").endswith("san/libllamarec_mi355x.so"):
        pytest.skip("already inside the sanitizer job")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("ROCm clang (sanitizer runtime) not present")
    p = subprocess.run(["bash", os.path.join(repo, "tools", "sanitize_cpu.sh")], capture_output=True, text=True, timeout=900)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0 and "passed" in tail and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
