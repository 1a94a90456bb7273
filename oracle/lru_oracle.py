"""ctypes front-end of oracle/lr_oracle.c (stage-1 CPU oracle). TEST INFRASTRUCTURE ONLY.

Follows model/lru.py:38-175, trainer/lru.py:30-42,82-84,113-115 and trainer/utils.py:43-90 of the
reference; see the header of lr_oracle.c for the line-by-line map.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from llamarec_amd._abi import LrLruWeightsDesc, lru_desc_from_state_dict

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, "lr_oracle.c"),
           os.path.join(_HERE, "..", "llamarec_amd", "csrc", "lr_math.h")]
    stale = not os.path.exists(_SO) or any(
        os.path.exists(s) and os.path.getmtime(_SO) < os.path.getmtime(s) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "-s"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        # LR_ORACLE_LIB: another build of the same oracle (the sanitizer build of tools/sanitize_cpu.sh)
        _lib = C.CDLL(os.environ.get("LR_ORACLE_LIB") or build())
        _lib.lro_num_threads.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class LruOracle:
    """CPU restatement of LRURec scoring for one state_dict."""

    def __init__(self, state_dict):
        self.desc, self._keep = lru_desc_from_state_dict(state_dict)
        self.num_items = self.desc.num_items

    def encode_last(self, ids: np.ndarray) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        B, L = ids.shape
        q = np.empty((B, 64), np.float32)
        rc = lib().lro_lru_encode_last(C.byref(self.desc), _p(ids, C.c_int64), B, L, _p(q, C.c_float))
        assert rc == 0
        return q

    def scores_last(self, ids: np.ndarray, exclude_history: bool) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        B, L = ids.shape
        s = np.empty((B, self.num_items + 1), np.float32)
        rc = lib().lro_lru_scores_last(
            C.byref(self.desc), _p(ids, C.c_int64), B, L, int(exclude_history), _p(s, C.c_float)
        )
        assert rc == 0
        return s

    def retrieve_topk(self, ids: np.ndarray, K: int, exclude_history: bool):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        B, L = ids.shape
        idx = np.empty((B, K), np.int32)
        sc = np.empty((B, K), np.float32)
        rc = lib().lro_lru_retrieve_topk(
            C.byref(self.desc), _p(ids, C.c_int64), B, L, K, int(exclude_history),
            _p(idx, C.c_int32), _p(sc, C.c_float),
        )
        assert rc == 0
        return idx, sc


def topk(scores: np.ndarray, K: int):
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    B, n = scores.shape
    idx = np.empty((B, K), np.int32)
    sc = np.empty((B, K), np.float32)
    rc = lib().lro_topk(_p(scores, C.c_float), B, n, K, _p(idx, C.c_int32), _p(sc, C.c_float))
    assert rc == 0
    return idx, sc


def rank_metric_sums(ranked: np.ndarray, labels: np.ndarray, ks) -> np.ndarray:
    """Numerators [nk,3] = (Recall, MRR, NDCG)@k summed over rows."""
    ranked = np.ascontiguousarray(ranked, dtype=np.int32)
    labels = np.ascontiguousarray(labels, dtype=np.int64).reshape(-1)
    ksa = np.ascontiguousarray(ks, dtype=np.int32)
    sums = np.zeros((len(ksa), 3), np.float64)
    rc = lib().lro_rank_metrics(
        _p(ranked, C.c_int32), ranked.shape[1], _p(labels, C.c_int64), ranked.shape[0],
        _p(ksa, C.c_int32), len(ksa), _p(sums, C.c_double),
    )
    assert rc == 0
    return sums


def num_threads() -> int:
    return int(lib().lro_num_threads())
