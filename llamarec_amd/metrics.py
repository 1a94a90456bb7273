"""Ranking metrics -- host-side mirror of trainer/utils.py:6-90 of the reference.

The GPU does the integer part (where does each row's label sit in its ranked list ->
histogram of ranks, lr_rank_histogram); Recall/MRR/NDCG at any k are float64 functions of that
histogram (lr_metrics_from_histogram), so data-parallel ranks all-reduce one int64 vector and
every rank count is exact.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import check, lib, stream_ptr


def rank_histogram(ranked: torch.Tensor, labels: torch.Tensor, hist: torch.Tensor | None = None) -> torch.Tensor:
    """hist[p] += #rows whose label is at 0-based rank p of `ranked` [B,Kmax] (int32, best first);
    hist[Kmax] += rows whose label is absent. Returns the int64 [Kmax+1] device tensor."""
    if ranked.dtype != torch.int32:
        ranked = ranked.to(torch.int32)
    ranked = ranked.contiguous()
    labels = labels.reshape(-1).to(device=ranked.device, dtype=torch.int64).contiguous()
    B, Kmax = ranked.shape
    if labels.numel() != B:
        raise ValueError("labels must have one entry per ranked row")
    if hist is None:
        hist = torch.zeros(Kmax + 1, dtype=torch.int64, device=ranked.device)
    with torch.cuda.device(ranked.device):
        check(lib().lr_rank_histogram(ranked.data_ptr(), Kmax, labels.data_ptr(), B, hist.data_ptr(),
                                      stream_ptr()), "lr_rank_histogram")
    return hist


def rank_classes(scores: torch.Tensor, items: torch.Tensor | None = None, out=None) -> torch.Tensor:
    """`(-scores).argsort(dim=1)` for C <= 64 classes with the tie rule lower id first
    (trainer/utils.py:55 on the reranker's [N,20] verbalizer scores). With `items` [B,C] (int32
    candidate ids in class order) the result holds the candidates' item ids in reranked order."""
    scores = scores.to(torch.float32).contiguous()
    B, Cn = scores.shape
    if out is None:
        out = torch.empty((B, Cn), dtype=torch.int32, device=scores.device)
    ip = None
    if items is not None:
        if items.dtype != torch.int32 or not items.is_contiguous() or tuple(items.shape) != (B, Cn):
            raise ValueError("items must be a contiguous int32 [B, C] tensor")
        ip = items.data_ptr()
    with torch.cuda.device(scores.device):
        check(lib().lr_rank_classes(scores.data_ptr(), B, Cn, ip, out.data_ptr(), stream_ptr()), "lr_rank_classes")
    return out


def metric_sums_from_histogram(hist, ks) -> np.ndarray:
    """float64 [nk,3] numerators (Recall, MRR, NDCG) for k in ks, from a host int64 histogram."""
    h = np.ascontiguousarray(np.asarray(hist.cpu() if isinstance(hist, torch.Tensor) else hist), dtype=np.int64)
    ksa = np.ascontiguousarray(list(ks), dtype=np.int32)
    sums = np.zeros((len(ksa), 3), np.float64)
    check(lib().lr_metrics_from_histogram(h.ctypes.data, len(h) - 1, ksa.ctypes.data, len(ksa),
                                          sums.ctypes.data), "lr_metrics_from_histogram")
    return sums


def metrics_from_histogram(hist, ks, denom=None) -> dict:
    """Metric dict with the reference's key names and ordering (k descending, trainer/utils.py:60-88).
    denom defaults to the number of rows counted in the histogram (batch mean, :68,76,87)."""
    h = np.asarray(hist.cpu() if isinstance(hist, torch.Tensor) else hist, dtype=np.int64)
    n = int(h.sum()) if denom is None else denom
    order = sorted(ks, reverse=True)
    sums = metric_sums_from_histogram(h, order)
    out = {}
    for j, k in enumerate(order):
        out["Recall@%d" % k] = float(sums[j, 0] / n) if n else 0.0
        out["MRR@%d" % k] = float(sums[j, 1] / n) if n else 0.0
        out["NDCG@%d" % k] = float(sums[j, 2] / n) if n else 0.0
    return out


def absolute_recall_mrr_ndcg_for_ks(scores, labels, ks, num_classes=None, preprocessed=False):
    """Mirror of trainer/utils.py:43-90. `scores` is either pre-ranked ids [B,K>=max(ks)]
    (preprocessed=True, the path `absolute_metrics_batch_wrapper` uses at trainer/lru.py:144-157) or
    raw scores over C <= 64 classes (the reranker's [N,20], trainer/llm.py:63-72). Raw scores over
    the whole item table are ranked by LRURec.retrieve_topk instead (the matrix never exists)."""
    if preprocessed:
        ranked = scores
    else:
        if scores.shape[1] > 64:
            raise NotImplementedError(
                "raw score matrices wider than 64 are not ranked here: use LRURec.retrieve_topk(ids, max(ks)) "
                "and pass the ranked ids with preprocessed=True")
        ranked = rank_classes(scores)
    if ranked.shape[1] < max(ks):
        raise ValueError("ranked lists are shorter than max(ks)")
    return metrics_from_histogram(rank_histogram(ranked, labels), ks)


def absolute_metrics_batch_wrapper(scores, labels, ks, num_classes=None, preprocessed=False, batch_size=10000):
    """Mirror of trainer/utils.py:6-40: size-weighted mean of chunk means == the global mean, so
    one histogram over all rows gives the same numbers (to float64 rounding)."""
    if labels.numel() == 0:
        return {}
    return absolute_recall_mrr_ndcg_for_ks(scores, labels, ks, num_classes, preprocessed)
