"""Retriever evaluation and candidate generation -- mirror of the reference's
LRUTrainer.calculate_metrics / generate_candidates (trainer/lru.py:30-42,44-175) and
BaseTrainer.validate / test (trainer/base.py:136-188), driven by the fused HIP retrieve
(encode -> item GEMM -> history mask -> ordered top-50) instead of `model(seqs)[:, -1, :]`, L
index_put_ launches and a per-user argsort loop.

Averaging quirks are preserved: `test()` reports the unweighted mean of
per-batch means; `generate_candidates()` reports per-user sums divided by args.num_users;
retrieval / non-retrieval metrics are means over their user subsets.
"""
from __future__ import annotations

import json
import os
import pickle

import numpy as np
import torch

from . import metrics as M

METRIC_KS = [1, 5, 10, 20, 50]  # config.py:136-139


def _positions(ranked: np.ndarray, labels: np.ndarray) -> np.ndarray:
    """0-based rank of each row's label in its ranked list, or K if absent (integer host work)."""
    hit = ranked == labels.reshape(-1, 1)
    return np.where(hit.any(1), hit.argmax(1), ranked.shape[1])


def _metrics_from_positions(pos, ks, kmax, denom):
    hist = np.bincount(pos, minlength=kmax + 1).astype(np.int64)
    return M.metrics_from_histogram(hist, ks, denom=denom)


class LRUEvaluator:
    def __init__(self, args, model, val_loader, test_loader, export_root=None):
        self.args, self.model = args, model
        self.val_loader, self.test_loader = val_loader, test_loader
        self.export_root = export_root
        self.metric_ks = list(getattr(args, "metric_ks", METRIC_KS))
        self.kmax = max(self.metric_ks)
        self.num_candidates = getattr(args, "llm_negative_sample_size", 19) + 1

    # trainer/lru.py:30-42
    def calculate_metrics(self, batch, exclude_history=True):
        seqs, labels = batch
        ranked, _ = self.model.retrieve_topk(seqs, self.kmax, exclude_history=exclude_history)
        labels = torch.as_tensor(np.asarray(labels)).reshape(-1)
        return M.absolute_recall_mrr_ndcg_for_ks(ranked, labels.to(ranked.device), self.metric_ks, preprocessed=True)

    def _mean_of_batch_means(self, loader, exclude_history):
        sums, n = {}, 0
        for batch in loader:
            for k, v in self.calculate_metrics(batch, exclude_history).items():
                sums[k] = sums.get(k, 0.0) + v
            n += 1
        return {k: v / n for k, v in sums.items()} if n else {}

    # trainer/base.py:136-154 (validation does not mask the history: "faster validation")
    def validate(self):
        return self._mean_of_batch_means(self.val_loader, exclude_history=False)

    # trainer/base.py:156-188
    def test(self, save_name=None):
        avg = self._mean_of_batch_means(self.test_loader, exclude_history=True)
        if self.export_root:
            os.makedirs(self.export_root, exist_ok=True)
            with open(os.path.join(self.export_root, save_name or "test_metrics.json"), "w") as f:
                json.dump(avg, f, indent=4)
        return avg

    def _pass(self, loader):
        ranked_all, labels_all = [], []
        for seqs, labels in loader:
            ranked, _ = self.model.retrieve_topk(seqs, self.kmax, exclude_history=True)
            ranked_all.append(ranked.cpu().numpy())
            labels_all.append(np.asarray(labels).reshape(-1))
        return np.concatenate(ranked_all), np.concatenate(labels_all)

    # trainer/lru.py:44-175
    def generate_candidates(self, retrieved_data_path):
        ks, kmax, nc = self.metric_ks, self.kmax, self.num_candidates
        num_users = self.args.num_users
        v_ranked, v_labels = self._pass(self.val_loader)
        v_pos = _positions(v_ranked, v_labels)
        val_metrics = _metrics_from_positions(v_pos, ks, kmax, num_users)
        v_keep = np.nonzero(v_pos < nc)[0]
        val_users = (v_keep + 1).tolist()  # user_id = running index + 1 (trainer/lru.py:85)
        val_candidates = v_ranked[v_keep, :nc].tolist()

        t_ranked, t_labels = self._pass(self.test_loader)
        t_pos = _positions(t_ranked, t_labels)
        test_metrics = _metrics_from_positions(t_pos, ks, kmax, num_users)
        t_keep = np.nonzero(t_pos < nc)[0]
        t_drop = np.nonzero(t_pos >= nc)[0]
        test_retrieval = {
            "original_size": int(len(t_ranked)),
            "retrieval_size": int(len(t_keep)),
            "original_metrics": test_metrics,
            "retrieval_metrics": _metrics_from_positions(t_pos[t_keep], ks, kmax, len(t_keep)) if len(t_keep) else {},
            "non_retrieval_metrics": _metrics_from_positions(t_pos[t_drop], ks, kmax, len(t_drop)) if len(t_drop) else {},
        }
        out = {
            "val_metrics": val_metrics, "val_users": val_users, "val_candidates": val_candidates,
            "test_probs": t_ranked.tolist(), "test_labels": t_labels.tolist(), "test_metrics": test_metrics,
            "test_users": (t_keep + 1).tolist(), "test_candidates": t_ranked[t_keep, :nc].tolist(),
            "non_test_users": (t_drop + 1).tolist(), "test_retrieval": test_retrieval,
        }
        if retrieved_data_path:
            os.makedirs(os.path.dirname(os.path.abspath(retrieved_data_path)), exist_ok=True)
            with open(retrieved_data_path, "wb") as f:
                pickle.dump(out, f)
        return out
