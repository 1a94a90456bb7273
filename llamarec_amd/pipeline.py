"""Two-stage retrieve-then-rerank scoring step on one GPU, everything device-resident.

One step over a batch of users =
  stage 1  LRURec encode -> item GEMM + history mask + ordered top-50      (trainer/lru.py:105-126)
           candidates = the first 20 of them, in retriever order           (dataloader/llm.py:374-376)
  stage 2  one Llama prefill over the users' prompts (token ids pre-materialised: no tokenizer
           offline) -> verbalizer gather [B,20] -> reranked item ids       (model/llm.py:131,
                                                                            trainer/llm.py:63-72)
  metrics  int64 rank histograms of the label in the retriever's top-50 and in the reranked top-20;
           ranks all-reduce the histograms once at the end (llamarec_amd.dist).
"""
from __future__ import annotations

import numpy as np
import torch

from . import metrics as M
from .dist import all_reduce_sum_

RETRIEVE_K = 50   # max(metric_ks) (config.py:136-143)
NUM_CAND = 20     # llm_negative_sample_size + 1


class TwoStagePipeline:
    def __init__(self, retriever, ranker, label_token_ids, device="cuda:0", shared_prefix=True):
        self.retriever, self.ranker = retriever, ranker
        self.shared_prefix = shared_prefix   # run the prompts' common template prefix once per step (exact)
        self.device = torch.device(device)
        self.label_ids = torch.as_tensor(np.asarray(label_token_ids, dtype=np.int32)).to(self.device)
        assert self.label_ids.numel() == NUM_CAND
        self.hist_retrieve = torch.zeros(RETRIEVE_K + 1, dtype=torch.int64, device=self.device)
        self.hist_rerank = torch.zeros(NUM_CAND + 1, dtype=torch.int64, device=self.device)
        self.users = 0

    def reset(self):
        self.hist_retrieve.zero_()
        self.hist_rerank.zero_()
        self.users = 0

    def step(self, hist_ids, labels, prompt_ids, cu_dev, cu_host, prefix_len=0):
        """hist_ids int64 [B,L] (device), labels int64 [B] (device), prompts packed (device + host cu);
        prefix_len = llm.common_prefix_len of the packed prompts (0: nothing shared / not computed).
        Returns (top50 item ids, reranked top-20 item ids)."""
        top, _ = self.retriever.retrieve_topk(hist_ids, RETRIEVE_K, exclude_history=True)
        M.rank_histogram(top, labels, self.hist_retrieve)
        cands = top[:, :NUM_CAND].contiguous()
        scores = self.ranker.prefill_verbalize_packed(prompt_ids, cu_dev, cu_host, self.label_ids,
                                                      prefix_len=prefix_len if self.shared_prefix else 0)
        reranked = M.rank_classes(scores, cands)
        M.rank_histogram(reranked, labels, self.hist_rerank)
        self.users += hist_ids.shape[0]
        return top, reranked

    def finish(self):
        """All-reduce the histograms over the data-parallel ranks (the job's only collective) and
        return (retrieve metrics, overall rerank metrics, n_users) with the reference's key names.
        Overall@k for k <= 10: non-retrieved users contribute 0 (trainer/llm.py:177-184)."""
        n = torch.tensor([self.users], dtype=torch.int64, device=self.device)
        packed = torch.cat([self.hist_retrieve, self.hist_rerank, n])
        all_reduce_sum_(packed)
        hr = packed[: RETRIEVE_K + 1].cpu().numpy()
        hk = packed[RETRIEVE_K + 1: -1].cpu().numpy()
        total = int(packed[-1].item())
        retr = M.metrics_from_histogram(hr, [1, 5, 10, 20, 50], denom=total)
        rer = M.metrics_from_histogram(hk, [1, 5, 10], denom=total)
        return retr, rer, total
