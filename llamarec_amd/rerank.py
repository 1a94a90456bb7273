"""Ranker evaluation -- mirror of the reference's LLMTrainer.test (trainer/llm.py:165-189) and
compute_metrics_for_ks (:63-72), without HF Trainer: prompts are packed (no padding), one HIP
prefill per batch returns the [B,20] verbalizer scores, and only an int64 rank histogram leaves
the GPU (the reference gathers [N,32000] fp32 logits to the host first).
"""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch

from . import metrics as M
from . import packing as PK
from . import prompt as P

RERANK_METRIC_KS = [1, 5, 10]  # config.py:140-143


def merge_overall_metrics(average_metrics: dict, test_retrieval: dict) -> dict:
    """trainer/llm.py:174-184: overall = (subset * n_ret + non_retrieval * (n_all - n_ret)) / n_all for
    the keys present in both (the rerank ks)."""
    original_size, retrieval_size = test_retrieval["original_size"], test_retrieval["retrieval_size"]
    overall = {}
    for key in test_retrieval["non_retrieval_metrics"].keys():
        if "test_" + key in average_metrics:
            overall["test_" + key] = (
                average_metrics["test_" + key] * retrieval_size
                + test_retrieval["non_retrieval_metrics"][key] * (original_size - retrieval_size)
            ) / original_size
    return overall


class LLMEvaluator:
    """test_items: list of dicts {"input_ids", "attention_mask", "labels"} as produced by
    prompt.seq_to_token_ids (== LLMTestDataset.__getitem__, dataloader/llm.py:368-387)."""

    def __init__(self, args, model, test_items, verbalizer, export_root=None, batch_size=None, token_budget="auto"):
        """batch_size: the reference's prompts-per-batch cap (config.py:98); token_budget: prompt tokens per prefill
        ("auto" = packing.TOKEN_BUDGET, None = plain fixed-size batches in dataset order like the reference's loader).
        Scores do not depend on the batching (unpadded execution), so this only changes speed."""
        self.args, self.model, self.items, self.verbalizer = args, model, test_items, verbalizer
        self.export_root = export_root
        self.ks = list(getattr(args, "rerank_metric_ks", RERANK_METRIC_KS))
        self.batch_size = batch_size or getattr(args, "test_batch_size", 16)
        self.token_budget = PK.TOKEN_BUDGET if token_budget == "auto" else token_budget
        self.max_text_len = getattr(args, "llm_max_text_len", P.LLM_MAX_TEXT_LEN)

    def predict(self):
        """Under torch.distributed every rank scores its contiguous share of the items and the int64 rank histograms
        are summed with ONE all-reduce; the reference gathers [N, 32000] fp32 logits instead
        (trainer/llm.py:122,127)."""
        from . import dist as D

        t0 = time.time()
        ncls = self.verbalizer.num_classes
        hist = torch.zeros(ncls + 1, dtype=torch.int64, device=self.model.device)
        rank, world, _ = D.env_world()
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            rank, world = 0, 1
        # contiguous shards balanced by prompt TOKENS, then token-budget batches inside the shard
        lens = np.array([min(len(it["input_ids"]), self.max_text_len) for it in self.items], dtype=np.int64)
        lo, hi = PK.shard_by_tokens(lens, world)[rank]
        mine = self.items[lo:hi]
        if self.token_budget:
            # the prompts' common template prefix is run once per batch (llm.prefill_verbalize): budget the rows it leaves
            from .llm import common_prefix_len, pack_prompts

            shared = common_prefix_len(*pack_prompts([np.asarray(it["input_ids"][-self.max_text_len:]) for it in mine])) \
                if len(mine) > 1 else 0
            batches = PK.token_budget_steps(lens[lo:hi], max(self.token_budget, int(lens.max()) if len(lens) else 1),
                                            shared_prefix=shared)
        else:
            batches = [np.arange(i, min(i + self.batch_size, len(mine))) for i in range(0, len(mine), self.batch_size)]
        for idx in batches:
            seqs, labels = P.eval_pack([mine[int(i)] for i in idx], self.max_text_len)
            scores = self.model.prefill_verbalize(seqs, self.verbalizer.label_token_ids)
            ranked = M.rank_classes(scores)
            M.rank_histogram(ranked, torch.from_numpy(labels).to(self.model.device), hist)
        D.all_reduce_sum_(hist)
        m = M.metrics_from_histogram(hist, self.ks) if len(self.items) else {}
        out = {"test_" + k: v for k, v in m.items()}
        out["test_loss"] = -1.0  # model/llm.py:128-129: eval loss is the constant -1
        out["test_runtime"] = time.time() - t0
        out["test_samples_per_second"] = len(self.items) / max(out["test_runtime"], 1e-9)
        return out

    def test(self, test_retrieval):
        average_metrics = self.predict()
        overall = merge_overall_metrics(average_metrics, test_retrieval)
        from . import dist as D

        if self.export_root and D.env_world()[0] == 0:            # every rank holds the same sums: rank 0 writes
            os.makedirs(self.export_root, exist_ok=True)
            with open(os.path.join(self.export_root, "subset_metrics.json"), "w") as f:
                json.dump(average_metrics, f, indent=4)
            with open(os.path.join(self.export_root, "overall_metrics.json"), "w") as f:
                json.dump(overall, f, indent=4)
        self.overall_metrics = overall
        return average_metrics


def build_val_items(dataset, retrieved, tokenizer, args=None, prompter=None):
    """LLMValidDataset (dataloader/llm.py:286-334): history = train[-llm_max_history:], candidates = the
    retriever's ordered top-20 for the validation answer."""
    max_hist = getattr(args, "llm_max_history", P.LLM_MAX_HISTORY)
    kw = dict(max_title_len=getattr(args, "llm_max_title_len", P.LLM_MAX_TITLE_LEN),
              max_text_len=getattr(args, "llm_max_text_len", P.LLM_MAX_TEXT_LEN),
              system_template=getattr(args, "llm_system_template", None) or P.DEFAULT_SYSTEM_TEMPLATE,
              input_template=getattr(args, "llm_input_template", None) or P.DEFAULT_INPUT_TEMPLATE)
    prompter = prompter or P.Prompter()
    items = []
    for user, cands in zip(retrieved["val_users"], retrieved["val_candidates"]):
        seq = list(dataset["train"][user])[-max_hist:]
        answer = dataset["val"][user][0]
        assert answer in cands  # dataloader/llm.py:322
        items.append(P.seq_to_token_ids(seq, cands, answer, dataset["meta"], tokenizer, prompter, **kw))
    return items


def build_test_items(dataset, retrieved, tokenizer, args=None, prompter=None):
    """LLMTestDataset (dataloader/llm.py:337-387): history = (train + val)[-llm_max_history:],
    candidates = the retriever's ordered top-20 (not shuffled), answer = test item."""
    max_hist = getattr(args, "llm_max_history", P.LLM_MAX_HISTORY)
    kw = dict(max_title_len=getattr(args, "llm_max_title_len", P.LLM_MAX_TITLE_LEN),
              max_text_len=getattr(args, "llm_max_text_len", P.LLM_MAX_TEXT_LEN),
              system_template=getattr(args, "llm_system_template", None) or P.DEFAULT_SYSTEM_TEMPLATE,
              input_template=getattr(args, "llm_input_template", None) or P.DEFAULT_INPUT_TEMPLATE)
    prompter = prompter or P.Prompter()
    items = []
    for user, cands in zip(retrieved["test_users"], retrieved["test_candidates"]):
        seq = (list(dataset["train"][user]) + list(dataset["val"][user]))[-max_hist:]
        answer = dataset["test"][user][0]
        assert answer in cands  # dataloader/llm.py:375
        items.append(P.seq_to_token_ids(seq, cands, answer, dataset["meta"], tokenizer, prompter, **kw))
    return items
