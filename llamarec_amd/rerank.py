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
        if token_budget == "auto":       # --eval_token_budget: None -> packing.TOKEN_BUDGET, 0 -> the reference's fixed batches
            tb = getattr(args, "eval_token_budget", None)
            token_budget = PK.TOKEN_BUDGET if tb is None else (tb or None)
        self.token_budget = token_budget
        # prompts per prefill in token-budget mode: the reference's batch knob (config.py:98) still applies when the user SET
        # it (a caller's batch_size argument or an explicit --test_batch_size); the template's default (16 / 32) does not
        # -- it would cut a 32 768-token step down to a third
        self.max_prompts = batch_size or (self.batch_size if getattr(args, "test_batch_size_explicit", False) else None)
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
        if isinstance(self.items, LazyEvalItems):
            # shard FIRST (by the title-cache estimate of every prompt's tokens), tokenise only this rank's users, in a
            # producer thread that stays ahead of the GPU loop
            lo, hi = D.broadcast_shard_edges(PK.shard_by_tokens(self.items.estimate_lengths(), world), self.model.device)[rank]
            self._shard = (lo, hi)
            if self.token_budget:
                stream = stream_token_budget_batches(self.items, lo, hi, self.token_budget, self.max_text_len,
                                                     self.max_prompts)
            else:
                stream = (P.eval_pack(self.items.build(i, min(i + self.batch_size, hi)), self.max_text_len)
                          for i in range(lo, hi, self.batch_size))
            for seqs, labels in stream:
                scores = self.model.prefill_verbalize(seqs, self.verbalizer.label_token_ids)
                M.rank_histogram(M.rank_classes(scores), torch.from_numpy(labels).to(self.model.device), hist)
            return self._finish(hist, t0)
        # contiguous shards balanced by prompt TOKENS, then token-budget batches inside the shard
        lens = np.array([min(len(it["input_ids"]), self.max_text_len) for it in self.items], dtype=np.int64)
        lo, hi = D.broadcast_shard_edges(PK.shard_by_tokens(lens, world), self.model.device)[rank]
        self._shard = (lo, hi)
        mine = self.items[lo:hi]
        if self.token_budget:
            # the prompts' common template prefix is run once per batch (llm.prefill_verbalize): budget the rows it leaves
            from .llm import common_prefix_len, pack_prompts

            shared = common_prefix_len(*pack_prompts([np.asarray(it["input_ids"][-self.max_text_len:]) for it in mine])) \
                if len(mine) > 1 else 0
            batches = PK.token_budget_steps(lens[lo:hi], max(self.token_budget, int(lens.max()) if len(lens) else 1),
                                            shared_prefix=shared, max_prompts=self.max_prompts)
        else:
            batches = [np.arange(i, min(i + self.batch_size, len(mine))) for i in range(0, len(mine), self.batch_size)]
        for idx in batches:
            seqs, labels = P.eval_pack([mine[int(i)] for i in idx], self.max_text_len)
            scores = self.model.prefill_verbalize(seqs, self.verbalizer.label_token_ids)
            ranked = M.rank_classes(scores)
            M.rank_histogram(ranked, torch.from_numpy(labels).to(self.model.device), hist)
        return self._finish(hist, t0)

    def _finish(self, hist, t0):
        from . import dist as D

        # every rank cut its own shard (the lazy path from its own least-squares estimate of the prompt lengths): the
        # shards must tile the items exactly and the all-reduced histogram must count every user once, or a rounding
        # difference between nodes would silently drop or double-count users
        lo, hi = self._shard          # set by predict() on every path (no default: a path that forgot it must fail here)
        cover = torch.tensor([hi - lo], dtype=torch.int64, device=hist.device)
        D.all_reduce_sum_(cover)
        D.all_reduce_sum_(hist)
        if int(cover.item()) != len(self.items) or int(hist.sum().item()) != len(self.items):
            raise RuntimeError(f"rerank shards cover {int(cover.item())} and the rank histogram counts {int(hist.sum().item())} "
                               f"of {len(self.items)} users: the ranks derived different shard edges")
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


class LazyEvalItems:
    """The evaluation prompts of LLMValidDataset / LLMTestDataset (dataloader/llm.py:286-387) built ON DEMAND.

    The reference (and round 2's build_test_items) tokenises every user's prompt serially on every rank before the first
    prefill: 41 tokenizer calls per user, nothing overlapping the GPU -- at 146 users/s per GPU x 8 that, not the prefill,
    bounds train_ranker.py on real data. Here
      * `estimate_lengths()` prices every user's prompt from per-ITEM title token counts (prompt.TitleCache: one
        tokenizer call per catalog item), so the data-parallel shards are cut BEFORE any prompt is tokenised;
      * `build(lo, hi)` materialises users [lo, hi) only: prompt strings from the title cache, then ONE batched
        tokenizer call (prompt.tokenize_prompts);
      * LLMEvaluator.predict runs `build` for its own shard chunk by chunk in a producer thread feeding a bounded queue
        ahead of the GPU loop.
    Items are exactly prompt.seq_to_token_ids' (tests/test_host_logic.py compares them sample by sample)."""

    def __init__(self, dataset, retrieved, tokenizer, args=None, split="test", prompter=None):
        assert split in ("val", "test")
        self.dataset, self.tokenizer, self.split = dataset, tokenizer, split
        self.users = list(retrieved[f"{split}_users"])
        self.cands = retrieved[f"{split}_candidates"]
        self.max_hist = getattr(args, "llm_max_history", P.LLM_MAX_HISTORY)
        self.max_title_len = getattr(args, "llm_max_title_len", P.LLM_MAX_TITLE_LEN)
        self.max_text_len = getattr(args, "llm_max_text_len", P.LLM_MAX_TEXT_LEN)
        self.system_template = getattr(args, "llm_system_template", None) or P.DEFAULT_SYSTEM_TEMPLATE
        self.input_template = getattr(args, "llm_input_template", None) or P.DEFAULT_INPUT_TEMPLATE
        self.prompter = prompter or P.Prompter()
        self.titles = P.TitleCache(dataset["meta"], tokenizer, self.max_title_len)

    def __len__(self):
        return len(self.users)

    def _sample(self, i):
        user, cands = self.users[i], list(self.cands[i])
        if self.split == "test":   # dataloader/llm.py:368-375
            seq = (list(self.dataset["train"][user]) + list(self.dataset["val"][user]))[-self.max_hist:]
            answer = self.dataset["test"][user][0]
        else:                      # dataloader/llm.py:315-322
            seq = list(self.dataset["train"][user])[-self.max_hist:]
            answer = self.dataset["val"][user][0]
        assert answer in cands
        return seq, cands, answer

    def estimate_lengths(self, calibrate=32):
        """Prompt tokens of every user WITHOUT tokenising the prompts: a + b * lines + c * (sum of the cached title token
        counts), with (a, b, c) fitted on `calibrate` users spread over the split whose prompts are really tokenised
        (template text and per-line markers tokenise differently under different tokenizers). Every rank computes the
        same numbers (same users, same arithmetic), so the shards cut from them agree across ranks."""
        n = len(self)
        feats = np.empty((n, 3), np.float64)
        for i in range(n):
            seq, cands, _ = self._sample(i)
            feats[i] = (1.0, len(seq) + len(cands),
                        self.titles.estimate_prompt_tokens(seq, cands, overhead=0, per_line=0))
        if n == 0:
            return np.zeros(0, np.int64)
        pick = np.unique(np.linspace(0, n - 1, min(n, calibrate)).astype(np.int64))
        args_long = self.max_text_len
        self.max_text_len = 1 << 30                       # fit on untruncated lengths
        try:
            real = np.array([len(self.build(int(i), int(i) + 1)[0]["input_ids"]) for i in pick], np.float64)
        finally:
            self.max_text_len = args_long
        coef, *_ = np.linalg.lstsq(feats[pick], real, rcond=None)
        est = np.rint(feats @ coef).astype(np.int64)
        return np.clip(est, 1, self.max_text_len)

    def build(self, lo, hi):
        prompts, labels = [], []
        for i in range(lo, hi):
            seq, cands, answer = self._sample(i)
            text, lab = P.eval_prompt_text(seq, cands, answer, self.dataset["meta"], self.tokenizer, self.prompter,
                                           self.max_title_len, self.system_template, self.input_template, self.titles)
            prompts.append(text)
            labels.append(lab)
        ids = P.tokenize_prompts(prompts, self.tokenizer, self.max_text_len)
        return [{"input_ids": x, "attention_mask": [1] * len(x), "labels": l} for x, l in zip(ids, labels)]


def stream_token_budget_batches(items, lo, hi, token_budget, max_text_len, max_prompts=None, chunk=512, depth=4):
    """Generator of (seqs, labels) batches over users [lo, hi) of a LazyEvalItems, produced by a background thread
    `depth` batches ahead of the consumer: each chunk of users is built (tokenised), appended to the pending pool, and
    the pool's full token-budget steps (packing.token_budget_steps; the last, possibly short step stays pending until
    the final chunk) are queued. The template prefix every prompt shares is measured on the first chunk."""
    import queue
    import threading

    from .llm import common_prefix_len, pack_prompts

    q = queue.Queue(maxsize=depth)
    END = object()
    stop = threading.Event()   # set when the consumer leaves early (closed generator, error downstream)

    def put(x):
        while not stop.is_set():
            try:
                q.put(x, timeout=0.25)
                return True
            except queue.Full:
                pass
        return False

    def produce():
        try:
            pending = []
            for c0 in range(lo, hi, chunk):
                pending += items.build(c0, min(c0 + chunk, hi))
                last = c0 + chunk >= hi
                seqs_all = [np.asarray(it["input_ids"][-max_text_len:], dtype=np.int32) for it in pending]
                # the budgeted prefix is THIS pool's (measured once on the first chunk, a later pool with a
                # left-truncated prompt -- which has lost the template -- was budgeted with a prefix its batches do not have
                # and ran up to (n - 1) * P rows over the token budget); prefill_verbalize shares the prefix of each
                # emitted batch, a subset of the pool, so the pool's common prefix never over-promises
                shared = common_prefix_len(*pack_prompts(seqs_all)) if len(seqs_all) > 1 else 0
                lens = np.array([len(s) for s in seqs_all], dtype=np.int64)
                sp = shared if len(lens) and shared < int(lens.min()) else 0
                steps = PK.token_budget_steps(lens, max(token_budget, int(lens.max())), shared_prefix=sp,
                                              max_prompts=max_prompts)
                emit = steps if last else steps[:-1]
                taken = set()
                for idx in emit:
                    if not put(P.eval_pack([pending[int(i)] for i in idx], max_text_len)):
                        return
                    taken.update(int(i) for i in idx)
                pending = [it for i, it in enumerate(pending) if i not in taken]
            put(END)
        except BaseException as e:  # surface producer failures in the consumer
            put(e)

    t = threading.Thread(target=produce, daemon=True, name="llamarec-tokenize")
    t.start()
    try:
        while True:
            b = q.get()
            if b is END:
                break
            if isinstance(b, BaseException):
                raise b
            yield b
    finally:   # also reached when the consumer abandons the generator: release a producer blocked on a full queue
        stop.set()
        t.join()


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
