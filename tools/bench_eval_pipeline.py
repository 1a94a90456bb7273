"""Real-data evaluation path of train_ranker.py at the Beauty shape, with a REAL HF fast tokenizer (tests/local_tokenizer.py;
no Llama tokenizer files exist offline): how fast do prompts reach the GPU?

  python tools/bench_eval_pipeline.py [--users 4000] [--gpu]

  host only (default): users/s of (a) the reference's per-user path -- 41 title round trips + 1 prompt tokenisation per
      user, serial (prompt.seq_to_token_ids = dataloader/llm.py:64-98) -- and (b) rerank.LazyEvalItems.build (per-item title
      cache + one batched tokenizer call per 512 users).
  --gpu: LLMEvaluator.predict over the same users with a random-weight Llama-2-7b: test_samples_per_second with the
      eager list (tokenise everything, then score) against the lazy, streamed items (producer thread ahead of the GPU).
"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def beauty_like(num_users, num_items=12101, seed=0):
    """dataset.pkl-shaped dict at the Beauty shape: histories of ~8.9 items (>= 5), titles of ~12 words."""
    from tests.local_tokenizer import _WORDS

    rng = np.random.default_rng(seed)
    train, val, test = {}, {}, {}
    for u in range(1, num_users + 1):
        n = int(np.clip(rng.geometric(1.0 / 5.0) + 4, 5, 60))
        items = (rng.choice(num_items, size=n, replace=False) + 1).tolist()
        train[u], val[u], test[u] = items[:-2], items[-2:-1], items[-1:]
    meta = {i: " ".join(rng.choice(_WORDS, size=int(rng.integers(6, 18)))) for i in range(1, num_items + 1)}
    cands = []
    for u in range(1, num_users + 1):
        c = (rng.choice(num_items, size=21, replace=False) + 1).tolist()
        c = [x for x in c if x != test[u][0]][:19] + [test[u][0]]
        cands.append(c)
    ds = {"train": train, "val": val, "test": test, "meta": meta, "umap": {u: u for u in train},
          "smap": {i: i for i in meta}}
    return ds, {"test_users": list(range(1, num_users + 1)), "test_candidates": cands}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=4000)
    ap.add_argument("--gpu", action="store_true")
    a = ap.parse_args()
    from llamarec_amd.rerank import LazyEvalItems, LLMEvaluator, build_test_items
    from tests.local_tokenizer import build_llama_like_tokenizer

    tok = build_llama_like_tokenizer()
    ds, retrieved = beauty_like(a.users)
    args = SimpleNamespace(llm_max_history=20, llm_max_title_len=32, llm_max_text_len=1536, llm_system_template=None,
                           llm_input_template=None, rerank_metric_ks=[1, 5, 10], test_batch_size=16,
                           test_batch_size_explicit=False, eval_token_budget=None)
    n_eager = min(a.users, 1000)
    sub = {"test_users": retrieved["test_users"][:n_eager], "test_candidates": retrieved["test_candidates"][:n_eager]}
    t0 = time.perf_counter()
    eager = build_test_items(ds, sub, tok, args)
    t_eager = time.perf_counter() - t0
    lazy = LazyEvalItems(ds, retrieved, tok, args, split="test")
    t0 = time.perf_counter()
    est = lazy.estimate_lengths()
    t_est = time.perf_counter() - t0
    t0 = time.perf_counter()
    got = []
    for c0 in range(0, a.users, 512):
        got += lazy.build(c0, min(c0 + 512, a.users))
    t_lazy = time.perf_counter() - t0
    assert [g["input_ids"] for g in got[:n_eager]] == [e["input_ids"] for e in eager]
    mean_tok = float(np.mean([len(g["input_ids"]) for g in got]))
    print(f"users {a.users}, mean prompt {mean_tok:.0f} tokens (real fast tokenizer, {os.cpu_count()} host cpus)")
    print(f"reference path (42 tokenizer calls per user, serial): {n_eager / t_eager:8.0f} users/s")
    print(f"lazy path: shard estimate {a.users / t_est:8.0f} users/s (incl. the catalog's title cache); "
          f"build {a.users / t_lazy:8.0f} users/s; identical ids")
    if not a.gpu:
        return
    import torch

    from llamarec_amd.llm import LLAMA2_7B, LlamaRanker
    from llamarec_amd.verb import ManualVerbalizer

    model = LlamaRanker.random_init(dict(LLAMA2_7B, vocab_size=max(32000, tok.vocab_size)), seed=1)
    verb = ManualVerbalizer(tokenizer=tok, prefix="", post_log_softmax=False, classes=list(range(20)),
                            label_words={i: chr(65 + i) for i in range(20)})
    LLMEvaluator(args, model, got[:64], verb).predict()                      # warm-up (workspace, kernels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    items = build_test_items(ds, retrieved, tok, args)
    m1 = LLMEvaluator(args, model, items, verb).predict()
    torch.cuda.synchronize()
    t_a = time.perf_counter() - t0
    t0 = time.perf_counter()
    m2 = LLMEvaluator(args, model, LazyEvalItems(ds, retrieved, tok, args, split="test"), verb).predict()
    torch.cuda.synchronize()
    t_b = time.perf_counter() - t0
    same = all(abs(m1[k] - m2[k]) < 1e-12 for k in m1 if k.startswith("test_") and "runtime" not in k and "second" not in k)
    print(f"GPU, {a.users} users end to end (tokenise + prefill + metrics): eager list {a.users / t_a:.1f} users/s "
          f"(GPU loop alone {m1['test_samples_per_second']:.1f}); lazy streamed {a.users / t_b:.1f} users/s; same metrics: {same}")


if __name__ == "__main__":
    main()
