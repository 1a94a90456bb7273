"""Token-budget batching of ranker prompts and token-balanced data-parallel shards.

The reference evaluates the ranker in fixed-size batches (ranker eval batch 16 / 32, config.py:98) of
left-padded prompts; the order of evaluation samples carries no meaning (its LLM loaders even shuffle,
dataloader/llm.py:196-202) and a prompt's scores do not depend on what else is in the batch (unpadded
execution; tests/test_gpu_llama.py::test_full_width_batch_invariance). On MI355X the prefill is a chain
of [sum(T), K] x [K, N] GEMMs on 256 x 256 output tiles over 256 CUs, so the batch that fills the chip is
the one whose TOKEN count -- not prompt count -- is just under a multiple of 256 rows chosen so that
ceil(sum(T) / 256) * (N / 256) is a whole number of 256-CU rounds for every Llama-2-7b projection:
TOKEN_BUDGET = 32768 rows = 128 row tiles -> N = 4096: 2048 tiles = 8.0 rounds, N = 12288: 24.0, N = 22016: 43.0
(16384 rows leave gate-up at 21.5 rounds; a 16-prompt Beauty batch is 58 row tiles = 3.6 rounds with a 74 % full last
row tile). Same-box A/B 16384 vs 32768: +0.6 % users/s; pass budget=16384 where the step latency
or the activation workspace (about 80 KB per row) matters more.

`token_budget_steps` walks the prompts in dataset order; each step takes the oldest pending prompt plus the
subset of the next `window - 1` whose lengths add up closest to (never above) the budget (an exact
subset-sum over Python big-int bit sets: 127 shifts of a 32 k-bit integer per step), so no prompt is
starved and a step's token count is normally the budget itself.

`shard_by_tokens` is SURVEY.md section 8(e)'s partitioning: contiguous user blocks (user ids stay positional,
trainer/lru.py:85,127) whose boundaries balance sum(T) instead of the user count.
"""
from __future__ import annotations

import numpy as np

TOKEN_BUDGET = 32768
WINDOW = 128


def token_budget_steps(lengths, budget: int = TOKEN_BUDGET, window: int = WINDOW, max_prompts: int | None = None,
                       shared_prefix: int = 0):
    """lengths: prompt lengths in tokens (each 1..budget). Returns a list of int64 index arrays (ascending inside
    a step), every index exactly once, each step's lengths summing to <= budget (and holding <= max_prompts
    prompts if given). shared_prefix = P > 0: every prompt starts with the same P tokens, which the prefill runs once
    per step (lr_llama_prefill_verbalize_prefix), so a step of prompts T_1..T_n occupies P + sum(T_i - P) rows: the
    budget is applied to that."""
    T = np.asarray(lengths, dtype=np.int64).reshape(-1)
    if shared_prefix:
        if shared_prefix < 0 or (T.size and shared_prefix >= T.min()):
            raise ValueError("shared_prefix must be shorter than every prompt")
        return token_budget_steps(T - shared_prefix, budget - shared_prefix, window, max_prompts)
    if T.size and (T.min() < 1 or T.max() > budget):
        raise ValueError(f"prompt lengths must be in [1, {budget}] (got {int(T.min())}..{int(T.max())})")
    if window < 1:
        raise ValueError("window must be >= 1")
    pending = list(range(T.size))
    steps = []
    while pending:
        win = pending[:window]
        first, rest = win[0], win[1:]
        cap = budget - int(T[first])
        full = (1 << (cap + 1)) - 1
        reach, before = 1, []                     # bit s of `reach`: some subset of the items seen so far sums to s
        for i in rest:
            before.append(reach)
            reach |= (reach << int(T[i])) & full
        s = reach.bit_length() - 1                # the largest reachable sum <= cap
        chosen = [first]
        for i, prev in zip(reversed(rest), reversed(before)):
            if not (prev >> s) & 1:               # s is not reachable without item i
                chosen.append(i)
                s -= int(T[i])
        chosen.sort()
        if max_prompts is not None and len(chosen) > max_prompts:
            chosen = chosen[:max_prompts]         # keeps `first` (the smallest index of the window)
        taken = set(chosen)
        pending = [i for i in pending if i not in taken]
        steps.append(np.asarray(chosen, dtype=np.int64))
    return steps


def shard_by_tokens(lengths, world: int):
    """Contiguous blocks [lo, hi) per rank with balanced token sums: rank r ends at the first index where the
    running sum reaches (r + 1) / world of the total. Every index belongs to exactly one block; blocks may be empty
    only when there are fewer items than ranks."""
    T = np.asarray(lengths, dtype=np.int64).reshape(-1)
    if world < 1:
        raise ValueError("world must be >= 1")
    csum = np.concatenate([[0], np.cumsum(T)])
    total = int(csum[-1])
    edges = [0]
    for r in range(1, world):
        target = total * r / world
        e = int(np.searchsorted(csum, target, side="left"))
        # csum[e] >= target > csum[e - 1]: cut at whichever side is closer to the target
        if e > 0 and e <= T.size and (target - csum[e - 1]) < (csum[min(e, T.size)] - target):
            e -= 1
        edges.append(min(max(e, edges[-1]), T.size))
    edges.append(T.size)
    return [(edges[r], edges[r + 1]) for r in range(world)]
