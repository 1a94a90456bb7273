"""CPU: token-budget batching, token-balanced shards, and the shared-prefix helpers (host logic around the ranker's
packed prefill; eval order is free in the reference: dataloader/llm.py:196-202, trainer/llm.py:63-72)."""
import numpy as np
import pytest

from llamarec_amd.packing import TOKEN_BUDGET, shard_by_tokens, token_budget_steps
from llamarec_amd.synth import TEMPLATE_PREFIX_TOKENS, synth_prompt_tokens, synth_users


@pytest.mark.parametrize("workload,n", [("beauty", 500), ("ml-100k", 610), ("games", 300)])
def test_token_budget_steps_fill_the_budget(workload, n):
    _, _, _, T = synth_users(workload, n)
    steps = token_budget_steps(T)
    flat = np.concatenate(steps)
    assert sorted(flat.tolist()) == list(range(n))                 # every prompt exactly once
    sums = np.array([int(T[s].sum()) for s in steps])
    assert (sums <= TOKEN_BUDGET).all()
    assert (sums[:-2] == TOKEN_BUDGET).all()                        # exact subset sums while the window is full
    # no starvation: every step takes the oldest prompt still pending
    pending = set(range(n))
    for s in steps:
        assert int(s[0]) == min(pending)
        pending -= set(s.tolist())


def test_token_budget_steps_edge_cases():
    assert token_budget_steps([]) == []
    assert [s.tolist() for s in token_budget_steps([5], budget=8)] == [[0]]
    assert [s.tolist() for s in token_budget_steps([8, 8, 8], budget=8)] == [[0], [1], [2]]
    # best subset of the window, oldest first: 7 + 1 fills 8 exactly, then 6 + 2, then 5 alone
    assert [s.tolist() for s in token_budget_steps([7, 6, 5, 1, 2], budget=8)] == [[0, 3], [1, 4], [2]]
    assert [s.tolist() for s in token_budget_steps([3, 3, 3, 3], budget=100, max_prompts=3)] == [[0, 1, 2], [3]]
    with pytest.raises(ValueError):
        token_budget_steps([9], budget=8)
    with pytest.raises(ValueError):
        token_budget_steps([0, 3], budget=8)


def test_shard_by_tokens_is_contiguous_and_balanced():
    rng = np.random.default_rng(0)
    T = rng.integers(50, 1500, size=1000)
    for W in (1, 2, 3, 8):
        edges = shard_by_tokens(T, W)
        assert edges[0][0] == 0 and edges[-1][1] == len(T)
        assert all(edges[r][1] == edges[r + 1][0] for r in range(W - 1))      # positional user ids survive
        sums = np.array([T[a:b].sum() for a, b in edges])
        assert sums.max() - sums.min() <= 2 * T.max()
    # by user count the same data would be off by far more than one prompt
    skew = np.concatenate([np.full(500, 1500), np.full(500, 50)])
    (a0, b0), (a1, b1) = shard_by_tokens(skew, 2)
    assert abs(int(skew[a0:b0].sum()) - int(skew[a1:b1].sum())) <= 1500 and b0 < 500
    assert shard_by_tokens([5, 5], 4)[-1][1] == 2 and sum(b - a for a, b in shard_by_tokens([5, 5], 4)) == 2


def test_common_prefix_len_host_and_c_helper_agree():
    from llamarec_amd._lib import lib
    from llamarec_amd.llm import common_prefix_len, pack_prompts

    rng = np.random.default_rng(1)
    cases = []
    for P in (0, 1, 5, 36):
        pre = rng.integers(3, 1000, size=P)
        seqs = [np.concatenate([pre, [2000 + b], rng.integers(3, 1000, size=int(n))]) for b, n in enumerate([0, 4, 40])]
        cases.append((P, seqs))
    cases.append((3, [np.arange(10), np.arange(4)]))          # capped at (shortest prompt - 1)
    cases.append((0, [np.arange(10)]))                        # a single prompt shares nothing
    cases.append((0, [np.array([1]), np.array([1, 2, 3])]))   # a 1-token prompt keeps its token
    for want, seqs in cases:
        ids, cu = pack_prompts(seqs)
        assert common_prefix_len(ids, cu) == want
        assert lib().lr_common_prefix_len(ids.ctypes.data, cu.ctypes.data, len(seqs)) == want   # pure host code


def test_synthetic_prompts_share_the_template_prefix():
    from llamarec_amd.llm import common_prefix_len

    T = np.array([600, 1536, 700, 48])
    ids, cu = synth_prompt_tokens(T, seed=3)
    assert (ids[cu[:-1]] == 1).all()                                   # BOS everywhere
    assert common_prefix_len(ids, cu) == 1                             # the left-truncated 1536-token prompt lost its head
    ids2, cu2 = synth_prompt_tokens(T[[0, 2, 3]], seed=3)
    assert common_prefix_len(ids2, cu2) == TEMPLATE_PREFIX_TOKENS
    ids3, cu3 = synth_prompt_tokens(T[[0, 2, 3]], seed=3, shared_prefix=False)
    assert common_prefix_len(ids3, cu3) == 1
