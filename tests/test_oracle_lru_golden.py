"""Pin the stage-1 C oracle against the reference's own outputs (tests/golden, made by
tests/gen_goldens.py running /root/reference/model/lru.py and trainer/lru.py)."""
import json
import os

import numpy as np
import pytest

from oracle import lru_oracle as O

TOL = 2e-5  # fp32: sequential scan vs the reference's recursive-doubling scan


@pytest.fixture(scope="module")
def lru(golden_dir):
    z = np.load(os.path.join(golden_dir, "lru_v300.npz"))
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    return z, sd, O.LruOracle(sd)


@pytest.mark.parametrize("case", ["L50", "L7", "L64", "L200"])
def test_scores_last_match_reference(lru, case):
    z, _, orc = lru
    ids = z[f"ids/{case}"]
    got = orc.scores_last(ids, exclude_history=False)
    ref = z[f"scores_last/{case}"]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < TOL


def test_topk_matches_reference_where_gap_is_clear(lru):
    z, _, orc = lru
    for case in ("L50", "L200"):
        ids = z[f"ids/{case}"]
        ref = z[f"scores_last/{case}"].copy()
        for b in range(ids.shape[0]):  # trainer/lru.py:35-38
            ref[b, ids[b]] = -1e9
            ref[b, 0] = -1e9
        idx, sc = orc.retrieve_topk(ids, 50, exclude_history=True)
        order = np.argsort(-ref, axis=1, kind="stable")[:, :51]
        for b in range(ids.shape[0]):
            srt = ref[b, order[b]]
            for j in range(50):
                if idx[b, j] != order[b, j]:  # only allowed inside a near-tie group
                    assert abs(ref[b, idx[b, j]] - srt[j]) < 4 * TOL
            assert np.abs(sc[b] - srt[:50]).max() < TOL


def test_retrieve_equals_scores_then_topk(lru):
    z, _, orc = lru
    ids = z["ids/L50"]
    for excl in (False, True):
        s = orc.scores_last(ids, excl)
        i1, s1 = O.topk(s, 20)
        i2, s2 = orc.retrieve_topk(ids, 20, excl)
        assert np.array_equal(i1, i2) and np.array_equal(s1, s2)
        if excl:
            assert (s[:, 0] == np.float32(-1e9)).all()


def test_tie_rule_lower_id_first():
    s = np.zeros((1, 10), np.float32)
    s[0, 7] = 1.0
    idx, sc = O.topk(s, 4)
    assert idx.tolist() == [[7, 0, 1, 2]]


def test_calculate_metrics_and_generate_candidates(golden_dir, lru):
    _, _, orc = lru
    g = json.load(open(os.path.join(golden_dir, "candidates.json")))
    zin = np.load(os.path.join(golden_dir, "candidates_inputs.npz"))
    ks, B, U = g["ks"], g["batch_size"], g["num_users"]
    # calculate_metrics (test split, history excluded): batch means
    idx, _ = orc.retrieve_topk(zin["test_ids"], 50, True)
    for bi, ref in enumerate(g["per_batch_test_metrics"]):
        sl = slice(bi * B, min((bi + 1) * B, U))
        sums = O.rank_metric_sums(idx[sl], zin["test_labels"][sl], ks)
        n = sl.stop - sl.start
        for j, k in enumerate(ks):
            for m, name in enumerate(("Recall", "MRR", "NDCG")):
                assert abs(sums[j, m] / n - ref[f"{name}@{k}"]) < 1e-6
    # validation: no history exclusion (trainer/base.py:141-143)
    idxv, _ = orc.retrieve_topk(zin["val_ids"], 50, False)
    for bi, ref in enumerate(g["per_batch_val_metrics_no_exclude"]):
        sl = slice(bi * B, min((bi + 1) * B, U))
        sums = O.rank_metric_sums(idxv[sl], zin["val_labels"][sl], ks)
        for j, k in enumerate(ks):
            assert abs(sums[j, 2] / (sl.stop - sl.start) - ref[f"NDCG@{k}"]) < 1e-6
    # generate_candidates: ordered top-20 lists, user ids, summed metrics / num_users
    r = g["retrieved"]
    assert r["test_probs"] == idx.tolist()
    top20 = idx[:, :20]
    users = [u + 1 for u in range(U) if zin["test_labels"][u] in top20[u]]
    assert users == r["test_users"]
    assert [top20[u - 1].tolist() for u in users] == r["test_candidates"]
    assert [u + 1 for u in range(U) if zin["test_labels"][u] not in top20[u]] == r["non_test_users"]
    sums = O.rank_metric_sums(idx, zin["test_labels"], ks)
    for j, k in enumerate(ks):
        for m, name in enumerate(("Recall", "MRR", "NDCG")):
            assert abs(sums[j, m] / U - r["test_metrics"][f"{name}@{k}"]) < 1e-6
    idxv20, _ = orc.retrieve_topk(zin["val_ids"], 20, True)
    vusers = [u + 1 for u in range(U) if zin["val_labels"][u] in idxv20[u]]
    assert vusers == r["val_users"]
    assert [idxv20[u - 1].tolist() for u in vusers] == r["val_candidates"]


def test_metrics_oracle_vs_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    g = json.load(open(os.path.join(golden_dir, "metrics.json")))
    sums = O.rank_metric_sums(z["ranked"], z["labels"], g["ks"])
    for j, k in enumerate(g["ks"]):
        for m, name in enumerate(("Recall", "MRR", "NDCG")):
            assert abs(sums[j, m] / len(z["labels"]) - g["full"][f"{name}@{k}"]) < 1e-6
            assert abs(sums[j, m] / len(z["labels"]) - g["wrapper_preprocessed_bs10"][f"{name}@{k}"]) < 1e-6
    # the reranker's 20-class case (trainer/llm.py:63-72)
    idx, _ = O.topk(z["s20"], 20)
    sums = O.rank_metric_sums(idx, z["l20"], g["rerank_ks"])
    for j, k in enumerate(g["rerank_ks"]):
        for m, name in enumerate(("Recall", "MRR", "NDCG")):
            assert abs(sums[j, m] / len(z["l20"]) - g["rerank"][f"{name}@{k}"]) < 1e-6
