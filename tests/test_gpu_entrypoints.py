"""GPU: the eval mirrors reproduce the reference's retrieved.pkl content and the entry points run
end to end on synthetic assets with the reference's output layout."""
import json
import os
import pickle
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_generate_candidates_equals_reference_dict(golden_dir):
    from llamarec_amd import data as D
    from llamarec_amd.lru import LRURec
    from llamarec_amd.retrieve import LRUEvaluator

    z = np.load(os.path.join(golden_dir, "lru_v300.npz"))
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    g = json.load(open(os.path.join(golden_dir, "candidates.json")))
    zin = np.load(os.path.join(golden_dir, "candidates_inputs.npz"))
    args = SimpleNamespace(metric_ks=g["ks"], num_users=g["num_users"], num_items=300, llm_negative_sample_size=19)
    B = g["batch_size"]
    ev = LRUEvaluator(args, LRURec.from_state_dict(sd), list(D.batches(zin["val_ids"], zin["val_labels"][:, None], B)),
                      list(D.batches(zin["test_ids"], zin["test_labels"][:, None], B)))
    out = ev.generate_candidates(None)
    ref = g["retrieved"]
    assert list(out.keys()) == list(ref.keys())
    for k in ("val_users", "val_candidates", "test_probs", "test_labels", "test_users", "test_candidates", "non_test_users"):
        assert out[k] == ref[k], k
    for k in ("val_metrics", "test_metrics"):
        assert list(out[k].keys()) == list(ref[k].keys())
        for m, v in ref[k].items():
            assert abs(out[k][m] - v) < 1e-6, (k, m)
    tr, rr = out["test_retrieval"], ref["test_retrieval"]
    assert tr["original_size"] == rr["original_size"] and tr["retrieval_size"] == rr["retrieval_size"]
    for k in ("original_metrics", "retrieval_metrics", "non_retrieval_metrics"):
        for m, v in rr[k].items():
            assert abs(tr[k][m] - v) < 1e-6, (k, m)
    # BaseTrainer.test: unweighted mean of per-batch means (trainer/base.py:170)
    avg = ev.test()
    nb = len(g["per_batch_test_metrics"])
    for m in avg:
        assert abs(avg[m] - sum(b[m] for b in g["per_batch_test_metrics"]) / nb) < 1e-6
    val = ev.validate()  # history not excluded
    for m in ("NDCG@10", "Recall@50"):
        assert abs(val[m] - sum(b[m] for b in g["per_batch_val_metrics_no_exclude"]) / nb) < 1e-6


def test_entry_points_synthetic(tmp_path):
    import train_ranker
    import train_retriever

    lru_root = str(tmp_path / "experiments" / "lru" / "synthetic")
    out = train_retriever.main(["--dataset_code", "synthetic", "--synthetic", "--export_root", lru_root])
    assert os.path.exists(os.path.join(lru_root, "test_metrics.json"))
    r = pickle.load(open(os.path.join(lru_root, "retrieved.pkl"), "rb"))
    assert set(r) == {"val_metrics", "val_users", "val_candidates", "test_probs", "test_labels", "test_metrics",
                      "test_users", "test_candidates", "non_test_users", "test_retrieval"}
    assert len(r["test_probs"]) == 300 and len(r["test_probs"][0]) == 50
    assert all(len(c) == 20 for c in r["test_candidates"])
    if not r["test_users"]:
        pytest.skip("random retriever retrieved nobody")
    llm_root = str(tmp_path / "experiments" / "tiny" / "synthetic")
    metrics, overall = train_ranker.main(["--dataset_code", "synthetic", "--synthetic", "--llm_retrieved_path", lru_root,
                                          "--export_root", llm_root])
    sub = json.load(open(os.path.join(llm_root, "subset_metrics.json")))
    ov = json.load(open(os.path.join(llm_root, "overall_metrics.json")))
    assert sub["test_loss"] == -1.0 and set(ov) == {f"test_{m}@{k}" for m in ("Recall", "MRR", "NDCG") for k in (1, 5, 10)}
    n_ret, n_all = r["test_retrieval"]["retrieval_size"], r["test_retrieval"]["original_size"]
    assert abs(ov["test_Recall@10"] - sub["test_Recall@10"] * n_ret / n_all) < 1e-12
