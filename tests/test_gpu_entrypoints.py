"""GPU: the eval mirrors reproduce the reference's retrieved.pkl content and the entry points run
end to end on synthetic assets with the reference's output layout."""
import json
import os
import pickle
from types import SimpleNamespace

import numpy as np
import torch
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
    out = train_retriever.main(["--dataset_code", "synthetic", "--synthetic", "--export_root", lru_root,
                                "--max_train_iterations", "30", "--val_iterations", "10"])
    assert os.path.exists(os.path.join(lru_root, "models", "best_acc_model.pth"))   # trainer/loggers.py:121
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
                                          "--export_root", llm_root, "--lora_max_steps", "6", "--lora_val_iterations", "3",
                                          "--warmup_steps", "2", "--lora_micro_batch_size", "4", "--train_batch_size", "8",
                                          "--lora_max_val_samples", "16", "--llm_max_history", "5"])
    # trainer.train() ran (train_ranker.py:110): the tuned adapter is on disk in PEFT's format, B is no longer zero
    from safetensors import safe_open

    for sub_dir in ("adapter", "best_adapter"):
        assert json.load(open(os.path.join(llm_root, sub_dir, "adapter_config.json")))["r"] == 8
        with safe_open(os.path.join(llm_root, sub_dir, "adapter_model.safetensors"), framework="pt") as f:
            keys = list(f.keys())
            assert "base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight" in keys and len(keys) == 8
            assert float(f.get_tensor(keys[0]).abs().max()) > 0
    hist = json.load(open(os.path.join(llm_root, "lora_eval_history.json")))
    assert [h["step"] for h in hist] == [3, 6] and "eval_NDCG@10" in hist[0]
    # scoring only, with the saved adapter merged at load: the same test metrics as the run that trained it
    m2, _ = train_ranker.main(["--dataset_code", "synthetic", "--synthetic", "--llm_retrieved_path", lru_root,
                               "--export_root", str(tmp_path / "again"), "--eval_only", "--llm_adapter_path",
                               os.path.join(llm_root, "best_adapter"), "--llm_max_history", "5"])
    for k in ("test_Recall@10", "test_NDCG@10", "test_MRR@5"):
        assert abs(m2[k] - metrics[k]) < 0.05
    sub = json.load(open(os.path.join(llm_root, "subset_metrics.json")))
    ov = json.load(open(os.path.join(llm_root, "overall_metrics.json")))
    assert sub["test_loss"] == -1.0 and set(ov) == {f"test_{m}@{k}" for m in ("Recall", "MRR", "NDCG") for k in (1, 5, 10)}
    n_ret, n_all = r["test_retrieval"]["retrieval_size"], r["test_retrieval"]["original_size"]
    assert abs(ov["test_Recall@10"] - sub["test_Recall@10"] * n_ret / n_all) < 1e-12


def test_retriever_training_learns_and_checkpoints(tmp_path):
    """train_retriever's training half end to end on a learnable synthetic dataset (each user walks the item ids
    upwards): Recall@10 on the held-out next item goes from chance to high, the best checkpoint is written in the
    reference's format and the scoring path reloads it."""
    from types import SimpleNamespace

    from llamarec_amd import data as D
    from llamarec_amd.lru import LRURec
    from llamarec_amd.retrieve import LRUEvaluator
    from llamarec_amd.train import LRUTrainer

    V, U, L = 150, 256, 12
    rng = np.random.default_rng(0)
    train, val, test = {}, {}, {}
    for u in range(1, U + 1):
        start, n = int(rng.integers(0, V)), int(rng.integers(6, 15))
        seq = [(start + i) % V + 1 for i in range(n)]
        train[u], val[u], test[u] = seq[:-2], seq[-2:-1], seq[-1:]
    ds = {"train": train, "val": val, "test": test, "umap": {u: u for u in train}, "smap": {i: i for i in range(1, V + 1)}}
    args = SimpleNamespace(num_items=V, bert_max_len=L, bert_num_blocks=2, train_batch_size=64, val_batch_size=64,
                           lr=3e-3, weight_decay=1e-2, adam_epsilon=1e-9, max_grad_norm=5.0, bert_dropout=0.1,
                           bert_attn_dropout=0.1, seed=1, num_epochs=60, val_strategy="iteration", val_iterations=40,
                           early_stopping_patience=50, best_metric="Recall@10", metric_ks=[1, 5, 10],
                           sliding_window_size=1.0, max_train_iterations=None)
    root = str(tmp_path / "lru")
    _, v_ids, v_lab = D.lru_eval_arrays(ds, "val", L)
    val_loader = list(D.batches(v_ids, v_lab, 64))
    tr = LRUTrainer(args, export_root=root)
    losses = tr.train(D.lru_train_sequences(ds, L), val_loader)
    assert tr.history[0]["Recall@10"] < 0.3 and tr.best_metric > 0.9, (tr.history[0], tr.best_metric)
    assert losses[-1] < losses[0] - 1.0
    ckpt = torch.load(os.path.join(root, "models", "best_acc_model.pth"), map_location="cpu", weights_only=False)
    sd = ckpt["model_state_dict"]
    assert sd["model.lru_blocks.0.lru_layer.in_proj.weight"].dtype == torch.complex64      # reference dtypes
    assert tuple(sd["embedding.token.weight"].shape) == (V + 1, 64)
    again = LRUEvaluator(args, LRURec.from_checkpoint(os.path.join(root, "models", "best_acc_model.pth")), val_loader, []).validate()
    assert abs(again["Recall@10"] - tr.best_metric) < 1e-9


def test_online_single_user_path(golden_dir):
    """demo/inference.py flow: retrieve (no mask, no padding) -> prompt -> rank, vs the oracles."""
    from llamarec_amd import data as D
    from llamarec_amd import inference as I
    from llamarec_amd.llm import LlamaRanker
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from llamarec_amd.synth import synth_llama_state
    from llamarec_amd.verb import ManualVerbalizer
    from oracle import llama_oracle as LO
    from oracle import lru_oracle as O
    from tests.fake_tokenizer import FakeTokenizer

    ds = D.synthetic_dataset(num_users=5, num_items=400, seed=3)
    sd = init_lru_state_dict(400, seed=9)
    retr = LRURec.from_state_dict(sd)
    query = ds["train"][1][-7:]
    cands = I.retrieve_candidates(retr, query, top_k=20)
    oi, _ = O.LruOracle(sd).retrieve_topk(np.asarray(query)[None, :], 20, False)
    assert cands == oi[0].tolist()
    prompt = I.generate_prompt(query, cands, ds["meta"])
    assert prompt.startswith("### Instruction:\n") and prompt.endswith("### Response:\n") and "(T) " in prompt
    tok = FakeTokenizer()
    cfg = dict(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               num_key_value_heads=2, max_position_embeddings=2048, rms_norm_eps=1e-5, rope_theta=10000.0)
    lsd = synth_llama_state(cfg, 11)
    ranker = LlamaRanker.from_state_dict(lsd, cfg)
    verb = ManualVerbalizer(tokenizer=tok, classes=list(range(20)), label_words={i: chr(65 + i) for i in range(20)})
    top = I.rank_candidates(ranker, tok, prompt, cands, verb, top_k=10)
    assert len(top) == 10 and set(top) <= set(cands)
    ids = tok(prompt)["input_ids"]
    ref = LO.prefill_verbalize(lsd, cfg, [np.asarray(ids)], verb.label_token_ids, "bf16")[0]
    got = ranker.prefill_verbalize([np.asarray(ids)], verb.label_token_ids)[0].cpu().numpy()
    assert np.abs(got - ref).max() < 3e-2
    # same ranking wherever the oracle's score gaps are clear of the tolerance
    order = np.argsort(-ref, kind="stable")
    gaps = -np.diff(ref[order])
    for j in range(9):
        if gaps[j] > 6e-2 and (j == 0 or gaps[j - 1] > 6e-2):
            assert top[j] == cands[order[j]]


def _write_hf_dir(path, sd, cfg, shards=2):
    """A local HF model directory as the reference's from_pretrained reads it (train_ranker.py:57-62): config.json +
    `shards` safetensors files + the index json, bf16 tensors under HF's parameter names."""
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    json.dump(dict(cfg, architectures=["LlamaForCausalLM"], model_type="llama", torch_dtype="bfloat16"),
              open(os.path.join(path, "config.json"), "w"))
    names = list(sd)
    weight_map = {}
    for s in range(shards):
        fn = f"model-{s + 1:05d}-of-{shards:05d}.safetensors"
        part = {n: torch.from_numpy(sd[n]).to(torch.bfloat16) for n in names[s::shards]}
        save_file(part, os.path.join(path, fn), metadata={"format": "pt"})
        weight_map.update({n: fn for n in part})
    json.dump({"metadata": {}, "weight_map": weight_map}, open(os.path.join(path, "model.safetensors.index.json"), "w"))


def _write_peft_adapter(path, cfg, r=8, alpha=32, seed=0):
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    rng = np.random.default_rng(seed)
    d = cfg["hidden_size"]
    hd = d // cfg["num_attention_heads"]
    w, lora = {}, {}
    for i in range(cfg["num_hidden_layers"]):
        for proj, out in (("q_proj", cfg["num_attention_heads"] * hd), ("v_proj", cfg["num_key_value_heads"] * hd)):
            base = f"model.layers.{i}.self_attn.{proj}"
            A = (rng.standard_normal((r, d)) * 0.05).astype(np.float32)
            B = (rng.standard_normal((out, r)) * 0.05).astype(np.float32)
            w[f"base_model.model.{base}.lora_A.weight"] = torch.from_numpy(A)      # peft 0.11 key layout on disk
            w[f"base_model.model.{base}.lora_B.weight"] = torch.from_numpy(B)
            lora[f"{base}.lora_A.weight"], lora[f"{base}.lora_B.weight"] = A, B
    save_file(w, os.path.join(path, "adapter_model.safetensors"))
    json.dump({"peft_type": "LORA", "r": r, "lora_alpha": alpha, "target_modules": ["q_proj", "v_proj"],
               "lora_dropout": 0.05, "bias": "none", "task_type": "CAUSAL_LM"},
              open(os.path.join(path, "adapter_config.json"), "w"))
    return dict(r=r, alpha=alpha, weights=lora)


@pytest.mark.parametrize("load_in_4bit", [True, False])
def test_from_pretrained_local_hf_dir_and_peft_adapter(tmp_path, load_in_4bit):
    """a local HF directory (config.json + sharded safetensors) and a PEFT adapter directory, loaded
    the way the reference loads them (train_ranker.py:57-79: NF4 base, adapter on top) -> the same scores, bit for bit,
    as from_state_dict on the same tensors; and the adapter / the quantisation really change the model."""
    from llamarec_amd.llm import LlamaRanker
    from llamarec_amd.synth import synth_llama_state

    cfg = dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    sd = synth_llama_state(cfg, 21)
    hf, ad = str(tmp_path / "hf"), str(tmp_path / "adapter")
    _write_hf_dir(hf, sd, cfg, shards=2)
    lora = _write_peft_adapter(ad, cfg)
    rng = np.random.default_rng(0)
    seqs = [np.concatenate([[1], rng.integers(3, 320, size=n)]) for n in (5, 70, 200, 33)]
    label_ids = list(range(40, 60))
    loaded = LlamaRanker.from_pretrained(hf, adapter_path=ad, load_in_4bit=load_in_4bit)
    direct = LlamaRanker.from_state_dict(sd, cfg, lora=lora, nf4=load_in_4bit)
    a, b = loaded.prefill_verbalize(seqs, label_ids), direct.prefill_verbalize(seqs, label_ids)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert loaded.config["hidden_size"] == 256 and loaded.config["num_hidden_layers"] == 2   # taken from config.json
    plain = LlamaRanker.from_pretrained(hf, load_in_4bit=load_in_4bit).prefill_verbalize(seqs, label_ids)
    assert (plain - a).abs().max() > 1e-2                                      # the adapter is really merged
    other = LlamaRanker.from_pretrained(hf, adapter_path=ad, load_in_4bit=not load_in_4bit).prefill_verbalize(seqs, label_ids)
    assert not torch.equal(other, a)                                           # and so is the NF4 round trip


def test_train_retriever_reads_dataset_pkl_from_the_preprocessed_folder(tmp_path):
    """`python train_retriever.py --dataset_code beauty --eval_only` WITHOUT --synthetic reads
    data/preprocessed/beauty_min_rating0-min_uc5-min_sc5/dataset.pkl (llamarec_datasets/base.py:125-140) and a
    best_acc_model.pth in the reference's checkpoint format (trainer/loggers.py:7-8), and writes the reference's
    output layout."""
    import train_retriever
    from llamarec_amd import data as D
    from llamarec_amd.lru import init_lru_state_dict

    ds = D.synthetic_dataset(num_users=120, num_items=400, seed=3)
    path = D.preprocessed_path(str(tmp_path / "data"), "beauty", 0, 5, 5)
    assert path.endswith(os.path.join("preprocessed", "beauty_min_rating0-min_uc5-min_sc5", "dataset.pkl"))
    os.makedirs(os.path.dirname(path))
    pickle.dump(ds, open(path, "wb"))
    assert set(D.load_dataset_pkl(path)) >= {"train", "val", "test", "meta", "umap", "smap"}
    root = str(tmp_path / "experiments" / "lru" / "beauty")
    os.makedirs(os.path.join(root, "models"))
    sd = init_lru_state_dict(400, seed=9)
    torch.save({"model_state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "epoch": 0},
               os.path.join(root, "models", "best_acc_model.pth"))
    out = train_retriever.main(["--dataset_code", "beauty", "--data_root", str(tmp_path / "data"), "--eval_only",
                                "--export_root", root])
    assert len(out["test_probs"]) == 120 and all(len(p) == 50 for p in out["test_probs"])
    r = pickle.load(open(os.path.join(root, "retrieved.pkl"), "rb"))
    assert r["test_probs"] == out["test_probs"] and os.path.exists(os.path.join(root, "test_metrics.json"))
    # Beauty template defaults apply without --synthetic (config.py:57-61,103-111): L = 50, eval batch 64
    from llamarec_amd import config as cfg

    a = cfg.parse(["--dataset_code", "beauty"], model_code="lru")
    assert a.bert_max_len == 50 and a.test_batch_size == 64
    with pytest.raises(SystemExit):       # no checkpoint and not synthetic: refuse, do not invent weights
        train_retriever.main(["--dataset_code", "beauty", "--data_root", str(tmp_path / "data"), "--eval_only",
                              "--export_root", str(tmp_path / "nowhere")])
    with pytest.raises(ValueError, match="missing key"):
        pickle.dump({"train": {}}, open(path, "wb"))
        D.load_dataset_pkl(path)
