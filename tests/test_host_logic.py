"""CPU tests of the host-side mirrors (prompt building, eval collation, verbalizer, metric merge)
against data captured from the reference (tests/golden/*.json|npz)."""
import json
import os

import numpy as np
import pytest

from llamarec_amd import prompt as P
from llamarec_amd.verb import ManualVerbalizer
from tests.fake_tokenizer import FakeTokenizer


@pytest.fixture(scope="module")
def prompts(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "prompts.json")))
    g["titles"] = {int(k): v for k, v in g["titles"].items()}
    return g


def test_templates_match_reference_defaults(prompts):
    assert P.DEFAULT_SYSTEM_TEMPLATE == prompts["system_template"]
    assert P.DEFAULT_INPUT_TEMPLATE == prompts["input_template"]


def test_prompt_text_and_token_ids(prompts):
    for c in prompts["cases"]:
        tok = FakeTokenizer()
        out = P.seq_to_token_ids(c["seq"], c["candidates"], c["label"], prompts["titles"], tok,
                                 max_title_len=c["llm_max_title_len"], max_text_len=c["llm_max_text_len"])
        assert tok.seen_texts[-1] == c["prompt_eval"]
        assert out["input_ids"] == c["eval"]["input_ids"]
        assert out["attention_mask"] == c["eval"]["attention_mask"]
        assert out["labels"] == c["eval"]["labels"]


def test_answer_must_be_a_candidate(prompts):
    with pytest.raises(ValueError):
        P.seq_to_token_ids([1], [2, 3], 9, prompts["titles"], FakeTokenizer())


def test_eval_collate_and_pack(prompts):
    cases = prompts["cases"]
    for col in prompts["collate_eval"]:
        batch = [cases[i]["eval"] for i in col["batch_case_indices"]]
        out = P.eval_collate(batch, col["llm_max_length"])
        for k in ("input_ids", "attention_mask", "labels"):
            assert out[k].tolist() == col["out"][k], k
        seqs, labels = P.eval_pack(batch, col["llm_max_length"])
        for row, m, s in zip(out["input_ids"], out["attention_mask"], seqs):
            assert row[m.astype(bool)].tolist() == s.tolist()  # packed == unpadded rows
        assert labels.tolist() == [r[0] for r in col["out"]["labels"]]


def test_verbalizer_ids_and_gather(golden_dir):
    z = np.load(os.path.join(golden_dir, "verbalizer.npz"))
    v = ManualVerbalizer(tokenizer=FakeTokenizer(), prefix="", post_log_softmax=False, classes=list(range(20)),
                         label_words={i: chr(ord("A") + i) for i in range(20)})
    assert v.label_words_ids.shape == z["label_words_ids"].shape
    assert np.array_equal(v.label_words_ids, z["label_words_ids"])
    assert np.array_equal(v.process_logits(z["logits"]), z["scores"])
    import torch

    assert np.array_equal(v.process_logits(torch.from_numpy(z["logits"])).numpy(), z["scores"])


def test_overall_metric_merge(golden_dir):
    from llamarec_amd.rerank import merge_overall_metrics

    g = json.load(open(os.path.join(golden_dir, "merge.json")))
    overall = merge_overall_metrics(g["subset_in"], g["test_retrieval"])
    assert list(overall.keys()) == list(g["overall_metrics_json"].keys())
    for k, v in g["overall_metrics_json"].items():
        assert abs(overall[k] - v) < 1e-12


def test_prompts_and_label_ids_with_a_real_hf_tokenizer(golden_dir, prompts):
    """Same cases as above through a REAL HF fast tokenizer (tests/local_tokenizer.py, built offline): sub-word
    title truncation + convert_tokens_to_string, BOS, HF's left truncation (BOS survives, the oldest tokens go),
    and the verbalizer's label-word ids -- against what the reference's own functions produced with it."""
    from llamarec_amd.verb import ManualVerbalizer
    from tests.gen_goldens_llm import _Recording
    from tests.local_tokenizer import build_llama_like_tokenizer

    g = json.load(open(os.path.join(golden_dir, "prompts_hf.json")))
    for c in g["cases"]:
        tok = _Recording(build_llama_like_tokenizer())
        out = P.seq_to_token_ids(c["seq"], c["candidates"], c["label"], prompts["titles"], tok,
                                 max_title_len=c["llm_max_title_len"], max_text_len=c["llm_max_text_len"])
        assert tok.seen_texts[-1] == c["prompt_eval"]
        assert list(out["input_ids"]) == c["eval"]["input_ids"]
        assert list(out["attention_mask"]) == c["eval"]["attention_mask"]
        assert out["labels"] == c["eval"]["labels"]
        assert len(out["input_ids"]) <= c["llm_max_text_len"] and out["input_ids"][0] == tok.bos_token_id
    # a title longer than 4 sub-word tokens really was cut, and a truncated prompt kept its BOS and its tail
    short = [c for c in g["cases"] if c["llm_max_text_len"] == 48]
    assert short and all(len(c["eval"]["input_ids"]) == 48 for c in short)
    verb = ManualVerbalizer(tokenizer=build_llama_like_tokenizer(), prefix="", post_log_softmax=False,
                            classes=list(range(20)), label_words={i: chr(ord("A") + i) for i in range(20)})
    ref_ids = np.asarray(g["label_words_ids"])
    assert ref_ids.shape[:2] == (20, 1)
    assert list(verb.label_token_ids) == ref_ids[:, 0, 0].tolist()


def test_lru_datasets_match_reference(golden_dir):
    """llamarec_amd.data mirrors of LRUTrainDataset / LRUValidDataset / LRUTestDataset (dataloader/lru.py:92-180)
    against every sample the reference classes produced (tests/gen_goldens_train.py G9)."""
    from llamarec_amd import data as D

    g = json.load(open(os.path.join(golden_dir, "lru_datasets.json")))
    ds = {k: {int(u): v for u, v in g[k].items()} for k in ("train", "val", "test")}
    for c in g["cases"]:
        L = c["max_len"]
        seqs = D.lru_train_sequences(ds, L, c["sliding_window_size"])
        tok, lab = D.lru_train_batch(seqs, L)
        assert len(seqs) == len(c["train_samples"])
        assert tok.tolist() == [s[0] for s in c["train_samples"]] and lab.tolist() == [s[1] for s in c["train_samples"]]
        users, ids, labels = D.lru_eval_arrays(ds, "val", L)
        assert users == c["val_users"] and ids.tolist() == [s[0] for s in c["val"]]
        assert labels.tolist() == [s[1][:1] for s in c["val"]]
        users, ids, labels = D.lru_eval_arrays(ds, "test", L)
        assert users == c["test_users"] and ids.tolist() == [s[0] for s in c["test"]]
        assert labels.tolist() == [s[1][:1] for s in c["test"]]


def test_train_prompt_and_labels(prompts):
    """Train branch (dataloader/llm.py:33-61): prompt + answer letter, EOS appended, labels[:-2] = -100."""
    titles = {int(k): v for k, v in prompts["titles"].items()}
    for c in prompts["cases"]:
        tok = FakeTokenizer()
        out = P.seq_to_token_ids_train(c["seq"], c["candidates"], c["label"], titles, tok,
                                       max_title_len=c["llm_max_title_len"], max_text_len=c["llm_max_text_len"])
        assert tok.seen_texts[-1] == c["prompt_train"]
        for k in ("input_ids", "attention_mask", "labels"):
            assert out[k] == c["train"][k], k
        assert out["labels"][-3] == -100 and out["labels"][-2] != -100


def test_llm_train_dataset_and_collate_match_reference(golden_dir):
    """LLMTrainDataset (dataloader/llm.py:236-283: prefix expansion, negative sampling from numpy's legacy stream,
    candidate shuffle) and the train collate (trainer/llm.py:15-60) in its packed form."""
    from types import SimpleNamespace

    from llamarec_amd.rank_train import LLMTrainSamples, loss_rows_and_targets

    g = json.load(open(os.path.join(golden_dir, "llm_train_dataset.json")))
    u2seq = {int(k): v for k, v in g["u2seq"].items()}
    titles = {int(k): v for k, v in g["titles"].items()}
    for c in g["cases"]:
        args = SimpleNamespace(num_items=40, llm_negative_sample_size=c["llm_negative_sample_size"],
                               llm_max_history=c["llm_max_history"], llm_max_title_len=32,
                               llm_max_text_len=c["llm_max_text_len"])
        ds = LLMTrainSamples(args, u2seq, titles, FakeTokenizer(), rng=np.random.RandomState(c["seed"]))
        assert ds.all_seqs == c["all_seqs"]
        samples = [ds[i] for i in range(len(ds))]
        for got, ref in zip(samples, c["samples"]):
            for k in ("input_ids", "attention_mask", "labels"):
                assert [int(x) for x in got[k]] == ref[k], k
        # packed form of the reference's left-padded train batch
        seqs, labels = P.train_pack(samples[:4], c["collate_max_length"], eos_token_id=2)
        ref = c["collate_first4"]
        for b in range(4):
            n = int(np.sum(ref["attention_mask"][b]))
            assert seqs[b].tolist() == ref["input_ids"][b][-n:]
            assert labels[b].tolist() == ref["labels"][b][-n:]
            assert all(x == -100 for x in ref["labels"][b][:-n])         # padding never carries a label
        rows, tgts = loss_rows_and_targets(seqs, labels)
        assert len(rows) == 2 * 4                                         # answer letter + EOS per prompt
        cu = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
        for b in range(4):
            assert rows[2 * b] == cu[b + 1] - 3 and rows[2 * b + 1] == cu[b + 1] - 2
            assert tgts[2 * b] == seqs[b][-2] and tgts[2 * b + 1] == seqs[b][-1] == 2


def test_linear_schedule_is_hf_linear_with_warmup():
    from llamarec_amd.rank_train import linear_schedule

    f = linear_schedule(100, 1000)
    assert f(0) == 0.0 and f(50) == 0.5 and f(100) == 1.0 and abs(f(550) - 0.5) < 1e-12 and f(1000) == 0.0
    try:
        from transformers import get_linear_schedule_with_warmup
        import torch

        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        sch = get_linear_schedule_with_warmup(opt, 100, 1000)
        for step in range(0, 1000, 37):
            assert abs(sch.lr_lambdas[0](step) - f(step)) < 1e-12
    except ImportError:
        pass


def test_llm_valid_and_test_datasets_match_reference(golden_dir):
    """LLMValidDataset / LLMTestDataset (dataloader/llm.py:286-387): history = train[-H:] (validation) or
    (train + val)[-H:] (test), the retriever's candidates in their order, eval tokenisation."""
    from types import SimpleNamespace

    from llamarec_amd.rerank import build_test_items, build_val_items

    g = json.load(open(os.path.join(golden_dir, "llm_train_dataset.json")))
    ev = g["eval_datasets"]
    titles = {int(k): v for k, v in g["titles"].items()}
    dataset = {"train": {int(k): v for k, v in ev["train"].items()}, "val": {int(k): v for k, v in ev["val"].items()},
               "test": {int(k): v for k, v in ev["test"].items()}, "meta": titles}
    retrieved = {k: ev[k] for k in ("val_users", "val_candidates", "test_users", "test_candidates")}
    for c in ev["cases"]:
        args = SimpleNamespace(llm_max_history=c["llm_max_history"], llm_max_title_len=c["llm_max_title_len"],
                               llm_max_text_len=1536, llm_system_template=None, llm_input_template=None)
        for got, ref in ((build_val_items(dataset, retrieved, FakeTokenizer(), args), c["val"]),
                         (build_test_items(dataset, retrieved, FakeTokenizer(), args), c["test"])):
            assert len(got) == len(ref)
            for a, b in zip(got, ref):
                assert a["input_ids"] == b["input_ids"] and a["attention_mask"] == b["attention_mask"]
                assert a["labels"] == b["labels"]


def test_flag_defaults_match_reference_config(golden_dir):
    """Every flag this implementation shares with the reference's parser has the reference's value after
    `set_template` (config.py:12-148), for each dataset / model code of BASELINE.json's configs."""
    from llamarec_amd import config as cfg

    g = json.load(open(os.path.join(golden_dir, "config_defaults.json")))
    shared = 0
    for key, ref in g.items():
        mc, ds = key.split("/")
        mine = vars(cfg.parse(["--dataset_code", ds], model_code=mc))
        for k, v in ref.items():
            if k in mine:
                shared += 1
                assert mine[k] == v, (key, k, mine[k], v)
    assert shared >= 6 * 40


def test_train_ranker_builds_its_replica_on_the_local_rank_device(tmp_path, monkeypatch):
    """Data parallel = one replica per GPU (train_ranker.py:46-47 of the reference: device_map {"": process_index}). Both
    constructors must receive cuda:<LOCAL_RANK>; everything downstream (workspace, LoRA gradient buffer, histograms handed
    to the all-reduce) takes model.device."""
    import pickle

    import train_ranker
    from llamarec_amd import llm

    pickle.dump({"val_users": [], "val_candidates": [], "test_users": [], "test_candidates": [],
                 "test_retrieval": {"original_size": 1, "retrieval_size": 0, "non_retrieval_metrics": {}}},
                open(tmp_path / "retrieved.pkl", "wb"))
    seen = {}

    class Stop(Exception):
        pass

    def fake_from_state_dict(cls, state_dict, config, device="cuda:0", lora=None, nf4=False):
        seen["from_state_dict"] = device
        raise Stop

    def fake_from_pretrained(cls, path, device="cuda:0", adapter_path=None, load_in_4bit=False):
        seen["from_pretrained"] = device
        raise Stop

    monkeypatch.setattr(llm.LlamaRanker, "from_state_dict", classmethod(fake_from_state_dict))
    monkeypatch.setattr(llm.LlamaRanker, "from_pretrained", classmethod(fake_from_pretrained))
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("LOCAL_RANK", "3")
    with pytest.raises(Stop):
        train_ranker.main(["--dataset_code", "synthetic", "--synthetic", "--llm_retrieved_path", str(tmp_path),
                           "--export_root", str(tmp_path / "out")])
    assert seen["from_state_dict"] == "cuda:3"
    # the real-asset branch: a dataset.pkl + a local tokenizer directory, then from_pretrained(device=...)
    from llamarec_amd import data as D
    from tests.local_tokenizer import build_llama_like_tokenizer

    root = tmp_path / "data"
    path = D.preprocessed_path(str(root), "beauty", 0, 5, 5)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    pickle.dump(D.synthetic_dataset(num_users=20, num_items=50, seed=0), open(path, "wb"))
    tok_dir = tmp_path / "tok"
    build_llama_like_tokenizer().save_pretrained(str(tok_dir))
    with pytest.raises(Stop):
        train_ranker.main(["--dataset_code", "beauty", "--data_root", str(root), "--llm_retrieved_path", str(tmp_path),
                           "--llm_base_model", str(tmp_path / "hf"), "--llm_base_tokenizer", str(tok_dir),
                           "--export_root", str(tmp_path / "out2")])
    assert seen["from_pretrained"] == "cuda:3"
    # --share_gpu (several ranks rehearsed on one card) pins every rank to cuda:0
    with pytest.raises(Stop):
        train_ranker.main(["--dataset_code", "synthetic", "--synthetic", "--share_gpu", "--llm_retrieved_path", str(tmp_path),
                           "--export_root", str(tmp_path / "out3")])
    assert seen["from_state_dict"] == "cuda:0"


def test_lora_trainer_refuses_a_dataset_smaller_than_one_optimizer_step():
    from types import SimpleNamespace

    from llamarec_amd.rank_train import LoraRankerTrainer

    args = SimpleNamespace(lora_micro_batch_size=4, train_batch_size=8, lora_max_steps=0, lora_num_epochs=1, warmup_steps=0)
    with pytest.raises(ValueError, match="fewer than one optimizer step"):
        LoraRankerTrainer(args, engine=None, train_samples=list(range(15)), val_items=[], verbalizer=None,
                          export_root=None, rank=0, world=2)
    LoraRankerTrainer(args, engine=None, train_samples=list(range(16)), val_items=[], verbalizer=None,
                      export_root=None, rank=0, world=2)       # exactly one step: fine


def _undefined_names(path):
    """Names a function loads that neither it, an enclosing function, the module's top level nor builtins bind (a poor
    man's pyflakes: the entry-point scripts import inside functions, and a branch that only runs on the GPU box must not
    hide a NameError)."""
    import ast
    import builtins

    tree = ast.parse(open(path).read())
    scopes = (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda, ast.ClassDef)

    def own_nodes(scope):
        """nodes of this scope's body, not descending into nested functions / classes (which are yielded themselves)"""
        stack = list(ast.iter_child_nodes(scope))
        while stack:
            n = stack.pop()
            yield n
            if not isinstance(n, scopes):
                stack.extend(ast.iter_child_nodes(n))

    def bound(scope):
        names = set()
        if isinstance(scope, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = scope.args
            names.update(x.arg for x in a.posonlyargs + a.args + a.kwonlyargs)
            names.update(x.arg for x in (a.vararg, a.kwarg) if x)
        for n in own_nodes(scope):
            if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                names.add(n.name)
            elif isinstance(n, (ast.Import, ast.ImportFrom)):
                names.update((x.asname or x.name).split(".")[0] for x in n.names)
            elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                names.add(n.id)
            elif isinstance(n, ast.ExceptHandler) and n.name:
                names.add(n.name)
            elif isinstance(n, (ast.Global, ast.Nonlocal)):
                names.update(n.names)
        return names

    missing = []

    def visit(scope, visible, label):
        local = visible | bound(scope)
        for n in own_nodes(scope):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in local:
                missing.append((label, n.id, n.lineno))
            elif isinstance(n, scopes):
                # a class body does not lend its names to nested functions; functions and lambdas do
                visit(n, visible if isinstance(scope, ast.ClassDef) else local, getattr(n, "name", "<lambda>"))

    visit(tree, set(dir(builtins)) | {"__file__", "__name__", "__doc__"}, "<module>")
    return missing


def _python_sources():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = ["bench.py", "__graft_entry__.py", "train_ranker.py", "train_retriever.py"]
    files += sorted(os.path.join("llamarec_amd", f) for f in os.listdir(os.path.join(root, "llamarec_amd")) if f.endswith(".py"))
    files += sorted(os.path.join("tools", f) for f in os.listdir(os.path.join(root, "tools")) if f.endswith(".py"))
    return root, files


@pytest.mark.parametrize("script", _python_sources()[1])
def test_sources_have_no_undefined_names(script):
    assert _undefined_names(os.path.join(_python_sources()[0], script)) == []


def test_lazy_eval_items_equal_the_eager_ones_and_stream_every_user_once(golden_dir):
    """rerank.LazyEvalItems (shard first, per-item title cache, one batched tokenizer call per chunk) builds exactly the
    items of build_val_items / build_test_items -- against the reference's golden samples with the FakeTokenizer and
    against the eager builder with a REAL HF fast tokenizer (sub-word title truncation, batch encoding) -- and
    stream_token_budget_batches hands every user of a shard to the GPU loop exactly once, in token-budget batches."""
    from types import SimpleNamespace

    from llamarec_amd import data as D
    from llamarec_amd.rerank import LazyEvalItems, build_test_items, build_val_items, stream_token_budget_batches
    from tests.local_tokenizer import build_llama_like_tokenizer

    g = json.load(open(os.path.join(golden_dir, "llm_train_dataset.json")))
    ev = g["eval_datasets"]
    titles = {int(k): v for k, v in g["titles"].items()}
    dataset = {"train": {int(k): v for k, v in ev["train"].items()}, "val": {int(k): v for k, v in ev["val"].items()},
               "test": {int(k): v for k, v in ev["test"].items()}, "meta": titles}
    retrieved = {k: ev[k] for k in ("val_users", "val_candidates", "test_users", "test_candidates")}
    for c in ev["cases"]:
        args = SimpleNamespace(llm_max_history=c["llm_max_history"], llm_max_title_len=c["llm_max_title_len"],
                               llm_max_text_len=1536, llm_system_template=None, llm_input_template=None)
        for split in ("val", "test"):
            lazy = LazyEvalItems(dataset, retrieved, FakeTokenizer(), args, split=split)
            got = lazy.build(0, len(lazy))
            assert len(got) == len(c[split])
            for a, b in zip(got, c[split]):
                assert a["input_ids"] == b["input_ids"] and a["attention_mask"] == b["attention_mask"]
                assert a["labels"] == b["labels"]

    # a bigger fabricated dataset through a real fast tokenizer: titles long enough to be truncated at 6 sub-word tokens
    tok = build_llama_like_tokenizer()
    ds = D.synthetic_dataset(num_users=150, num_items=400, seed=3, title_words=7)
    rng = np.random.default_rng(0)
    users = sorted(ds["train"].keys())
    cands = []
    for u in users:
        pool = [i for i in rng.choice(400, size=40, replace=False) + 1 if i != ds["test"][u][0]][:19]
        cc = pool + [ds["test"][u][0]]
        rng.shuffle(cc)
        cands.append([int(x) for x in cc])
    retrieved = {"test_users": users, "test_candidates": cands}
    args = SimpleNamespace(llm_max_history=20, llm_max_title_len=6, llm_max_text_len=300, llm_system_template=None,
                           llm_input_template=None)
    eager = build_test_items(ds, retrieved, tok, args)
    lazy = LazyEvalItems(ds, retrieved, tok, args, split="test")
    got = lazy.build(0, len(lazy))
    assert [a["input_ids"] for a in got] == [b["input_ids"] for b in eager]
    assert [a["labels"] for a in got] == [b["labels"] for b in eager]
    assert any(len(b["input_ids"]) == 300 for b in eager)            # left truncation happened somewhere
    # the length estimate the shards are cut by is close to the real token count (balance, not exactness)
    args_long = SimpleNamespace(**{**vars(args), "llm_max_text_len": 1536})
    est = LazyEvalItems(ds, retrieved, tok, args_long, split="test").estimate_lengths().astype(np.float64)
    real = np.array([len(b["input_ids"]) for b in build_test_items(ds, retrieved, tok, args_long)], np.float64)
    assert abs(est.sum() / real.sum() - 1.0) < 0.25 and np.corrcoef(est, real)[0, 1] > 0.8, (est[:8], real[:8])
    # streaming: users [20, 131) in chunks of 16, budget 2000 tokens
    seen, n_batches = [], 0
    for seqs, labels in stream_token_budget_batches(lazy, 20, 131, 2000, 300, chunk=16, depth=2):
        assert sum(len(s) for s in seqs) <= 2000 + 300
        for s, l in zip(seqs, labels):
            seen.append((tuple(int(x) for x in s), int(l)))
        n_batches += 1
    want = [(tuple(b["input_ids"][-300:]), b["labels"]) for b in eager[20:131]]
    assert sorted(seen) == sorted(want) and len(seen) == 111 and n_batches < 30
    # a consumer that leaves early (error downstream, closed generator) must not strand the producer on a full queue
    import threading

    gen = stream_token_budget_batches(lazy, 0, 150, 600, 300, chunk=8, depth=1)
    next(gen)
    gen.close()
    assert not any(t.name == "llamarec-tokenize" and t.is_alive() for t in threading.enumerate())


def test_bench_expected_metrics_helper_matches_the_oracle_formulas():
    """bench.py's host-side expectation for the planted labels (metrics_of_ranks: Recall / MRR / NDCG from 0-based ranks,
    -1 = not retrieved) equals the oracle's rank-metric sums on ranked lists that realise those ranks."""
    import bench
    from oracle import lru_oracle as O

    rng = np.random.default_rng(0)
    n = 500
    pos = (np.arange(n) % bench.PLANT_PERIOD).astype(np.int64)
    pos[pos >= 50] = -1
    ranked = np.stack([rng.permutation(1000)[:50] + 1 for _ in range(n)]).astype(np.int32)
    labels = np.where(pos >= 0, ranked[np.arange(n), np.maximum(pos, 0)], 5000).astype(np.int64)
    got = bench.metrics_of_ranks(pos, [10, 50])
    sums = O.rank_metric_sums(ranked, labels, [50, 10])
    for j, k in enumerate((50, 10)):
        for c, name in enumerate(("Recall", "MRR", "NDCG")):
            assert abs(got[f"{name}@{k}"] - sums[j, c] / n) < 1e-12
    assert got["NDCG@10"] > 0.05 and got["Recall@50"] == pytest.approx(50 / 60 * (n // 60 * 60) / n + (n % 60 if n % 60 < 50 else 50) / n, abs=1e-12)


def test_bench_refuses_rank_mismatches_without_touching_a_gpu():
    """bench.py's two refusal paths are decided before any HIP call (and before torch is imported in the first case), so they
    run here: (1) under a launcher whose WORLD_SIZE differs from --gpus -> exit code 2 with a message and no result line;
    (2) plain `python bench.py --gpus N` on a node that exposes fewer than N GPUs (this container: 0) -> exit code 2, "refusing",
    no result line -- a smaller run is never reported as n_gpus = N."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2"], cwd=repo, capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr and "{" not in r.stdout
    import torch

    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(max(n, 2))], cwd=repo, capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and "{" not in r.stdout


def test_bench_expected_metrics_helpers():
    """The float64 expectations bench.py compares the GPU's histogram metrics with (metrics_of_ranks, ndcg_at_10) against
    trainer/utils.py:43-90's formulas for one relevant item per row, by hand."""
    import importlib.util

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    pos = np.array([0, 4, 9, 10, 49, -1])          # 0-based ranks; -1 = not retrieved
    m = b.metrics_of_ranks(pos, [10, 50])
    assert abs(m["Recall@10"] - 3 / 6) < 1e-15 and abs(m["Recall@50"] - 5 / 6) < 1e-15
    assert abs(m["MRR@10"] - (1 + 1 / 5 + 1 / 10) / 6) < 1e-15
    assert abs(m["NDCG@10"] - (1 / np.log2(2) + 1 / np.log2(6) + 1 / np.log2(11)) / 6) < 1e-15
    assert abs(m["NDCG@50"] - (1 / np.log2(2) + 1 / np.log2(6) + 1 / np.log2(11) + 1 / np.log2(12) + 1 / np.log2(51)) / 6) < 1e-15
    ranked = np.array([[5, 3, 9], [1, 2, 3]])
    assert abs(b.ndcg_at_10(ranked, [9, 7]) - (1 / np.log2(4) + 0.0) / 2) < 1e-15


@pytest.mark.parametrize("local", [0, 3, 7])
def test_device_placement_census_on_a_mock_eight_gpu_node(monkeypatch, local):
    """bench.py's placement census (llamarec_amd/dist.py device_placement) on a mock 8-GPU node: under WORLD_SIZE = 8,
    LOCAL_RANK = r a rank reports (1, 0) when HIP's current device is cuda:r and the allocator holds nothing elsewhere, and
    names the stray bytes / the wrong current device otherwise -- what the driver's 8-GPU line then carries in `placement`."""
    import torch

    from llamarec_amd import dist as D

    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", str(local))
    monkeypatch.setenv("LOCAL_RANK", str(local))
    assert D.env_world() == (local, 8, local)
    held = {d: 0 for d in range(8)}
    held[local] = 5 << 30
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: local)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda d=None: held[d])
    assert D.device_placement(local) == (1, 0)
    held[(local + 1) % 8] = 4096                       # a tensor built with a default "cuda:0"-style device
    assert D.device_placement(local) == (1, 4096)
    held[(local + 1) % 8] = 0
    monkeypatch.setattr(torch.cuda, "current_device", lambda: (local + 1) % 8)   # set_device never ran
    assert D.device_placement(local) == (0, 0)
