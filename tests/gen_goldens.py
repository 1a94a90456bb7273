"""Generate tests/golden/*.npz|json by RUNNING the reference (/root/reference) in this container.

Only data leaves this script: inputs, weights (or the seeds that regenerate them) and the
reference's outputs. No reference source text is stored. The reference is imported read-only
(PYTHONDONTWRITEBYTECODE=1, scratch cwd). Run from the repo root:

    PYTHONDONTWRITEBYTECODE=1 python tests/gen_goldens.py

Sections (SURVEY.md 8(c) "Golden vectors to capture"):
  G1 lru_v300.npz        LRURec weights + ids -> last-position scores (model/lru.py)
  G2 metrics.npz         absolute_recall_mrr_ndcg_for_ks / batch wrapper (trainer/utils.py)
  G3 candidates.json     LRUTrainer.calculate_metrics / generate_candidates (trainer/lru.py)
  G4 prompts.json        seq_to_token_ids + Prompter + eval collate (dataloader/llm.py,
                         dataloader/utils.py, trainer/llm.py) with a deterministic fake tokenizer
  G5 llama_tiny_*.npz    patched LlamaForCausalLM last-position logits (model/llm.py)
  G6 verbalizer.npz      ManualVerbalizer.process_logits (demo/verb.py == trainer/verb.py:433-614)
  G7 merge.json          LLMTrainer.test metric merge (trainer/llm.py:165-189)
"""
from __future__ import annotations

import ast
import importlib.util
import json
import os
import pickle
import sys
import tempfile
import types
import warnings
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def methods_from(path, class_name, names, namespace):
    """Compile selected methods of a reference class as free functions (run, never stored)."""
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            fns = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in names]
            mod = ast.Module(body=fns, type_ignores=[])
            exec(compile(mod, path, "exec"), namespace)
            return {n: namespace[n] for n in names}
    raise KeyError(class_name)


def functions_from(path, names, namespace):
    tree = ast.parse(open(path).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), namespace)
    return {n: namespace[n] for n in names}


# ----------------------------------------------------------------------------- G1
def make_lru(num_items, seed):
    sys.path.insert(0, REF)
    from model.lru import LRURec  # noqa: E402  (reference import)

    args = SimpleNamespace(num_items=num_items, bert_hidden_units=64, bert_num_blocks=2,
                           bert_dropout=0.2, bert_attn_dropout=0.2)
    torch.manual_seed(seed)
    m = LRURec(args).eval()
    with torch.no_grad():  # make LayerNorm affine and the item bias non-trivial
        for n, p in m.named_parameters():
            if "layer_norm" in n or n == "model.bias":
                p.add_(torch.randn_like(p) * 0.1)
    return m


def left_pad_ids(rng, B, L, V, lens):
    ids = np.zeros((B, L), np.int64)
    for b, n in enumerate(lens):
        n = min(n, L)
        if n:
            ids[b, L - n:] = rng.choice(np.arange(1, V + 1), size=n, replace=n > V)
    return ids


def g1():
    V = 300
    m = make_lru(V, 0)
    rng = np.random.default_rng(1)
    cases = {
        "L50": left_pad_ids(rng, 8, 50, V, [10, 1, 0, 50, 49, 3, 25, 50]),
        "L7": left_pad_ids(rng, 4, 7, V, [7, 7, 2, 5]),
        "L64": left_pad_ids(rng, 4, 64, V, [64, 33, 32, 1]),
        "L200": left_pad_ids(rng, 4, 200, V, [200, 100, 129, 17]),
    }
    cases["L50"][4, -1] = 0  # last id is the pad id but the rest is history
    out = {}
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.numpy()
    for name, ids in cases.items():
        with torch.no_grad():
            s = m(torch.from_numpy(ids))[:, -1, :].numpy()
        out[f"ids/{name}"] = ids
        out[f"scores_last/{name}"] = s
    np.savez_compressed(os.path.join(OUT, "lru_v300.npz"), **out)
    return m, cases


# ----------------------------------------------------------------------------- G2
def g2():
    U = load_by_path("ref_trainer_utils", os.path.join(REF, "trainer", "utils.py"))
    rng = np.random.default_rng(2)
    ks = [1, 5, 10, 20, 50]
    scores = rng.standard_normal((37, 120)).astype(np.float32)
    labels = rng.integers(0, 120, size=37)
    m1 = U.absolute_recall_mrr_ndcg_for_ks(torch.from_numpy(scores), torch.from_numpy(labels), ks)
    ranked = np.argsort(-scores, axis=1, kind="stable")[:, :50]
    m2 = U.absolute_metrics_batch_wrapper(torch.from_numpy(ranked), torch.from_numpy(labels), ks,
                                          num_classes=120, preprocessed=True, batch_size=10)
    s20 = rng.standard_normal((23, 20)).astype(np.float32)
    l20 = rng.integers(0, 20, size=23)
    m3 = U.absolute_recall_mrr_ndcg_for_ks(torch.from_numpy(s20), torch.from_numpy(l20), [1, 5, 10])
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), scores=scores, labels=labels, ranked=ranked,
                        s20=s20, l20=l20)
    json.dump({"ks": ks, "full": m1, "wrapper_preprocessed_bs10": m2, "rerank_ks": [1, 5, 10],
               "rerank": m3}, open(os.path.join(OUT, "metrics.json"), "w"), indent=1)
    return U


# ----------------------------------------------------------------------------- G3
def g3(m, U):
    V = 300
    rng = np.random.default_rng(3)
    n_users, L, B = 45, 50, 16
    lens = rng.integers(1, 60, size=n_users)
    val_ids = left_pad_ids(rng, n_users, L, V, lens)
    test_ids = left_pad_ids(rng, n_users, L, V, lens + 1)
    # labels biased towards items the model ranks high so that some users are "retrieved"
    with torch.no_grad():
        sv = m(torch.from_numpy(val_ids))[:, -1, :].numpy().copy()
        st = m(torch.from_numpy(test_ids))[:, -1, :].numpy().copy()
    def pick(s, ids):
        lab = np.zeros(len(s), np.int64)
        for u in range(len(s)):
            s[u, ids[u]] = -1e9
            s[u, 0] = -1e9
            order = np.argsort(-s[u], kind="stable")
            lab[u] = order[rng.integers(0, 40)]
        return lab
    val_lab, test_lab = pick(sv, val_ids), pick(st, test_ids)

    def loader(ids, lab):
        return [(torch.from_numpy(ids[i:i + B]), torch.from_numpy(lab[i:i + B, None]))
                for i in range(0, len(ids), B)]

    ks = [1, 5, 10, 20, 50]
    ref_args = SimpleNamespace(metric_ks=ks, num_items=V, num_users=n_users, llm_negative_sample_size=19)
    ns = {"torch": torch, "pickle": pickle, "tqdm": lambda x: x, "args": ref_args,
          "absolute_recall_mrr_ndcg_for_ks": U.absolute_recall_mrr_ndcg_for_ks,
          "absolute_metrics_batch_wrapper": U.absolute_metrics_batch_wrapper, "print": lambda *a, **k: None}
    fns = methods_from(os.path.join(REF, "trainer", "lru.py"), "LRUTrainer",
                       ["calculate_metrics", "generate_candidates"], ns)
    me = SimpleNamespace(model=m, metric_ks=ks, args=ref_args, val_loader=loader(val_ids, val_lab),
                         test_loader=loader(test_ids, test_lab), to_device=lambda b: b)
    # calculate_metrics per batch (test: exclude_history=True; validation: False -- trainer/base.py:141-143)
    with torch.no_grad():
        per_batch_test = [fns["calculate_metrics"](me, b) for b in me.test_loader]
        per_batch_val = [fns["calculate_metrics"](me, b, exclude_history=False) for b in me.val_loader]
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "retrieved.pkl")
        fns["generate_candidates"](me, path)
        retrieved = pickle.load(open(path, "rb"))
    np.savez_compressed(os.path.join(OUT, "candidates_inputs.npz"), val_ids=val_ids, test_ids=test_ids,
                        val_labels=val_lab, test_labels=test_lab)
    json.dump({"ks": ks, "batch_size": B, "num_users": n_users, "per_batch_test_metrics": per_batch_test,
               "per_batch_val_metrics_no_exclude": per_batch_val, "retrieved": retrieved},
              open(os.path.join(OUT, "candidates.json"), "w"))


def main():
    os.makedirs(OUT, exist_ok=True)
    scratch = tempfile.mkdtemp()
    os.chdir(scratch)
    which = set(sys.argv[1:]) or {"g1", "g2", "g3", "g4", "g5", "g6", "g7"}
    m = cases = U = None
    if which & {"g1", "g3"}:
        m, cases = g1()
    if which & {"g2", "g3"}:
        U = g2()
    if "g3" in which:
        g3(m, U)
    from tests import gen_goldens_llm as G  # stage-2 sections live in a second file

    G.run(which, load_by_path, methods_from, functions_from, OUT, REF)
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()
