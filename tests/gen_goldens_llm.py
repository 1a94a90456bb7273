"""Stage-2 golden sections (G4-G7) of tests/gen_goldens.py -- runs the reference, stores data only."""
from __future__ import annotations

import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

from llamarec_amd.synth import synth_llama_state
from tests.fake_tokenizer import FakeTokenizer

TITLES = {
    1: "Toy Story (1995)", 2: "Heat (1995)", 3: "A very long title that certainly exceeds the limit of tokens",
    4: "Casino (1995)", 5: "Se7en (1995)", 6: "Usual Suspects, The (1995)", 7: "Braveheart (1995)",
    8: "Apollo 13 (1995)", 9: "Léon: The Professional (1994)", 10: "Pulp  Fiction (1994)",
}


def g4(functions_from, OUT, REF):
    # dataloader.llm imports config, which parses sys.argv at import and opens its template
    # relative to the cwd (config.py:274, dataloader/utils.py:14)
    argv, cwd = sys.argv, os.getcwd()
    sys.argv = ["x"]
    os.chdir(REF)
    sys.path.insert(0, REF)
    try:
        import dataloader.llm as DL  # noqa
        from dataloader.utils import Prompter
        import config as ref_config

        prompter = Prompter()
        ref_args = ref_config.args
    finally:
        sys.argv = argv
        os.chdir(cwd)
    cases = []
    for title_len, text_len in ((32, 1536), (3, 1536), (3, 40)):
        args = SimpleNamespace(llm_max_title_len=title_len, llm_max_text_len=text_len,
                               llm_system_template=ref_args.llm_system_template,
                               llm_input_template=ref_args.llm_input_template, llm_train_on_inputs=False)
        for seq, cands, label in (([1, 2, 3], [4, 5, 2], 5), ([7], [8, 9, 10, 6], 8),
                                  ([1, 2, 3, 4, 5, 6, 7, 8, 9], [10, 1], 10)):
            tok = FakeTokenizer()
            ev = DL.seq_to_token_ids(args, seq, cands, label, TITLES, tok, prompter, eval=True)
            prompt_eval = tok.seen_texts[-1]
            tok2 = FakeTokenizer()
            tr = DL.seq_to_token_ids(args, seq, cands, label, TITLES, tok2, prompter, eval=False)
            cases.append({"llm_max_title_len": title_len, "llm_max_text_len": text_len, "seq": seq,
                          "candidates": cands, "label": label, "prompt_eval": prompt_eval,
                          "eval": {k: ev[k] for k in ("input_ids", "attention_mask", "labels")},
                          "prompt_train": tok2.seen_texts[-1],
                          "train": {k: tr[k] for k in ("input_ids", "attention_mask", "labels")}})
    fns = functions_from(os.path.join(REF, "trainer", "llm.py"), ["llama_collate_fn_w_truncation"],
                         {"torch": torch})
    collates = []
    batch = [c["eval"] for c in cases[:3]]
    for max_len in (1536, 45):
        out = fns["llama_collate_fn_w_truncation"](max_len, eval=True)(batch)
        collates.append({"llm_max_length": max_len, "batch_case_indices": [0, 1, 2],
                         "out": {k: v.tolist() for k, v in out.items()}})
    json.dump({"titles": {str(k): v for k, v in TITLES.items()},
               "system_template": ref_args.llm_system_template,
               "input_template": ref_args.llm_input_template, "cases": cases, "collate_eval": collates},
              open(os.path.join(OUT, "prompts.json"), "w"), indent=1, ensure_ascii=False)


class _Recording:
    """Delegates to a real tokenizer and remembers the texts it was called with."""

    def __init__(self, tok):
        self._tok, self.seen_texts = tok, []

    def __getattr__(self, name):
        return getattr(self._tok, name)

    def __call__(self, text, **kw):
        self.seen_texts.append(text)
        return self._tok(text, **kw)


def g4hf(functions_from, load_by_path, OUT, REF):
    """G4 again with a REAL HF fast tokenizer built offline (tests/local_tokenizer.py): sub-word title truncation,
    metaspace decoding, BOS handling and HF's left truncation are the reference's, not a stand-in's. Also the
    verbalizer's label-word ids for that tokenizer (trainer/verb.py:486-522 via demo/verb.py, identical class)."""
    from tests.local_tokenizer import build_llama_like_tokenizer

    argv, cwd = sys.argv, os.getcwd()
    sys.argv = ["x"]
    os.chdir(REF)
    sys.path.insert(0, REF)
    try:
        import dataloader.llm as DL  # noqa
        from dataloader.utils import Prompter
        import config as ref_config

        prompter = Prompter()
        ref_args = ref_config.args
    finally:
        sys.argv = argv
        os.chdir(cwd)
    cases = []
    for title_len, text_len in ((32, 1536), (3, 1536), (4, 48)):
        args = SimpleNamespace(llm_max_title_len=title_len, llm_max_text_len=text_len,
                               llm_system_template=ref_args.llm_system_template,
                               llm_input_template=ref_args.llm_input_template, llm_train_on_inputs=False)
        for seq, cands, label in (([1, 2, 3], [4, 5, 2], 5), ([7], [8, 9, 10, 6], 8),
                                  ([1, 2, 3, 4, 5, 6, 7, 8, 9], [10, 1], 10)):
            tok = _Recording(build_llama_like_tokenizer())
            ev = DL.seq_to_token_ids(args, seq, cands, label, TITLES, tok, prompter, eval=True)
            cases.append({"llm_max_title_len": title_len, "llm_max_text_len": text_len, "seq": seq,
                          "candidates": cands, "label": label, "prompt_eval": tok.seen_texts[-1],
                          "eval": {k: [int(x) for x in ev[k]] if isinstance(ev[k], (list, tuple)) else int(ev[k])
                                   for k in ("input_ids", "attention_mask", "labels")}})
    V = load_by_path("ref_demo_verb_hf", os.path.join(REF, "demo", "verb.py"))
    verb = V.ManualVerbalizer(tokenizer=build_llama_like_tokenizer(), prefix="", post_log_softmax=False,
                              classes=list(range(20)), label_words={i: chr(ord("A") + i) for i in range(20)})
    json.dump({"cases": cases, "label_words_ids": verb.label_words_ids.detach().numpy().tolist()},
              open(os.path.join(OUT, "prompts_hf.json"), "w"), indent=1, ensure_ascii=False)


LLAMA_CONFIGS = {
    # name: (vocab, hidden, inter, layers, heads, kv_heads)
    "tiny_hd16": (320, 64, 128, 2, 4, 4),
    "tiny_hd128": (320, 256, 512, 2, 2, 2),
    "tiny_gqa": (320, 128, 256, 2, 4, 2),
}


def hf_cfg_dict(name):
    v, d, f, nl, nh, nkv = LLAMA_CONFIGS[name]
    return dict(vocab_size=v, hidden_size=d, intermediate_size=f, num_hidden_layers=nl,
                num_attention_heads=nh, num_key_value_heads=nkv, max_position_embeddings=256,
                rms_norm_eps=1e-5, rope_theta=10000.0)


def g5(OUT, REF):
    sys.path.insert(0, REF)
    import model.llm as ML  # noqa: F401  (patches LlamaForCausalLM.forward at import, model/llm.py:145)
    from transformers import LlamaConfig, LlamaForCausalLM

    for ci, name in enumerate(LLAMA_CONFIGS):
        cd = hf_cfg_dict(name)
        rope = {"rope_parameters": {"rope_type": "default", "rope_theta": cd["rope_theta"]}}
        kw = {k: v for k, v in cd.items() if k != "rope_theta"}
        try:
            cfg = LlamaConfig(**kw, **rope, attention_bias=False, mlp_bias=False,
                              tie_word_embeddings=False, attn_implementation="eager")
        except TypeError:
            cfg = LlamaConfig(**cd, attention_bias=False, mlp_bias=False, tie_word_embeddings=False,
                              attn_implementation="eager")
        seed = 100 + ci
        sd = synth_llama_state(cd, seed)
        model = LlamaForCausalLM(cfg).eval()
        missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not [k for k in missing.missing_keys if "rotary" not in k], missing
        rng = np.random.default_rng(seed)
        lens = [37, 5, 64, 1, 20]
        T = max(lens)
        ids = np.zeros((len(lens), T), np.int64)
        mask = np.zeros((len(lens), T), np.int64)
        for b, n in enumerate(lens):
            ids[b, T - n:] = rng.integers(3, cd["vocab_size"], size=n)
            ids[b, T - n] = 1
            mask[b, T - n:] = 1
        labels = np.zeros((len(lens), 1), np.int64)
        with torch.no_grad():
            o32 = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                        labels=torch.from_numpy(labels))
            loss = float(o32.loss)
            l32 = o32.logits.numpy()
            # unpadded, one prompt at a time (varlen execution is legal: SURVEY.md 8(a) a15)
            lun = np.stack([model(input_ids=torch.from_numpy(ids[b:b + 1, T - n:])).logits[0].numpy()
                            for b, n in enumerate(lens)])
            mb = model.to(torch.bfloat16)
            lbf = mb(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).logits
            assert lbf.dtype == torch.float32
            lbf = lbf.numpy()
        assert l32.shape == (len(lens), cd["vocab_size"]) and loss == -1.0
        np.savez_compressed(os.path.join(OUT, f"llama_{name}.npz"), config=json.dumps(cd), weight_seed=seed,
                            input_ids=ids, attention_mask=mask, lens=np.array(lens),
                            logits_fp32=l32, logits_fp32_unpadded=lun, logits_bf16=lbf, eval_loss=loss)


def g6(load_by_path, OUT, REF):
    V = load_by_path("ref_demo_verb", os.path.join(REF, "demo", "verb.py"))
    tok = FakeTokenizer()
    verb = V.ManualVerbalizer(tokenizer=tok, prefix="", post_log_softmax=False, classes=list(range(20)),
                              label_words={i: chr(ord("A") + i) for i in range(20)})
    rng = np.random.default_rng(6)
    logits = rng.standard_normal((5, tok.vocab_size)).astype(np.float32)
    out = verb.process_logits(torch.from_numpy(logits)).numpy()
    ids = verb.label_words_ids.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "verbalizer.npz"), logits=logits, scores=out, label_words_ids=ids,
                        words_ids_mask=verb.words_ids_mask.numpy(), label_words_mask=verb.label_words_mask.numpy())


def g7(methods_from, OUT, REF):
    ns = {"os": os, "json": json, "print": lambda *a, **k: None}
    fns = methods_from(os.path.join(REF, "trainer", "llm.py"), "LLMTrainer", ["test"], ns)
    subset = {"test_Recall@10": 0.81, "test_MRR@10": 0.4321, "test_NDCG@10": 0.5234, "test_Recall@5": 0.7,
              "test_MRR@5": 0.41, "test_NDCG@5": 0.49, "test_Recall@1": 0.25, "test_MRR@1": 0.25,
              "test_NDCG@1": 0.25, "test_loss": -1.0, "test_runtime": 12.5}
    test_retrieval = {
        "original_size": 610, "retrieval_size": 137,
        "original_metrics": {f"{m}@{k}": 0.01 * k for m in ("Recall", "MRR", "NDCG") for k in (1, 5, 10, 20, 50)},
        "retrieval_metrics": {f"{m}@{k}": 0.02 * k for m in ("Recall", "MRR", "NDCG") for k in (1, 5, 10, 20, 50)},
        "non_retrieval_metrics": {**{f"{m}@{k}": 0.0 for m in ("Recall", "MRR", "NDCG") for k in (1, 5, 10, 20)},
                                  "Recall@50": 0.2, "MRR@50": 0.006, "NDCG@50": 0.04},
    }
    with tempfile.TemporaryDirectory() as td:
        me = SimpleNamespace(export_root=td, predict=lambda test_dataset=None: SimpleNamespace(metrics=dict(subset)))
        ret = fns["test"](me, test_retrieval)
        sub = json.load(open(os.path.join(td, "subset_metrics.json")))
        overall = json.load(open(os.path.join(td, "overall_metrics.json")))
    json.dump({"subset_in": subset, "test_retrieval": test_retrieval, "returned": ret,
               "subset_metrics_json": sub, "overall_metrics_json": overall},
              open(os.path.join(OUT, "merge.json"), "w"), indent=1)


def run(which, load_by_path, methods_from, functions_from, OUT, REF):
    if "g4" in which:
        g4(functions_from, OUT, REF)
        g4hf(functions_from, load_by_path, OUT, REF)
    if "g5" in which:
        g5(OUT, REF)
    if "g6" in which:
        g6(load_by_path, OUT, REF)
    if "g7" in which:
        g7(methods_from, OUT, REF)
