"""Generate tests/golden/llama_lora_train_*.npz by RUNNING the reference's ranker training forward on CPU.

G10: the reference's patched `LlamaForCausalLM.forward` in TRAINING mode
(model/llm.py:116-127: shifted CrossEntropyLoss over the positions whose label is not -100), on
the batches its own collate builds (trainer/llm.py:14-58: left padding with 0 / label -100) from
samples tokenised like dataloader/llm.py:33-61 (`labels[:-2] = -100`: only the answer letter and
EOS carry a label), with LoRA r=8, alpha=32 on q_proj / v_proj (config.py:257-260,
train_ranker.py:71-79), torch autograd gradients of every LoRA matrix, gradient clipping at HF
`TrainingArguments`' default max_grad_norm = 1.0 and AdamW steps.

Third-party boundary: LoRA is **peft 0.11.1** (environment.yml:248), absent here. Its published
layer is restated below (`LoraLinear`: y = W x + (alpha / r) * B(A(dropout(x)))); the base model is
the reference's patched HF class. The reference optimiser is bitsandbytes' 8-bit AdamW
(trainer/llm.py:117, absent, block-quantised state): the goldens use torch.optim.AdamW with HF's
defaults (betas 0.9/0.999, eps 1e-8, weight_decay 0) -- what the 8-bit optimiser approximates.
Dropout is 0 in the captured steps (torch's masks cannot be reproduced elsewhere).

Only data leaves this script. Run from the repo root:
    PYTHONDONTWRITEBYTECODE=1 python tests/gen_goldens_rank_train.py
"""
from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

from llamarec_amd.synth import bf16_round, hash_uniform, synth_llama_state  # noqa: E402
from tests.gen_goldens import functions_from  # noqa: E402
from tests.gen_goldens_llm import LLAMA_CONFIGS, hf_cfg_dict  # noqa: E402

R, ALPHA = 8, 32


class LoraLinear(torch.nn.Module):
    """peft.tuners.lora.Linear.forward, restated: result = base(x) + lora_B(lora_A(dropout(x))) * scaling."""

    def __init__(self, base, a, b):
        super().__init__()
        self.base = base
        self.lora_A = torch.nn.Parameter(a)   # [r, in]
        self.lora_B = torch.nn.Parameter(b)   # [out, r]
        self.scaling = ALPHA / R

    def forward(self, x):
        res = self.base(x)
        xa = torch.nn.functional.linear(x.to(self.lora_A.dtype), self.lora_A)
        return res + (torch.nn.functional.linear(xa, self.lora_B) * self.scaling).to(res.dtype)


def lora_init(cfg, seed):
    """Deterministic, bf16-representable LoRA matrices; B is NOT zero (peft's init) so that A has a gradient."""
    d, nh, nkv = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = d // nh
    out = {}
    for l in range(cfg["num_hidden_layers"]):
        for j, (name, rows) in enumerate((("q_proj", nh * hd), ("v_proj", nkv * hd))):
            out[f"layers.{l}.{name}.lora_A"] = bf16_round(hash_uniform(seed * 77 + l * 4 + j * 2, (R, d), 0.05))
            out[f"layers.{l}.{name}.lora_B"] = bf16_round(hash_uniform(seed * 77 + l * 4 + j * 2 + 1, (rows, R), 0.05))
    return out


def build(name, seed, dtype):
    sys.path.insert(0, REF)
    import model.llm as ML  # noqa: F401  (patches LlamaForCausalLM.forward, model/llm.py:145)
    from transformers import LlamaConfig, LlamaForCausalLM

    cd = hf_cfg_dict(name)
    rope = {"rope_parameters": {"rope_type": "default", "rope_theta": cd["rope_theta"]}}
    kw = {k: v for k, v in cd.items() if k != "rope_theta"}
    try:
        cfg = LlamaConfig(**kw, **rope, attention_bias=False, mlp_bias=False, tie_word_embeddings=False,
                          attn_implementation="eager")
    except TypeError:
        cfg = LlamaConfig(**cd, attention_bias=False, mlp_bias=False, tie_word_embeddings=False,
                          attn_implementation="eager")
    sd = synth_llama_state(cd, seed)
    model = LlamaForCausalLM(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(dtype)
    for p in model.parameters():
        p.requires_grad_(False)
    lora = lora_init(cd, seed)
    params = {}
    for l, layer in enumerate(model.model.layers):
        for pn in ("q_proj", "v_proj"):
            a = torch.from_numpy(lora[f"layers.{l}.{pn}.lora_A"]).float().clone()   # fp32 masters, like peft
            b = torch.from_numpy(lora[f"layers.{l}.{pn}.lora_B"]).float().clone()
            mod = LoraLinear(getattr(layer.self_attn, pn), a, b)
            setattr(layer.self_attn, pn, mod)
            params[f"layers.{l}.{pn}.lora_A"] = mod.lora_A
            params[f"layers.{l}.{pn}.lora_B"] = mod.lora_B
    return model.train(), cd, lora, params


def make_batch(cd, seed, lens, train_on_inputs, collate):
    rng = np.random.default_rng(seed)
    samples = []
    for n in lens:
        ids = [1] + rng.integers(3, cd["vocab_size"], size=n - 2).tolist() + [2]   # BOS ... answer EOS
        labels = list(ids)
        if not train_on_inputs:
            labels[:-2] = [-100] * len(labels[:-2])                                # dataloader/llm.py:55-58
        samples.append({"input_ids": ids, "attention_mask": [1] * n, "labels": labels})
    return samples, collate(samples)


def run(name, ci):
    seed = 300 + ci
    fns = functions_from(os.path.join(REF, "trainer", "llm.py"), ["llama_collate_fn_w_truncation"], {"torch": torch})
    collate = fns["llama_collate_fn_w_truncation"](1536, eval=False)
    out = {}
    model, cd, lora, params = build(name, seed, torch.float32)
    out["config"] = json.dumps(cd)
    out["weight_seed"] = seed
    out["lora_r"], out["lora_alpha"] = R, ALPHA
    names = sorted(params)
    out["param_names"] = np.array(names)
    for n in names:
        out["init/" + n] = lora[n].copy()
    opt = torch.optim.AdamW([params[n] for n in names], lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    # two steps on different batches in the reference's label layout (letter + EOS; its collate asserts exactly
    # that, trainer/llm.py:46-48); the second with a small clip limit so that clipping really rescales
    for step, (lens, toi, limit) in enumerate((([37, 9, 64, 21], False, 1.0), ([12, 50, 5], False, 0.05))):
        samples, batch = make_batch(cd, seed * 10 + step, lens, toi, collate)
        out[f"step{step}/lens"] = np.array(lens)
        out[f"step{step}/packed_ids"] = np.concatenate([np.array(s["input_ids"]) for s in samples]).astype(np.int32)
        out[f"step{step}/packed_labels"] = np.concatenate([np.array(s["labels"]) for s in samples]).astype(np.int32)
        opt.zero_grad()
        o = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"])
        o.loss.backward()
        out[f"step{step}/loss"] = np.float32(o.loss.item())
        for n in names:
            out[f"step{step}/grad/" + n] = params[n].grad.detach().numpy().copy()
        norm = torch.nn.utils.clip_grad_norm_([params[n] for n in names], limit)
        out[f"step{step}/grad_norm"] = np.float32(float(norm))
        out[f"step{step}/clip_limit"] = np.float32(limit)
        opt.step()
        for n in names:
            out[f"step{step}/param/" + n] = params[n].detach().numpy().copy()
    # the same first step under bf16 autocast with bf16 base weights (the reference's arithmetic, bf16=True,
    # trainer/llm.py:110): calibrates the tolerance of the bf16 HIP path
    model, cd, lora, params = build(name, seed, torch.bfloat16)
    samples, batch = make_batch(cd, seed * 10, [37, 9, 64, 21], False, collate)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"])
    o.loss.backward()
    out["bf16/loss"] = np.float32(o.loss.item())
    for n in names:
        out["bf16/grad/" + n] = params[n].grad.detach().float().numpy().copy()
    path = os.path.join(OUT, f"llama_lora_train_{name}.npz")
    np.savez_compressed(path, **out)
    g32 = np.concatenate([out["step0/grad/" + n].ravel() for n in names])
    gbf = np.concatenate([out["bf16/grad/" + n].ravel() for n in names])
    print("wrote", path, os.path.getsize(path), "bytes; loss", out["step0/loss"], out["step1/loss"], "bf16 loss",
          out["bf16/loss"], "grad norm", out["step0/grad_norm"], "bf16 vs fp32 grad rel err",
          float(np.linalg.norm(gbf - g32) / np.linalg.norm(g32)))


def train_dataset_golden():
    """G11: LLMTrainDataset.__getitem__ (dataloader/llm.py:236-283) with the deterministic stand-in tokenizer and a
    seeded RandomState, and the reference's train collate on those samples -> tests/golden/llm_train_dataset.json."""
    from types import SimpleNamespace

    from tests.fake_tokenizer import FakeTokenizer
    from tests.gen_goldens_llm import TITLES

    argv, cwd = sys.argv, os.getcwd()
    sys.argv = ["x"]
    os.chdir(REF)
    sys.path.insert(0, REF)
    try:
        import dataloader.llm as DL  # noqa
        from dataloader.utils import Prompter
        import config as ref_config

        prompter, ref_args = Prompter(), ref_config.args
    finally:
        sys.argv = argv
        os.chdir(cwd)
    u2seq = {3: [1, 2, 3, 4], 1: [5, 6, 7], 2: [8, 9, 10, 1, 2, 3, 4, 5]}
    titles = dict(TITLES)
    for i in range(11, 41):
        titles[i] = f"Film number {i} ({1950 + i})"
    out = {"u2seq": {str(k): v for k, v in u2seq.items()}, "titles": {str(k): v for k, v in titles.items()},
           "cases": []}
    fns = functions_from(os.path.join(REF, "trainer", "llm.py"), ["llama_collate_fn_w_truncation"], {"torch": torch})
    for max_hist, neg, text_len, seed in ((20, 3, 1536, 7), (2, 4, 1536, 8), (3, 3, 60, 9)):
        args = SimpleNamespace(num_items=40, llm_negative_sample_size=neg, llm_max_title_len=32,
                               llm_max_text_len=text_len, llm_system_template=ref_args.llm_system_template,
                               llm_input_template=ref_args.llm_input_template, llm_train_on_inputs=False)
        ds = DL.LLMTrainDataset(args, u2seq, max_hist, np.random.RandomState(seed), titles, FakeTokenizer(), prompter)
        samples = [ds[i] for i in range(len(ds))]
        coll = fns["llama_collate_fn_w_truncation"](48, eval=False)(samples[:4])
        out["cases"].append({"llm_max_history": max_hist, "llm_negative_sample_size": neg, "llm_max_text_len": text_len,
                             "seed": seed, "all_seqs": ds.all_seqs,
                             "samples": [{k: [int(x) for x in s[k]] for k in ("input_ids", "attention_mask", "labels")}
                                         for s in samples],
                             "collate_max_length": 48, "collate_first4": {k: v.tolist() for k, v in coll.items()}})
    # G12: LLMValidDataset / LLMTestDataset (dataloader/llm.py:286-387) on a hand-made dataset + retrieved dict
    train = {1: [5, 6, 7, 8], 2: [9, 10, 11], 3: list(range(12, 40)), 4: [1, 2]}
    val = {1: [20], 2: [21], 3: [3], 4: [30]}
    test = {1: [25], 2: [26], 3: [4], 4: [31]}
    val_users, test_users = [1, 3, 4], [2, 3]
    rs = np.random.RandomState(11)
    def cands(answer):
        c = [int(x) for x in rs.permutation(np.arange(1, 41))[:6] if int(x) != answer][:5]
        c.insert(int(rs.randint(0, 6)), answer)
        return c
    val_c = [cands(val[u][0]) for u in val_users]
    test_c = [cands(test[u][0]) for u in test_users]
    ev = {"train": {str(k): v for k, v in train.items()}, "val": {str(k): v for k, v in val.items()},
          "test": {str(k): v for k, v in test.items()}, "val_users": val_users, "val_candidates": val_c,
          "test_users": test_users, "test_candidates": test_c, "cases": []}
    for max_hist, title_len in ((20, 32), (3, 2)):
        args = SimpleNamespace(num_items=40, llm_max_title_len=title_len, llm_max_text_len=1536,
                               llm_system_template=ref_args.llm_system_template,
                               llm_input_template=ref_args.llm_input_template)
        vd = DL.LLMValidDataset(args, train, val, max_hist, np.random, titles, FakeTokenizer(), prompter, val_users, val_c)
        td = DL.LLMTestDataset(args, train, val, test, max_hist, np.random, titles, FakeTokenizer(), prompter, test_users,
                               test_c)
        def items(ds):
            return [{k: ([int(x) for x in s[k]] if isinstance(s[k], (list, tuple)) else int(s[k]))
                     for k in ("input_ids", "attention_mask", "labels")} for s in (ds[i] for i in range(len(ds)))]
        ev["cases"].append({"llm_max_history": max_hist, "llm_max_title_len": title_len, "val": items(vd),
                            "test": items(td)})
    out["eval_datasets"] = ev
    path = os.path.join(OUT, "llm_train_dataset.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    train_dataset_golden()
    if "--data-only" not in sys.argv:
        for ci, name in enumerate(LLAMA_CONFIGS):
            run(name, ci)
