"""Generate tests/golden/lru_train_v120.npz by RUNNING the reference's training step on CPU.

G8: LRURec teacher-forced cross-entropy over all positions
(trainer/lru.py:20-28, ignore_index=0), torch autograd gradients of every parameter, gradient
clipping (trainer/base.py:201-202) and two AdamW steps with the reference's parameter groups
(trainer/base.py:219-246: no weight decay on names containing "bias" or "layer_norm";
lr 1e-3, weight_decay 1e-2, eps 1e-9 -- config.py:121-124,181). Dropout is 0 in the captured
step (the reference's masks come from torch's RNG and cannot be reproduced elsewhere).
Batches follow LRUTrainDataset.__getitem__ (dataloader/lru.py:119-131): tokens = seq[:-1][-L:],
labels = seq[-L:], both left-padded with 0 -- so a short sequence has ONE pad position that
carries a real label.

Only data leaves this script (inputs, initial weights, the reference's outputs). Run from the
repo root:  PYTHONDONTWRITEBYTECODE=1 python tests/gen_goldens_train.py
"""
from __future__ import annotations

import os
import sys
import warnings
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")


def to_np(t):
    t = t.detach()
    if torch.is_complex(t):
        return torch.view_as_real(t).contiguous().numpy().astype(np.float32)  # [..., 2] (re, im)
    return t.numpy().astype(np.float32)


def datasets_golden():
    """G9: LRUTrainDataset / LRUValidDataset / LRUTestDataset of the reference (dataloader/lru.py:92-180) on a small
    synthetic user->sequence map -> tests/golden/lru_datasets.json (every sample, in order)."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("ref_dataloader_lru", os.path.join(REF, "dataloader", "lru.py"))
    DL = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(DL)
    rng = np.random.default_rng(5)
    train, val, test = {}, {}, {}
    for u, n in enumerate([3, 9, 10, 11, 17, 18, 19, 26, 40, 7, 2, 1], start=1):
        items = rng.integers(1, 60, size=n + 2).tolist()
        train[u], val[u], test[u] = items[:-2], items[-2:-1], items[-1:]
    val[10], test[11] = [], []          # users the eval datasets must drop
    out = {"train": {str(k): v for k, v in train.items()}, "val": {str(k): v for k, v in val.items()},
           "test": {str(k): v for k, v in test.items()}, "cases": []}
    args = SimpleNamespace(num_items=60)
    for max_len, sliding in ((8, 1.0), (8, 0.5), (5, 1.0)):
        ds = DL.LRUTrainDataset(args, train, max_len, sliding, rng)
        samples = [[t.tolist() for t in ds[i]] for i in range(len(ds))]
        vd = DL.LRUValidDataset(args, train, val, max_len, rng)
        td = DL.LRUTestDataset(args, train, val, test, max_len, rng)
        out["cases"].append({"max_len": max_len, "sliding_window_size": sliding, "train_samples": samples,
                             "val_users": list(vd.users), "val": [[t.tolist() for t in vd[i]] for i in range(len(vd))],
                             "test_users": list(td.users), "test": [[t.tolist() for t in td[i]] for i in range(len(td))]})
    path = os.path.join(OUT, "lru_datasets.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    datasets_golden()
    sys.path.insert(0, REF)
    from model.lru import LRURec  # reference import

    V, L, B = 120, 12, 7
    args = SimpleNamespace(num_items=V, bert_hidden_units=64, bert_num_blocks=2, bert_dropout=0.0,
                           bert_attn_dropout=0.0)
    torch.manual_seed(1234)
    model = LRURec(args).train()
    rng = np.random.default_rng(77)
    with torch.no_grad():  # non-trivial LayerNorm affine and item bias
        for n, p in model.named_parameters():
            if "layer_norm.weight" in n:
                p.copy_(torch.from_numpy(rng.uniform(0.7, 1.3, p.shape).astype(np.float32)))
            elif "layer_norm.bias" in n or n == "model.bias":
                p.copy_(torch.from_numpy(rng.uniform(-0.1, 0.1, p.shape).astype(np.float32)))
    # sequences of assorted lengths: shorter than L (left pad, one labelled pad position), exactly L + 1, longer
    lens = [3, 5, 13, 20, 9, 2, 12]
    toks, labs = [], []
    for n in lens:
        seq = rng.integers(1, V + 1, size=n).tolist()
        lab = seq[-L:]
        tok = seq[:-1][-L:]
        toks.append([0] * (L - len(tok)) + tok)
        labs.append([0] * (L - len(lab)) + lab)
    tokens = torch.tensor(toks, dtype=torch.long)
    labels = torch.tensor(labs, dtype=torch.long)
    assert tokens.shape == (B, L)

    out = {"tokens": tokens.numpy(), "labels": labels.numpy(), "num_items": np.int64(V)}
    names = [n for n, _ in model.named_parameters()]
    for n, p in model.named_parameters():
        out["init/" + n] = to_np(p)

    # --- trainer/base.py:219-246 parameter groups and AdamW
    no_decay = ["bias", "layer_norm"]
    groups = [
        {"params": [p for n, p in model.named_parameters() if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
        {"params": [p for n, p in model.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]
    opt = torch.optim.AdamW(groups, lr=1e-3, eps=1e-9)
    ce = torch.nn.CrossEntropyLoss(ignore_index=0)

    for step in range(2):
        opt.zero_grad()
        logits = model(tokens)                      # trainer/lru.py:22-27
        loss = ce(logits.view(-1, logits.size(-1)), labels.view(-1))
        loss.backward()
        out[f"step{step}/loss"] = np.float32(loss.item())
        for n, p in model.named_parameters():
            out[f"step{step}/grad/" + n] = to_np(p.grad)
        # a small limit on step 1 so that clipping really rescales (trainer/base.py:109,201-202: limit 5.0)
        limit = 5.0 if step == 0 else 0.05
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), limit)
        out[f"step{step}/grad_norm"] = np.float32(float(norm))
        out[f"step{step}/clip_limit"] = np.float32(limit)
        opt.step()
        for n, p in model.named_parameters():
            out[f"step{step}/param/" + n] = to_np(p)
    # last-position scores after the two steps (ties the trained weights back to the scoring path)
    model.eval()
    with torch.no_grad():
        out["final_scores_last"] = model(tokens)[:, -1, :].numpy().astype(np.float32)
    # a longer trajectory (losses only): 40 more steps at the reference's own limit 5.0
    model.train()
    traj = []
    for step in range(40):
        opt.zero_grad()
        logits = model(tokens)
        loss = ce(logits.view(-1, logits.size(-1)), labels.view(-1))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        traj.append(loss.item())
    out["traj_loss"] = np.asarray(traj, np.float32)
    out["param_names"] = np.array(names)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "lru_train_v120.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; loss", out["step0/loss"], out["step1/loss"],
          "grad norms", out["step0/grad_norm"], out["step1/grad_norm"])


if __name__ == "__main__":
    main()
