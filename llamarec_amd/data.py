"""On-disk formats either side of the scoring path.

  dataset.pkl  : dict(train, val, test: {user: [item,...]}, meta: {item: title}, umap, smap)
                 written by the reference's llamarec_datasets (base.py:58-62,129-140) under
                 data/preprocessed/<code>_min_rating0-min_uc5-min_sc5/dataset.pkl
  retrieved.pkl: the 10-key dict of trainer/lru.py:160-175 (written by retrieve.py here)
  best_acc_model.pth: torch.save({"model_state_dict": ...}) (trainer/base.py:326-330)
"""
from __future__ import annotations

import os
import pickle

import numpy as np


def preprocessed_path(root, dataset_code, min_rating=0, min_uc=5, min_sc=5):
    """llamarec_datasets/base.py:129-140 folder naming."""
    return os.path.join(root, "preprocessed", f"{dataset_code}_min_rating{min_rating}-min_uc{min_uc}-min_sc{min_sc}",
                        "dataset.pkl")


def load_dataset_pkl(path):
    with open(path, "rb") as f:
        d = pickle.load(f)
    for k in ("train", "val", "test", "umap", "smap"):
        if k not in d:
            raise ValueError(f"{path}: missing key {k!r}")
    return d


def left_pad(seq, max_len):
    seq = list(seq)[-max_len:]
    return [0] * (max_len - len(seq)) + seq


def lru_eval_arrays(dataset, mode, max_len):
    """Inputs of the retriever's eval datasets (dataloader/lru.py:129-180): for `val` the history is
    train, the answer val; for `test` the history is train+val, the answer test; users are taken in
    sorted order and kept only if their answer lists are non-empty; histories keep the last max_len
    items and are left-padded with 0. Returns (users, ids int64 [U,L], labels int64 [U,1])."""
    train, val, test = dataset["train"], dataset["val"], dataset["test"]
    users = sorted(train.keys())
    if mode == "val":
        users = [u for u in users if len(val[u]) > 0]
        rows = [left_pad(train[u], max_len) for u in users]
        labels = [val[u][:1] for u in users]
    elif mode == "test":
        users = [u for u in users if len(val[u]) > 0 and len(test[u]) > 0]
        rows = [left_pad(list(train[u]) + list(val[u]), max_len) for u in users]
        labels = [test[u][:1] for u in users]
    else:
        raise ValueError(mode)
    return users, np.asarray(rows, dtype=np.int64).reshape(len(users), max_len), \
        np.asarray(labels, dtype=np.int64).reshape(len(users), 1)


def lru_train_sequences(dataset, max_len, sliding_window_size=1.0):
    """LRUTrainDataset.__init__ (dataloader/lru.py:92-112): users in sorted order; a sequence shorter than
    max_len + step is one sample, a longer one is cut into windows of max_len taken from the END backwards
    with step int(sliding_window_size * max_len)."""
    step = int(sliding_window_size * max_len)
    assert step > 0
    out = []
    u2seq = dataset["train"]
    for u in sorted(u2seq.keys()):
        seq = list(u2seq[u])
        if len(seq) < max_len + step:
            out.append(seq)
        else:
            out.extend(seq[i:i + max_len] for i in range(len(seq) - max_len, -1, -step))
    return out


def lru_train_batch(seqs, max_len):
    """LRUTrainDataset.__getitem__ (dataloader/lru.py:117-131) for a list of samples: tokens = seq[:-1][-L:],
    labels = seq[-L:], both left-padded with 0 -> int64 [B, L] x 2. A sample of <= L items therefore has one pad
    position that carries a label (its first item)."""
    B = len(seqs)
    tokens = np.zeros((B, max_len), np.int64)
    labels = np.zeros((B, max_len), np.int64)
    for i, seq in enumerate(seqs):
        lab = seq[-max_len:]
        tok = seq[:-1][-max_len:]
        if tok:
            tokens[i, max_len - len(tok):] = tok
        labels[i, max_len - len(lab):] = lab
    return tokens, labels


def train_batches(all_seqs, batch_size, max_len, rng, rank=0, world=1):
    """One epoch of shuffled training batches (DataLoader(shuffle=True), dataloader/lru.py:52-60; the order comes
    from `rng`, torch's sampler stream is not reproduced). With world > 1 every rank sees the same permutation and
    takes its contiguous share of each global batch."""
    order = rng.permutation(len(all_seqs))
    for i in range(0, len(order), batch_size * world):
        idx = order[i + rank * batch_size:i + (rank + 1) * batch_size]
        if len(idx) == 0:
            idx = order[i:i + 1]          # keep ranks in lock-step on a ragged tail
        yield lru_train_batch([all_seqs[j] for j in idx], max_len)


def batches(ids, labels, batch_size):
    """Unshuffled eval batches (dataloader/lru.py:74: shuffle=False keeps user ids positional)."""
    for i in range(0, len(ids), batch_size):
        yield ids[i:i + batch_size], labels[i:i + batch_size]


def synthetic_dataset(num_users, num_items, mean_len=12, seed=0, title_words=4):
    """A small fabricated dataset.pkl-shaped dict (no real data exists offline)."""
    rng = np.random.default_rng(seed)
    train, val, test = {}, {}, {}
    for u in range(1, num_users + 1):
        n = int(np.clip(rng.geometric(1.0 / mean_len) + 4, 5, num_items))
        items = (rng.choice(num_items, size=n, replace=False) + 1).tolist()
        train[u], val[u], test[u] = items[:-2], items[-2:-1], items[-1:]
    words = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet"]
    meta = {i: " ".join(rng.choice(words, size=title_words)) + f" ({1990 + i % 30})" for i in range(1, num_items + 1)}
    return {"train": train, "val": val, "test": test, "meta": meta,
            "umap": {u: u for u in range(1, num_users + 1)}, "smap": {i: i for i in range(1, num_items + 1)}}
