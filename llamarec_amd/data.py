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
