"""Online single-user path -- mirror of the reference's demo/inference.py:46-109 (second caller of
the same boundary): retrieve_candidates (no left padding, no history mask,
top_k 20..50), generate_prompt (titles NOT truncated), rank_candidates (verbalizer scores at the
last token -> top-k candidates). The Streamlit UI around it is out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from . import metrics as M
from . import prompt as P


def retrieve_candidates(model, query, top_k: int = 20):
    """demo/inference.py:46-53: `model(seqs)[:, -1, :]` then torch.topk -- here the fused retrieve
    without history exclusion (the demo does not mask the history)."""
    seqs = np.asarray(query, dtype=np.int64).reshape(1, -1)
    idx, _ = model.retrieve_topk(seqs, top_k, exclude_history=False)
    return idx[0].tolist()


def generate_prompt(query, candidates, dataset_map, instruction=P.DEFAULT_SYSTEM_TEMPLATE,
                    input_template=P.DEFAULT_INPUT_TEMPLATE, prompter=None):
    """demo/inference.py:79-109 (titles verbatim, alpaca_short layout)."""
    q_t = " \n ".join("(" + str(i + 1) + ") " + dataset_map[item] for i, item in enumerate(query))
    c_t = " \n ".join("(" + chr(ord("A") + i) + ") " + dataset_map[item] for i, item in enumerate(candidates))
    return (prompter or P.Prompter()).generate_prompt(instruction, input_template.format(q_t, c_t))


def rank_candidates(model, tokenizer, prompt, candidates, verbalizer, top_k: int = 10):
    """demo/inference.py:56-76: tokenise, one forward, verbalizer scores at the last position,
    top-k of the candidates (here: ordered by score desc, ties -> earlier candidate)."""
    ids = tokenizer(prompt, truncation=False, padding=False, return_tensors=None)["input_ids"]
    scores = model.prefill_verbalize([np.asarray(ids, dtype=np.int32)], verbalizer.label_token_ids[: len(candidates)])
    order = M.rank_classes(scores)[0].tolist()
    return [candidates[i] for i in order[:top_k]]
