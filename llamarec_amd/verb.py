"""Verbalizer -- host-side mirror of the reference's ManualVerbalizer as configured at
trainer/llm.py:93-101 (prefix "", one single-token word per class, post_log_softmax=False,
multi_token_handler "first"): trainer/verb.py:486-522 builds `label_words_ids` [C,1,len],
:524-544 projects `logits[:, ids]`, :602-614 averages over the single word. Net effect (verified
against the reference, tests/golden/verbalizer.npz): scores[b, c] = logits[b, id_c].

On the GPU path the gather is folded into the prefill's head kernel (only the C label rows of
lm_head are multiplied); `process_logits` exists for callers that already hold full logits.
"""
from __future__ import annotations

import numpy as np


class ManualVerbalizer:
    def __init__(self, tokenizer, classes=None, num_classes=None, label_words=None, prefix="",
                 multi_token_handler="first", post_log_softmax=False):
        if post_log_softmax:
            raise NotImplementedError("the reference runs with post_log_softmax=False (trainer/llm.py:96)")
        if multi_token_handler != "first":
            raise NotImplementedError("only the reference's default multi_token_handler='first'")
        if classes is None:
            classes = list(range(num_classes))
        self.classes = list(classes)
        self.num_classes = len(self.classes)
        self.prefix = prefix
        if isinstance(label_words, dict):
            label_words = [label_words[c] for c in self.classes]
        words = []
        for w in label_words:
            w = [w] if isinstance(w, str) else list(w)
            if len(w) != 1:
                raise NotImplementedError("one label word per class (trainer/llm.py:98-100)")
            words.append(prefix + w[0].lstrip(prefix) if prefix else w[0])
        self.label_words = words
        ids = []
        for w in words:
            enc = tokenizer.encode(w, add_special_tokens=False)  # trainer/verb.py:494
            if len(enc) < 1:
                raise ValueError(f"label word {w!r} encodes to no token")
            ids.append(int(enc[0]))  # "first" token handler (trainer/verb.py:292-293)
        self.label_token_ids = np.asarray(ids, dtype=np.int32)
        # shape the reference exposes: [C, 1 word, max_len tokens]
        self.label_words_ids = self.label_token_ids.reshape(self.num_classes, 1, 1).astype(np.int64)

    def project(self, logits):
        return logits[:, self.label_token_ids.astype(np.int64)]

    def process_logits(self, logits):
        """[B, vocab] -> [B, C] (works on numpy arrays and torch tensors on any device)."""
        idx = self.label_token_ids.astype(np.int64)
        try:
            import torch

            if isinstance(logits, torch.Tensor):
                return logits[:, torch.as_tensor(idx, device=logits.device)]
        except ImportError:  # pragma: no cover
            pass
        return logits[:, idx]
