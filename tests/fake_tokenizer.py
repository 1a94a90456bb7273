"""Deterministic stand-in for the Llama-2 sentencepiece tokenizer (absent offline).

Exposes exactly the calls the reference makes on its tokenizer (dataloader/llm.py:21-27,67-69,
trainer/verb.py:494): tokenize, convert_tokens_to_string, __call__(truncation, max_length),
encode(add_special_tokens=False), plus the attributes set at dataloader/llm.py:122-126.
"""
import zlib


class FakeTokenizer:
    bos_token_id = 1
    eos_token_id = 2
    unk_token_id = 0
    pad_token = "<unk>"
    unk_token = "<unk>"
    padding_side = "left"
    truncation_side = "left"
    vocab_size = 1000

    def __init__(self):
        self.seen_texts = []

    def tokenize(self, text):
        return [t for t in text.split(" ") if t != ""]

    def convert_tokens_to_string(self, tokens):
        return " ".join(tokens)

    def _id(self, tok):
        return 3 + zlib.crc32(tok.encode("utf-8")) % (self.vocab_size - 3)

    def encode(self, text, add_special_tokens=True):
        ids = [self._id(t) for t in self.tokenize(text)]
        return ([self.bos_token_id] + ids) if add_special_tokens else ids

    def __call__(self, text, truncation=False, max_length=None, padding=False, return_tensors=None):
        self.seen_texts.append(text)
        ids = self.encode(text, add_special_tokens=True)
        if truncation and max_length is not None and len(ids) > max_length:
            ids = ids[-max_length:] if self.truncation_side == "left" else ids[:max_length]
        return {"input_ids": ids, "attention_mask": [1] * len(ids)}
