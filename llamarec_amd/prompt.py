"""Prompt construction and eval collation for the ranker -- host-side mirror of the reference's
dataloader/llm.py:19-30,64-98, dataloader/utils.py:24-40 and trainer/llm.py:15-60 (eval branch).

Pure host logic (strings and integer lists); the tokenizer is any object with the calls the
reference makes: tokenize, convert_tokens_to_string, __call__(text, truncation, max_length,
padding, return_tensors) and encode(text, add_special_tokens=False).
"""
from __future__ import annotations

import json
import os

import numpy as np

# config.py:242-249 of the reference (defaults of --llm_system_template / --llm_input_template)
DEFAULT_SYSTEM_TEMPLATE = ("Given user history in chronological order, recommend an item from the candidate pool "
                           "with its index letter.")
DEFAULT_INPUT_TEMPLATE = "User history: {}; \n Candidate pool: {}"
LLM_MAX_TITLE_LEN = 32    # config.py:235
LLM_MAX_TEXT_LEN = 1536   # config.py:236
LLM_MAX_HISTORY = 20      # config.py:237

# Prompt layouts. "alpaca_short" must produce byte-identical prompts to the reference's default
# template (dataloader/templates/alpaca_short.json:3-4, pinned by tests/golden/prompts.json).
_SECTION = "### {}:\n"
TEMPLATES = {
    "alpaca_short": {
        "prompt_input": _SECTION.format("Instruction") + "{instruction}\n\n" + _SECTION.format("Input")
        + "{input}\n\n" + _SECTION.format("Response"),
        "prompt_no_input": _SECTION.format("Instruction") + "{instruction}\n\n" + _SECTION.format("Response"),
        "response_split": "### Response:",
    },
}


class Prompter:
    """dataloader/utils.py:8-40: template lookup + `generate_prompt(instruction, input, label)`."""

    __slots__ = ("template", "_verbose")

    def __init__(self, template_name: str = "", verbose: bool = False):
        self._verbose = verbose
        if not template_name:
            template_name = "alpaca_short"
        if template_name in TEMPLATES:
            self.template = TEMPLATES[template_name]
        elif os.path.exists(template_name):  # a user-supplied JSON file with the same three keys
            with open(template_name) as fp:
                self.template = json.load(fp)
        else:
            raise ValueError(f"Can't read {template_name}")

    def generate_prompt(self, instruction, input=None, label=None) -> str:
        if input:
            res = self.template["prompt_input"].format(instruction=instruction, input=input)
        else:
            res = self.template["prompt_no_input"].format(instruction=instruction)
        if label:
            res = f"{res}{label}"
        return res


def truncate_title(title, tokenizer, max_title_len=LLM_MAX_TITLE_LEN):
    """dataloader/llm.py:67-70."""
    return tokenizer.convert_tokens_to_string(tokenizer.tokenize(title)[:max_title_len])


class TitleCache:
    """Per-ITEM memo of truncate_title. The reference truncates every title of every prompt by a tokenizer round trip
    (dataloader/llm.py:67-70: 41 tokenizer calls per evaluation user); the result is a pure function of (title,
    tokenizer, max_title_len), so one call per item of the catalog (12 k on Beauty against 22 k users x 41) builds the
    same prompt strings. `ntok[item]` (the truncated title's token count) also gives a prompt-length estimate without
    tokenising the prompt (estimate_prompt_tokens: what the data-parallel shards are balanced by BEFORE tokenisation)."""

    def __init__(self, text_dict, tokenizer, max_title_len=LLM_MAX_TITLE_LEN):
        self.text_dict, self.tokenizer, self.max_title_len = text_dict, tokenizer, max_title_len
        self._text, self.ntok = {}, {}

    def __call__(self, item):
        t = self._text.get(item)
        if t is None:
            toks = self.tokenizer.tokenize(self.text_dict[item])[: self.max_title_len]
            t = self._text[item] = self.tokenizer.convert_tokens_to_string(toks)
            self.ntok[item] = len(toks)
        return t

    def estimate_prompt_tokens(self, seq, candidates, overhead=64, per_line=5):
        """~ tokens of the prompt built from (seq, candidates): template text + per-line markers + title tokens."""
        n = overhead + per_line * (len(seq) + len(candidates))
        for item in list(seq) + list(candidates):
            if item not in self.ntok:
                self(item)
            n += self.ntok[item]
        return n


def build_input_text(seq, candidates, text_dict, tokenizer, max_title_len=LLM_MAX_TITLE_LEN,
                     input_template=DEFAULT_INPUT_TEMPLATE, truncate=True, title_cache=None):
    """History as "(1) title \\n (2) title", candidates as "(A) title \\n (B) title"
    (dataloader/llm.py:72-83,92). truncate=False is the online demo's variant (demo/inference.py:79-109).
    title_cache: a TitleCache over (text_dict, tokenizer, max_title_len) -- the same strings, one tokenizer call per item."""
    if not truncate:
        tt = lambda item: text_dict[item]
    elif title_cache is not None:
        tt = title_cache
    else:
        tt = lambda item: truncate_title(text_dict[item], tokenizer, max_title_len)
    seq_t = " \n ".join("(" + str(i + 1) + ") " + tt(item) for i, item in enumerate(seq))
    can_t = " \n ".join("(" + chr(ord("A") + i) + ") " + tt(item) for i, item in enumerate(candidates))
    return input_template.format(seq_t, can_t)


def eval_prompt_text(seq, candidates, label, text_dict, tokenizer, prompter=None, max_title_len=LLM_MAX_TITLE_LEN,
                     system_template=DEFAULT_SYSTEM_TEMPLATE, input_template=DEFAULT_INPUT_TEMPLATE, title_cache=None):
    """The string seq_to_token_ids tokenises and the answer's class index (dataloader/llm.py:64-98, eval branch)."""
    prompter = prompter or Prompter()
    candidates = list(candidates)
    text = build_input_text(seq, candidates, text_dict, tokenizer, max_title_len, input_template, title_cache=title_cache)
    return prompter.generate_prompt(system_template, text), candidates.index(label)


def tokenize_prompts(prompts, tokenizer, max_text_len=LLM_MAX_TEXT_LEN):
    """input_ids of generate_and_tokenize_eval (dataloader/llm.py:19-30: truncation=True, max_length, no padding) for a
    LIST of prompt strings. A HF fast tokenizer encodes the list in one call (its Rust encoder runs the prompts in
    parallel and releases the GIL, so a producer thread overlaps it with the GPU loop); any other tokenizer is called
    per prompt. Same ids either way (tests/test_host_logic.py)."""
    if getattr(tokenizer, "is_fast", False) and len(prompts) > 1:
        enc = tokenizer(list(prompts), truncation=True, max_length=max_text_len, padding=False, return_tensors=None)
        return [list(x) for x in enc["input_ids"]]
    return [list(tokenizer(p, truncation=True, max_length=max_text_len, padding=False, return_tensors=None)["input_ids"])
            for p in prompts]


def seq_to_token_ids(seq, candidates, label, text_dict, tokenizer, prompter=None,
                     max_title_len=LLM_MAX_TITLE_LEN, max_text_len=LLM_MAX_TEXT_LEN,
                     system_template=DEFAULT_SYSTEM_TEMPLATE, input_template=DEFAULT_INPUT_TEMPLATE):
    """Eval branch of dataloader/llm.py:64-98 + generate_and_tokenize_eval (:19-30): returns
    {"input_ids", "attention_mask", "labels"} with labels = index of the answer's letter."""
    prompter = prompter or Prompter()
    candidates = list(candidates)
    output = chr(ord("A") + candidates.index(label))
    text = build_input_text(seq, candidates, text_dict, tokenizer, max_title_len, input_template)
    prompt = prompter.generate_prompt(system_template, text)
    tok = tokenizer(prompt, truncation=True, max_length=max_text_len, padding=False, return_tensors=None)
    out = {"input_ids": list(tok["input_ids"]), "attention_mask": list(tok["attention_mask"])}
    out["labels"] = ord(output) - ord("A")
    return out


def seq_to_token_ids_train(seq, candidates, label, text_dict, tokenizer, prompter=None,
                           max_title_len=LLM_MAX_TITLE_LEN, max_text_len=LLM_MAX_TEXT_LEN,
                           system_template=DEFAULT_SYSTEM_TEMPLATE, input_template=DEFAULT_INPUT_TEMPLATE,
                           train_on_inputs=False):
    """Train branch of dataloader/llm.py:64-98 + generate_and_tokenize_train (:33-61): the prompt with the answer
    letter appended, EOS / BOS added when the tokenizer did not, labels = input_ids with everything except the last
    two tokens (letter, EOS) set to -100."""
    prompter = prompter or Prompter()
    candidates = list(candidates)
    output = chr(ord("A") + candidates.index(label))
    text = build_input_text(seq, candidates, text_dict, tokenizer, max_title_len, input_template)
    full = prompter.generate_prompt(system_template, text, output)
    tok = tokenizer(full, truncation=True, max_length=max_text_len, padding=False, return_tensors=None)
    ids, mask = list(tok["input_ids"]), list(tok["attention_mask"])
    if ids[-1] != tokenizer.eos_token_id:
        ids.append(tokenizer.eos_token_id)
        mask.append(1)
    if ids[0] != tokenizer.bos_token_id:
        ids.insert(0, tokenizer.bos_token_id)
        mask.insert(0, 1)
    labels = list(ids)
    if not train_on_inputs:
        labels[:-2] = [-100] * len(labels[:-2])
    return {"input_ids": ids, "attention_mask": mask, "labels": labels}


def train_pack(batch, llm_max_length=LLM_MAX_TEXT_LEN, eos_token_id=2):
    """The packed (unpadded) equivalent of llama_collate_fn_w_truncation(eval=False) (trainer/llm.py:15-60): per-prompt
    ids and labels after the same left truncation and the same sanity checks; padding never reaches the GPU."""
    longest = max(len(b["input_ids"]) for b in batch)
    max_length = min(llm_max_length, longest)
    seqs, labels = [], []
    for b in batch:
        ids, lab = list(b["input_ids"])[-max_length:], list(b["labels"])[-max_length:]
        assert ids[-1] == eos_token_id                          # trainer/llm.py:46-48
        assert lab[-3] == -100 and lab[-2] != -100
        seqs.append(np.asarray(ids, dtype=np.int32))
        labels.append(np.asarray(lab, dtype=np.int64))
    return seqs, labels


def eval_collate(batch, llm_max_length=LLM_MAX_TEXT_LEN):
    """llama_collate_fn_w_truncation(eval=True) (trainer/llm.py:15-60): left-truncate to
    min(llm_max_length, longest), left-pad ids with 0 and the mask with 0; labels [B,1].
    Returns int64 numpy arrays."""
    longest = max(len(b["input_ids"]) for b in batch)
    max_length = min(llm_max_length, longest)
    ids = np.zeros((len(batch), max_length), np.int64)
    mask = np.zeros((len(batch), max_length), np.int64)
    labels = np.zeros((len(batch), 1), np.int64)
    for i, b in enumerate(batch):
        x, m = list(b["input_ids"]), list(b["attention_mask"])
        if len(x) > max_length:
            x, m = x[-max_length:], m[-max_length:]
        ids[i, max_length - len(x):] = x
        mask[i, max_length - len(m):] = m
        labels[i, 0] = b["labels"]
    return {"input_ids": ids, "attention_mask": mask, "labels": labels}


def eval_pack(batch, llm_max_length=LLM_MAX_TEXT_LEN):
    """The packed (unpadded) equivalent of eval_collate for the HIP prefill: per-prompt id lists
    after the same left truncation, plus labels. Padding never reaches the GPU."""
    seqs = [np.asarray(b["input_ids"][-llm_max_length:], dtype=np.int32) for b in batch]
    labels = np.asarray([b["labels"] for b in batch], dtype=np.int64)
    return seqs, labels
