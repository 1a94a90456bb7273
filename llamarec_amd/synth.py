"""Deterministic synthetic weights and workloads (no datasets or checkpoints exist offline).

Everything here is integer-hash based or uses numpy.random.default_rng with the seeds fixed in
BASELINE.md section 3, so the same inputs are rebuilt bit-for-bit on any machine.
"""
from __future__ import annotations

import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit hashes of the counters offset..offset+n-1 under `seed` (splitmix64 finaliser)."""
    with np.errstate(over="ignore"):
        z = (np.arange(offset + 1, offset + n + 1, dtype=np.uint64) * _M1) + np.uint64(seed) * _M3
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(seed: int, shape, std: float = 0.02) -> np.ndarray:
    """float32 array, uniform with standard deviation `std` (exactly reproducible: only
    integer hashing and power-of-two scaling followed by one float32 multiply)."""
    n = int(np.prod(shape))
    u = (splitmix64(seed, n) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)
    return (u * np.float32(std * np.sqrt(3.0))).reshape(shape)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even float32 -> bfloat16 bit patterns (uint16). NaNs stay NaN."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    out = ((u + r) >> np.uint32(16)).astype(np.uint16)
    nan = np.isnan(x)
    if nan.any():
        out = np.where(nan, np.uint16(0x7FC0), out)
    return out


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 values rounded to the nearest bfloat16 (returned as float32)."""
    return bf16_bits_to_f32(f32_to_bf16_bits(x))


def llama_param_shapes(cfg: dict) -> list[tuple[str, tuple]]:
    """HF LlamaForCausalLM parameter names and shapes for a config dict (no biases)."""
    d, f, v = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = d // nh
    out = [("model.embed_tokens.weight", (v, d))]
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        out += [
            (p + "self_attn.q_proj.weight", (nh * hd, d)),
            (p + "self_attn.k_proj.weight", (nkv * hd, d)),
            (p + "self_attn.v_proj.weight", (nkv * hd, d)),
            (p + "self_attn.o_proj.weight", (d, nh * hd)),
            (p + "mlp.gate_proj.weight", (f, d)),
            (p + "mlp.up_proj.weight", (f, d)),
            (p + "mlp.down_proj.weight", (d, f)),
            (p + "input_layernorm.weight", (d,)),
            (p + "post_attention_layernorm.weight", (d,)),
        ]
    out += [("model.norm.weight", (d,)), ("lm_head.weight", (v, d))]
    return out


def synth_llama_state(cfg: dict, seed: int, std: float = 0.02, norm_jitter: float = 0.1) -> dict:
    """name -> float32 array. Matrices ~ U(std); norm weights = 1 + U(norm_jitter).
    Values are bf16-representable so that fp32 and bf16 runs use identical weights."""
    sd = {}
    for i, (name, shape) in enumerate(llama_param_shapes(cfg)):
        if len(shape) == 1:
            w = np.float32(1.0) + hash_uniform(seed * 1000 + i, shape, norm_jitter)
        else:
            w = hash_uniform(seed * 1000 + i, shape, std)
        sd[name] = bf16_round(w)
    return sd


# ---------------------------------------------------------------------------------------------
# Synthetic workloads of BASELINE.json's configs (BASELINE.md section 3 / SURVEY.md 8(d)).
# V items, U users, L = bert_max_len, title-length range of the prompt model, rerank batch
# (config.py:98), cfg_idx = position in BASELINE.json "configs" (seeds 42 + cfg_idx).
# ---------------------------------------------------------------------------------------------
WORKLOADS = {
    "ml-100k": dict(V=3650, U=610, L=200, title=(3, 10), rerank_batch=32, cfg_idx=1),
    "beauty": dict(V=12086, U=22332, L=50, title=(10, 32), rerank_batch=16, cfg_idx=2),
    "games": dict(V=7676, U=15264, L=50, title=(6, 24), rerank_batch=16, cfg_idx=3),
    "synth-1m": dict(V=1_000_000, U=1_000_000, L=200, title=(10, 32), rerank_batch=16, cfg_idx=4),
}
LLM_MAX_TEXT_LEN = 1536   # config.py:236
LLM_MAX_HISTORY = 20      # config.py:237
NUM_CANDIDATES = 20       # llm_negative_sample_size + 1 (config.py:238-241)


def history_lengths(name: str, n_users: int, rng) -> np.ndarray:
    w = WORKLOADS[name]
    L = w["L"]
    if name == "ml-100k":
        n = np.clip(np.rint(rng.lognormal(4.4, 1.0, size=n_users)), 3, 200)
    elif name == "beauty":
        n = np.minimum(L, 3 + rng.geometric(0.2, size=n_users))
    elif name == "games":
        n = np.minimum(L, 3 + rng.geometric(0.17, size=n_users))
    else:
        n = rng.integers(20, 201, size=n_users)
    return np.minimum(n, L).astype(np.int64)


def synth_users(name: str, n_users: int | None = None, first_user: int = 0, seed_offset: int = 0):
    """Left-padded int64 histories [U, L], labels [U] (not in the history), history lengths, and
    prompt lengths T_u. Users are generated independently from (seed, user index) blocks so that a
    data-parallel rank can build exactly its own contiguous shard."""
    w = WORKLOADS[name]
    V, L = w["V"], w["L"]
    U = w["U"] if n_users is None else n_users
    rng = np.random.default_rng([42 + w["cfg_idx"] + seed_offset, first_user])
    n = history_lengths(name, U, rng)
    ids = np.zeros((U, L), np.int64)
    labels = np.zeros(U, np.int64)
    for u in range(U):
        # n_u + 1 distinct items: history + label
        pick = rng.choice(V, size=int(n[u]) + 1, replace=False) + 1 if V < 50_000 else \
            np.unique(rng.integers(1, V + 1, size=int(n[u]) + 8))[: int(n[u]) + 1]
        if len(pick) < n[u] + 1:  # pragma: no cover (astronomically unlikely for V >= 50k)
            pick = rng.choice(V, size=int(n[u]) + 1, replace=False) + 1
        pick = rng.permutation(pick)
        ids[u, L - n[u]:] = pick[: n[u]]
        labels[u] = pick[n[u]]
    lo, hi = w["title"]
    hist = np.minimum(n, LLM_MAX_HISTORY)
    T = np.empty(U, np.int64)
    for u in range(U):
        t = rng.integers(lo, hi + 1, size=int(hist[u]) + NUM_CANDIDATES)
        T[u] = min(LLM_MAX_TEXT_LEN, 48 + int((4 + t).sum()))
    return ids, labels, n, T


# Every prompt opens with the same template text (dataloader/utils.py:24-40, templates/alpaca_short.json:3,
# config.py:247-249): BOS + "### Instruction:\n<system sentence>\n\n### Input:\nUser history:" -- with the Llama-2
# sentencepiece vocabulary 1 + 5 + 20 + 2 + 4 + 3 = 35 tokens, then "(1)" of the first history item; the other
# template tokens of SURVEY.md 8(d)'s 48 ("; \n Candidate pool:", "\n\n### Response:\n") sit at user-dependent
# positions. Prompts clamped by the 1536-token LEFT truncation (config.py:236) lose that prefix.
TEMPLATE_PREFIX_TOKENS = 36


def synth_prompt_tokens(T: np.ndarray, seed: int, vocab: int = 32000, shared_prefix: bool = True):
    """Packed int32 prompt ids for prompt lengths T: BOS (=1) then uniform ids in [3, vocab). With shared_prefix the
    first TEMPLATE_PREFIX_TOKENS ids of every prompt shorter than LLM_MAX_TEXT_LEN are the same fixed template ids
    (the ids of a truncated prompt stay random: its head was cut off)."""
    rng = np.random.default_rng(seed)
    T = np.asarray(T, dtype=np.int64)
    cu = np.zeros(len(T) + 1, np.int32)
    cu[1:] = np.cumsum(T)
    ids = rng.integers(3, vocab, size=int(cu[-1]), dtype=np.int32)
    ids[cu[:-1]] = 1
    if shared_prefix:
        template = np.random.default_rng(20240807).integers(3, vocab, size=TEMPLATE_PREFIX_TOKENS, dtype=np.int32)
        template[0] = 1
        for b in range(len(T)):
            if T[b] < LLM_MAX_TEXT_LEN:
                n = min(TEMPLATE_PREFIX_TOKENS, int(T[b]))
                ids[cu[b]: cu[b] + n] = template[:n]
    return ids, cu


# ---- a deterministic stand-in tokenizer for --synthetic runs and tests (no Llama tokenizer files exist offline) ----
# Exposes exactly the calls the reference makes on its tokenizer (dataloader/llm.py:21-27,67-69,
# trainer/verb.py:494): tokenize, convert_tokens_to_string, __call__(truncation, max_length),
# encode(add_special_tokens=False), plus the attributes set at dataloader/llm.py:122-126.
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
