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
