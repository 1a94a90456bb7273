"""ctypes mirror of include/llamarec_mi355x.h (struct layouts + prototypes).

Kept free of any compute: it only describes the C ABI so that the product loader
(`llamarec_amd._lib`) and the test-side oracle wrapper (`oracle/lru_oracle.py`) agree on
struct layout.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

LR_MAX_LRU_BLOCKS = 4
LR_MAX_TOPK = 64

c_float_p = C.POINTER(C.c_float)
c_u16_p = C.POINTER(C.c_uint16)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)


class LrLruBlockWeights(C.Structure):
    _fields_ = [
        (n, c_float_p)
        for n in (
            "params_log", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "ln1_w", "ln1_b",
            "ffn_w1", "ffn_b1", "ffn_w2", "ffn_b2", "ln2_w", "ln2_b",
        )
    ]


class LrLruWeightsDesc(C.Structure):
    _fields_ = [
        ("num_items", C.c_int32),
        ("hidden", C.c_int32),
        ("num_blocks", C.c_int32),
        ("reserved", C.c_int32),
        ("item_emb", c_float_p),
        ("item_bias", c_float_p),
        ("emb_ln_w", c_float_p),
        ("emb_ln_b", c_float_p),
        ("blocks", LrLruBlockWeights * LR_MAX_LRU_BLOCKS),
    ]


class LrLlamaConfig(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32),
        ("hidden_size", C.c_int32),
        ("intermediate_size", C.c_int32),
        ("num_layers", C.c_int32),
        ("num_heads", C.c_int32),
        ("num_kv_heads", C.c_int32),
        ("head_dim", C.c_int32),
        ("max_positions", C.c_int32),
        ("rms_eps", C.c_float),
        ("rope_theta", C.c_float),
    ]


class LrLlamaLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("input_norm", "wqkv", "wo", "post_norm", "wgu", "wdown")]


class LrLlamaWeightsDesc(C.Structure):
    _fields_ = [
        ("embed", C.c_void_p),
        ("final_norm", C.c_void_p),
        ("lm_head", C.c_void_p),
        ("layers", C.POINTER(LrLlamaLayerWeights)),
    ]


# Reference state_dict names (SURVEY.md 8(a) a-W; model/lru.py) -> desc fields.
_BLOCK_KEYS = {
    "params_log": "lru_layer.params_log",
    "in_proj_w": "lru_layer.in_proj.weight",
    "in_proj_b": "lru_layer.in_proj.bias",
    "out_proj_w": "lru_layer.out_proj.weight",
    "out_proj_b": "lru_layer.out_proj.bias",
    "ln1_w": "lru_layer.layer_norm.weight",
    "ln1_b": "lru_layer.layer_norm.bias",
    "ffn_w1": "feed_forward.w_1.weight",
    "ffn_b1": "feed_forward.w_1.bias",
    "ffn_w2": "feed_forward.w_2.weight",
    "ffn_b2": "feed_forward.w_2.bias",
    "ln2_w": "feed_forward.layer_norm.weight",
    "ln2_b": "feed_forward.layer_norm.bias",
}
_BLOCK_SHAPES = {
    "params_log": (3, 128), "in_proj_w": (128, 64, 2), "in_proj_b": (128, 2),
    "out_proj_w": (64, 128, 2), "out_proj_b": (64, 2), "ln1_w": (64,), "ln1_b": (64,),
    "ffn_w1": (256, 64), "ffn_b1": (256,), "ffn_w2": (64, 256), "ffn_b2": (64,),
    "ln2_w": (64,), "ln2_b": (64,),
}


def _as_f32(a) -> np.ndarray:
    """numpy / torch tensor (real or complex64) -> contiguous float32 array ((re,im) last)."""
    if hasattr(a, "detach"):
        a = a.detach().cpu()
        if a.is_complex():
            import torch

            a = torch.view_as_real(a)
        a = a.numpy()
    a = np.asarray(a)
    if np.iscomplexobj(a):
        a = np.ascontiguousarray(a.astype(np.complex64)).view(np.float32).reshape(a.shape + (2,))
    return np.ascontiguousarray(a, dtype=np.float32)


class LrLruTrainConfig(C.Structure):
    """include/llamarec_mi355x.h: LrLruTrainConfig."""
    _fields_ = [("weight_decay", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("max_grad_norm", C.c_float), ("dropout", C.c_float), ("attn_dropout", C.c_float),
                ("seed", C.c_uint64), ("ce_mode", C.c_int32)]


class LrLlamaLayerWeightsT(C.Structure):
    """include/llamarec_mi355x.h: transposed copies of the frozen matrices (data-gradient GEMMs)."""
    _fields_ = [("wqkv_t", C.c_void_p), ("wo_t", C.c_void_p), ("wgu_t", C.c_void_p), ("wdown_t", C.c_void_p)]


class LrLlamaWeightsTDesc(C.Structure):
    _fields_ = [("layers", C.POINTER(LrLlamaLayerWeightsT)), ("lm_head_t", C.c_void_p)]


class LrLoraTrainConfig(C.Structure):
    """include/llamarec_mi355x.h: LrLoraTrainConfig."""
    _fields_ = [("r", C.c_int32), ("alpha", C.c_float), ("dropout", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("seed", C.c_uint64)]


def lru_desc_from_state_dict(sd) -> tuple[LrLruWeightsDesc, list]:
    """Build the C weight descriptor from an LRURec state_dict (torch tensors or numpy arrays).

    Returns (desc, keepalive): the arrays referenced by `desc` live in `keepalive`.
    """
    keep: list[np.ndarray] = []

    def ptr(a, shape):
        a = _as_f32(a)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"LRURec weight has shape {a.shape}, expected {shape}")
        keep.append(a)
        return a.ctypes.data_as(c_float_p)

    emb = _as_f32(sd["embedding.token.weight"])
    if emb.ndim != 2 or emb.shape[1] != 64:
        raise ValueError(f"embedding.token.weight must be [V+1, 64], got {emb.shape}")
    n_rows = emb.shape[0]
    nb = 0
    while f"model.lru_blocks.{nb}.lru_layer.params_log" in sd:
        nb += 1
    if not 1 <= nb <= LR_MAX_LRU_BLOCKS:
        raise ValueError(f"unsupported number of LRU blocks: {nb}")
    d = LrLruWeightsDesc()
    d.num_items = n_rows - 1
    d.hidden = 64
    d.num_blocks = nb
    d.item_emb = ptr(emb, (n_rows, 64))
    d.item_bias = ptr(sd["model.bias"], (n_rows,))
    d.emb_ln_w = ptr(sd["embedding.layer_norm.weight"], (64,))
    d.emb_ln_b = ptr(sd["embedding.layer_norm.bias"], (64,))
    for b in range(nb):
        for field, key in _BLOCK_KEYS.items():
            setattr(d.blocks[b], field, ptr(sd[f"model.lru_blocks.{b}.{key}"], _BLOCK_SHAPES[field]))
    return d, keep
