"""Loader for libllamarec_mi355x.so (the C-ABI HIP library). There is NO fallback: if the
library is missing or a call fails, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
# LLAMAREC_LIB selects another BUILD of the same library (same-box A/B runs of two commits: tools/gpu_ab_lib.sh) without
# overwriting the product file; it is never a fallback -- a missing file raises like the default one.
LIB_PATH = os.environ.get("LLAMAREC_LIB") or os.path.join(_HERE, "lib", "libllamarec_mi355x.so")
_lib = None


class LlamaRecError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol include/llamarec_mi355x.h declares
PROTOTYPES = {
    "lr_last_error": (C.c_char_p, []),
    "lr_version": (C.c_char_p, []),
    "lr_lru_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "lr_lru_pack": (C.c_int, [C.POINTER(A.LrLruWeightsDesc), C.c_void_p, C.c_size_t]),
    "lr_lru_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "lr_lru_destroy": (None, [C.c_void_p]),
    "lr_lru_set_encoder_pipeline": (C.c_int, [C.c_void_p, C.c_int32]),
    "lr_lru_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "lr_lru_encode_last": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_lru_retrieve_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_lru_topk_path": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_int32), C.c_void_p]),
    "lr_lru_scores_last": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_rank_histogram": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "lr_rank_classes": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lr_profile_start": (C.c_int, [C.c_int32]),
    "lr_profile_stop": (C.c_int, []),
    "lr_profile_collect": (C.c_int, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_int64)]),
    "lr_profile_records": (C.c_int64, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "lr_gemm_bf16_nt_epi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_gemm_bf16_nt_residual_rmsnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                   C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_int32,
                                                   C.POINTER(C.c_int32), C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_rope_table_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "lr_rope_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "lr_metrics_from_histogram": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "lr_lru_train_state_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "lr_lru_train_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "lr_lru_train_destroy": (None, [C.c_void_p]),
    "lr_lru_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "lr_lru_train_loss_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_lru_train_apply": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "lr_lru_train_set_graph": (C.c_int, [C.c_void_p, C.c_int32]),
    "lr_lru_train_set_fused": (C.c_int, [C.c_void_p, C.c_int32]),
    "lr_lru_train_set_deterministic": (C.c_int, [C.c_void_p, C.c_int32]),
    "lr_lru_train_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_size_t)]),
    "lr_lru_train_param_range": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lr_llama_create": (C.c_int, [C.POINTER(A.LrLlamaConfig), C.POINTER(A.LrLlamaWeightsDesc),
                                  C.POINTER(C.c_void_p)]),
    "lr_llama_destroy": (None, [C.c_void_p]),
    "lr_llama_set_variants": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "lr_llama_set_last_layer_pruning": (C.c_int, [C.c_void_p, C.c_int32]),
    "lr_fold_norm_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "lr_llama_set_folded_norms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "lr_llama_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "lr_llama_prefill_verbalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                             C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.c_void_p]),
    "lr_llama_prefill_verbalize_prefix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                    C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                                    C.c_void_p]),
    "lr_common_prefix_len": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32]),
    "lr_llama_last_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_llama_pack_gate_up": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "lr_llama_pack_qkv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p]),
    "lr_gemm_bf16_nt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p]),
    "lr_gemm_bf16_nt_ws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_attention_varlen": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "lr_attention_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "lr_attention_varlen_ws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lr_nf4_scratch_bytes": (C.c_size_t, [C.c_size_t]),
    "lr_nf4_roundtrip_bf16": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "lr_nf4_dynamic_map": (C.c_int, [C.POINTER(C.c_float)]),
    "lr_transpose_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "lr_llama_lora_state_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(A.LrLoraTrainConfig)]),
    "lr_llama_lora_create": (C.c_int, [C.c_void_p, C.POINTER(A.LrLlamaWeightsTDesc), C.POINTER(A.LrLoraTrainConfig),
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "lr_llama_lora_destroy": (None, [C.c_void_p]),
    "lr_llama_lora_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "lr_llama_lora_param_range": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_size_t)]),
    "lr_llama_lora_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "lr_llama_lora_loss_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_void_p]),
    "lr_llama_lora_apply": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "lr_llama_lora_eval_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "lr_llama_lora_prefill_verbalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                                  C.c_void_p]),
    "lr_attention_varlen_lse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "lr_attention_bwd_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "lr_attention_varlen_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_size_t, C.c_void_p]),
}


def lib():
    """The loaded library with prototypes set. Raises LlamaRecError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LlamaRecError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        # torch first: it bundles its own libamdhip64, and whichever HIP runtime a process loads FIRST is the one that owns
        # the devices -- with this library (linked against /opt/rocm's runtime) loaded before torch, its calls then fail with
        # "no ROCm-capable device is detected" (seen when build() and smoke() ran in one process). Loaded after torch the
        # library binds to the runtime torch already brought in, whatever the import order of the caller.
        import torch  # noqa: F401

        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().lr_last_error().decode("utf-8", "replace")
        raise LlamaRecError(f"{what} failed (rc={rc}): {msg}")


def stream_ptr(stream=None) -> int:
    import torch

    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)
