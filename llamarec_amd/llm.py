"""Llama-2 ranker -- host-side mirror of the reference's patched `LlamaForCausalLM` for the
SCORING path, backed by the HIP prefill kernels (no torch compute in the forward, no fallback).

Reference interface being replaced (paths into the reference tree):
  LlamaForCausalLM.forward(input_ids[B,T], attention_mask[B,T], labels=...) ->
      CausalLMOutputWithPast(loss=tensor(-1.0), logits=fp32[B,vocab])   model/llm.py:35-145
  ManualVerbalizer.process_logits(logits) -> fp32[B,20]                 trainer/verb.py:546-586
  weights: bf16 compute over NF4-quantised base + LoRA(q_proj,v_proj)   train_ranker.py:49-79;
           here bf16 weights with the adapter merged at load (W + (alpha/r) B A).
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _abi as A
from ._lib import check, lib, stream_ptr

LLAMA2_7B = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=4096,
                 rms_norm_eps=1e-5, rope_theta=10000.0)


@dataclass
class CausalLMOutput:
    """Subset of transformers' CausalLMOutputWithPast that the scoring path reads."""
    loss: torch.Tensor | None
    logits: torch.Tensor
    past_key_values: None = None
    hidden_states: None = None
    attentions: None = None

    def __getitem__(self, i):
        return ((self.loss, self.logits) if self.loss is not None else (self.logits,))[i]


def pack_prompts(seqs) -> tuple[np.ndarray, np.ndarray]:
    """list of token-id sequences -> (packed int32 ids, cu_seqlens int32 [B+1])."""
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    if (lens < 1).any():
        raise ValueError("empty prompt")
    cu = np.zeros(len(seqs) + 1, np.int32)
    cu[1:] = np.cumsum(lens)
    ids = np.concatenate([np.asarray(s, dtype=np.int32).reshape(-1) for s in seqs])
    return ids, cu


def common_prefix_len(ids: np.ndarray, cu: np.ndarray) -> int:
    """Length of the longest token prefix shared by ALL prompts of a packed batch, capped at min(T) - 1 so every prompt
    keeps at least one token of its own (0 for a single prompt: nothing to share)."""
    B = len(cu) - 1
    if B < 2:
        return 0
    lens = np.diff(cu)
    n = int(lens.min()) - 1
    if n <= 0:
        return 0
    head = ids[cu[0]: cu[0] + n]
    for b in range(1, B):
        neq = np.nonzero(ids[cu[b]: cu[b] + n] != head[:n])[0]
        if neq.size:
            n = int(neq[0])
            if n == 0:
                return 0
    return n


def unpad_left(input_ids, attention_mask=None):
    """Left-padded [B,T] batch (trainer/llm.py:34-37 pads ids with 0, mask with 0) -> list of
    per-prompt id arrays."""
    ids = np.asarray(input_ids.cpu() if isinstance(input_ids, torch.Tensor) else input_ids)
    if attention_mask is None:
        return [row for row in ids]
    m = np.asarray(attention_mask.cpu() if isinstance(attention_mask, torch.Tensor) else attention_mask)
    out = []
    for r, mr in zip(ids, m):
        n = int(mr.sum())
        if n and not mr[-n:].all():
            raise ValueError("attention_mask must be left padding (ones form a suffix)")
        out.append(r[len(r) - n:])
    return out


def load_peft_adapter(adapter_path):
    """A local PEFT LoRA directory (adapter_config.json + adapter_model.safetensors) -> the `lora` argument of
    LlamaRanker.from_state_dict."""
    from safetensors import safe_open

    ac = json.load(open(os.path.join(adapter_path, "adapter_config.json")))
    w = {}
    with safe_open(os.path.join(adapter_path, "adapter_model.safetensors"), framework="pt", device="cpu") as f:
        for k in f.keys():
            w[k.replace("base_model.model.", "").replace(".default", "")] = f.get_tensor(k).float().numpy()
    return dict(r=ac["r"], alpha=ac["lora_alpha"], weights=w)


class LlamaRanker:
    """One replica of the ranker's weights on one MI355X + the prefill/verbalizer entry points."""

    def __init__(self, config: dict, device="cuda:0"):
        self.config = dict(config)
        self.device = torch.device(device)
        c = self.config
        self.hd = c.get("head_dim") or c["hidden_size"] // c["num_attention_heads"]
        self._h = C.c_void_p()
        self._tensors = {}
        self._ws = None
        self._layers_arr = None
        self._folded_arr = None
        # RMSNorm folded into the next projection (set_fold_norms): OFF by default. Numerically it is one more valid bf16
        # realisation (as close to the fp32 oracle as the separate pass: tools/parity_growth.py), but on this GEMM it does
        # not pay: the norm passes it removes (2.6 ms of a 163 ms step) cost less than the rstd sweep (1.3 ms) plus the
        # epilogue's dependent rstd load, which nothing hides at one workgroup per CU (GEMMs 1370 -> 1347 TF/s):
        # 164.2 vs 163.5 ms per step in a same-box A/B
        self.fold_norms = False
        self.training = False

    # -- construction ------------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, state_dict, config, device="cuda:0", lora=None, nf4=False):
        """state_dict: HF LlamaForCausalLM names -> torch tensors / numpy arrays (any float dtype).
        lora: optional dict(r=8, alpha=32, weights={"...q_proj.lora_A.weight": A, "...lora_B.weight": B}).
        nf4: pass every Linear of the base through the NF4 + double-quantisation round trip first, like the
        reference's BitsAndBytesConfig (train_ranker.py:49-56; lm_head, embeddings and norms stay as they are --
        bitsandbytes skips lm_head); the adapter is merged AFTER it, as peft adds it to the dequantised output."""
        self = cls(config, device)
        dev = self.device
        scratch = {}

        def nf4_roundtrip(w):
            w = w.to(torch.bfloat16).contiguous()
            need = lib().lr_nf4_scratch_bytes(w.numel())
            if scratch.get("n", 0) < need:
                scratch["buf"], scratch["n"] = torch.empty(need, dtype=torch.uint8, device=dev), need
            with torch.cuda.device(dev):
                check(lib().lr_nf4_roundtrip_bf16(w.data_ptr(), w.numel(), 1, w.data_ptr(), scratch["buf"].data_ptr(),
                                                  scratch["n"], stream_ptr()), "lr_nf4_roundtrip_bf16")
            return w

        def t(name):
            w = state_dict[name]
            if not isinstance(w, torch.Tensor):
                w = torch.from_numpy(np.ascontiguousarray(w))
            return w.to(dev)

        def linear(name):
            return nf4_roundtrip(t(name)) if nf4 else t(name)

        def merged(name):
            w = linear(name).float()
            if lora is not None:
                base = name[: -len(".weight")]
                ka, kb = base + ".lora_A.weight", base + ".lora_B.weight"
                if ka in lora["weights"]:
                    a = torch.as_tensor(np.asarray(lora["weights"][ka])).to(dev).float()
                    b = torch.as_tensor(np.asarray(lora["weights"][kb])).to(dev).float()
                    w = w + (lora["alpha"] / lora["r"]) * (b @ a)
            return w

        L = config["num_hidden_layers"]
        T = self._tensors
        T["embed"] = t("model.embed_tokens.weight").to(torch.bfloat16).contiguous()
        T["final_norm"] = t("model.norm.weight").to(torch.bfloat16).contiguous()
        T["lm_head"] = (t("lm_head.weight") if "lm_head.weight" in state_dict else T["embed"]).to(torch.bfloat16).contiguous()
        for i in range(L):
            p = f"model.layers.{i}."
            q, k, v = (merged(p + f"self_attn.{n}_proj.weight") for n in "qkv")
            T[f"{i}.wqkv"] = torch.cat([self._interleave_rope_rows(q), self._interleave_rope_rows(k), v],
                                       0).to(torch.bfloat16).contiguous()
            T[f"{i}.wo"] = linear(p + "self_attn.o_proj.weight").to(torch.bfloat16).contiguous()
            T[f"{i}.wgu"] = self._interleave_gate_up(linear(p + "mlp.gate_proj.weight"), linear(p + "mlp.up_proj.weight"))
            T[f"{i}.wdown"] = linear(p + "mlp.down_proj.weight").to(torch.bfloat16).contiguous()
            T[f"{i}.input_norm"] = t(p + "input_layernorm.weight").to(torch.bfloat16).contiguous()
            T[f"{i}.post_norm"] = t(p + "post_attention_layernorm.weight").to(torch.bfloat16).contiguous()
        self._create()
        return self

    @classmethod
    def random_init(cls, config, seed=42, std=0.02, device="cuda:0"):
        """Random bf16 weights ~ N(0, std) generated directly on the GPU (BASELINE.md section 3:
        no checkpoints exist offline). Norm weights are ones."""
        self = cls(config, device)
        c, dev = self.config, self.device
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        d, f, v = c["hidden_size"], c["intermediate_size"], c["vocab_size"]
        nh, nkv, hd = c["num_attention_heads"], c["num_key_value_heads"], self.hd

        def rnd(*shape):
            return (torch.randn(*shape, generator=g, device=dev, dtype=torch.float32) * std).to(torch.bfloat16)

        T = self._tensors
        T["embed"] = rnd(v, d)
        T["final_norm"] = torch.ones(d, dtype=torch.bfloat16, device=dev)
        T["lm_head"] = rnd(v, d)
        for i in range(c["num_hidden_layers"]):
            T[f"{i}.wqkv"] = rnd((nh + 2 * nkv) * hd, d)
            T[f"{i}.wo"] = rnd(d, nh * hd)
            T[f"{i}.wgu"] = rnd(2 * f, d)  # already in the interleaved gate/up layout (random anyway)
            T[f"{i}.wdown"] = rnd(d, f)
            T[f"{i}.input_norm"] = torch.ones(d, dtype=torch.bfloat16, device=dev)
            T[f"{i}.post_norm"] = torch.ones(d, dtype=torch.bfloat16, device=dev)
        self._create()
        return self

    @classmethod
    def from_pretrained(cls, path, device="cuda:0", adapter_path=None, load_in_4bit=False):
        """Local HF directory (config.json + *.safetensors), optionally a PEFT LoRA adapter directory
        (adapter_config.json + adapter_model.safetensors) merged at load. load_in_4bit: the reference's NF4 round trip
        of the base Linears (see from_state_dict). No network access."""
        from safetensors import safe_open

        cfg = json.load(open(os.path.join(path, "config.json")))
        sd = {}
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".safetensors"):
                with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as f:
                    for k in f.keys():
                        sd[k] = f.get_tensor(k)
        return cls.from_state_dict(sd, cfg, device, load_peft_adapter(adapter_path) if adapter_path else None,
                                   nf4=load_in_4bit)

    def _interleave_rope_rows(self, w):
        """Rows of every head reordered to (0, hd/2, 1, hd/2+1, ...): rotation pairs become adjacent
        output columns for the fused rotary epilogue (== lr_llama_pack_qkv)."""
        hd = self.hd
        heads = w.shape[0] // hd
        return w.view(heads, 2, hd // 2, w.shape[1]).transpose(1, 2).reshape(heads * hd, w.shape[1])

    @staticmethod
    def _interleave_gate_up(gate, up):
        f, d = gate.shape
        if f % 16:
            raise ValueError("intermediate_size must be a multiple of 16")
        g = gate.to(torch.bfloat16).view(f // 16, 16, d)
        u = up.to(torch.bfloat16).view(f // 16, 16, d)
        return torch.stack([g, u], dim=1).reshape(2 * f, d).contiguous()

    def _create(self):
        c, T = self.config, self._tensors
        cfg = A.LrLlamaConfig(
            vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
            num_layers=c["num_hidden_layers"], num_heads=c["num_attention_heads"],
            num_kv_heads=c["num_key_value_heads"], head_dim=self.hd,
            max_positions=int(c.get("max_position_embeddings", 4096)), rms_eps=float(c["rms_norm_eps"]),
            rope_theta=float(c.get("rope_theta", 10000.0)))
        L = c["num_hidden_layers"]
        arr = (A.LrLlamaLayerWeights * L)()
        for i in range(L):
            for field in ("input_norm", "wqkv", "wo", "post_norm", "wgu", "wdown"):
                setattr(arr[i], field, T[f"{i}.{field}"].data_ptr())
        desc = A.LrLlamaWeightsDesc(embed=T["embed"].data_ptr(), final_norm=T["final_norm"].data_ptr(),
                                    lm_head=T["lm_head"].data_ptr(), layers=arr)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            check(lib().lr_llama_create(C.byref(cfg), C.byref(desc), C.byref(h)), "lr_llama_create")
        self._h, self._layers_arr = h, arr
        if self.fold_norms:
            self.set_fold_norms(True)

    def set_fold_norms(self, enable=True):
        """Fold the two RMSNorms of every layer into the following projection (include/llamarec_mi355x.h,
        lr_llama_set_folded_norms): wqkv * diag(input_norm) and wgu * diag(post_norm) are built once on the GPU and
        kept beside the originals (+66 % of the q/k/v/gate/up bytes; the originals serve LoRA fine-tuning and the
        pruned last layer). Opt-in: see the note at LlamaRanker.fold_norms."""
        T, L = self._tensors, self.config["num_hidden_layers"]
        if not enable:
            check(lib().lr_llama_set_folded_norms(self._h, None, None), "lr_llama_set_folded_norms")
            for i in range(L):
                T.pop(f"{i}.wqkv_folded", None)
                T.pop(f"{i}.wgu_folded", None)
            self._folded_arr = None
            self.fold_norms = False
            return self
        qa, ga = (C.c_void_p * L)(), (C.c_void_p * L)()
        with torch.cuda.device(self.device):
            for i in range(L):
                for name, norm, arr in (("wqkv", "input_norm", qa), ("wgu", "post_norm", ga)):
                    w = T[f"{i}.{name}"]
                    out = torch.empty_like(w)
                    check(lib().lr_fold_norm_bf16(w.data_ptr(), T[f"{i}.{norm}"].data_ptr(), w.shape[0], w.shape[1],
                                                  out.data_ptr(), stream_ptr()), "lr_fold_norm_bf16")
                    T[f"{i}.{name}_folded"] = out
                    arr[i] = out.data_ptr()
            torch.cuda.synchronize()
            check(lib().lr_llama_set_folded_norms(self._h, qa, ga), "lr_llama_set_folded_norms")
        self._folded_arr = (qa, ga)
        self.fold_norms = True
        return self

    def set_variants(self, gemm=0, attention=0):
        """Kernel selection (include/llamarec_mi355x.h): 0 = auto; gemm=5 = latency mode for the online
        single-user path (split-K where a short prompt would leave most CUs idle)."""
        check(lib().lr_llama_set_variants(self._h, gemm, attention), "lr_llama_set_variants")
        return self

    def set_last_layer_pruning(self, enable=True):
        check(lib().lr_llama_set_last_layer_pruning(self._h, int(bool(enable))), "lr_llama_set_last_layer_pruning")
        return self

    def eval(self):
        return self

    def __del__(self):
        try:
            if self._h.value:
                lib().lr_llama_destroy(self._h)
        except Exception:
            pass

    # -- scoring -----------------------------------------------------------------------------
    def _workspace(self, n_tokens, n_seqs):
        need = lib().lr_llama_workspace_bytes(self._h, n_tokens, n_seqs)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _packed(self, seqs):
        ids, cu = pack_prompts(seqs)
        return (torch.from_numpy(ids).to(self.device), torch.from_numpy(cu).to(self.device), cu)

    def prefill_verbalize_packed(self, ids_dev, cu_dev, cu_host, label_ids_dev, out=None, prefix_len=0):
        """Device-resident inputs (the benchmark's timed region starts here). prefix_len > 0: the caller has checked
        (common_prefix_len on the host ids) that all prompts start with the same prefix_len tokens; they are then run
        once per batch (bit-identical scores, lr_llama_prefill_verbalize_prefix)."""
        B, Cn = len(cu_host) - 1, label_ids_dev.numel()
        if out is None:
            out = torch.empty((B, Cn), dtype=torch.float32, device=self.device)
        ws = self._workspace(int(cu_host[-1]), B)
        with torch.cuda.device(self.device):
            check(lib().lr_llama_prefill_verbalize_prefix(
                self._h, ids_dev.data_ptr(), cu_dev.data_ptr(), cu_host.ctypes.data, B, int(prefix_len),
                label_ids_dev.data_ptr(), Cn, out.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr()),
                "lr_llama_prefill_verbalize_prefix")
        return out

    def prefill_verbalize(self, seqs, label_token_ids, share_prefix=True):
        """scores[b, c] = logit of label word c at the last token of prompt b (fp32 [B, C]). The prompts' common
        prefix (the template text) is evaluated once unless share_prefix is False."""
        ids_host, cu_host = pack_prompts(seqs)
        P = common_prefix_len(ids_host, cu_host) if share_prefix else 0
        lab = torch.as_tensor(np.asarray(label_token_ids, dtype=np.int32)).to(self.device)
        return self.prefill_verbalize_packed(torch.from_numpy(ids_host).to(self.device),
                                             torch.from_numpy(cu_host).to(self.device), cu_host, lab, prefix_len=P)

    def last_logits(self, seqs):
        ids, cu, cu_host = self._packed(seqs)
        B = len(cu_host) - 1
        out = torch.empty((B, self.config["vocab_size"]), dtype=torch.float32, device=self.device)
        ws = self._workspace(int(cu_host[-1]), B)
        with torch.cuda.device(self.device):
            check(lib().lr_llama_last_logits(self._h, ids.data_ptr(), cu.data_ptr(), cu_host.ctypes.data, B,
                                             out.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr()),
                  "lr_llama_last_logits")
        return out

    def forward(self, input_ids=None, attention_mask=None, labels=None, **_unused):
        """Patched-forward compatible call (model/llm.py:35-143): last-position logits in fp32;
        in eval with labels the loss is the constant -1.0 (model/llm.py:128-129)."""
        logits = self.last_logits(unpad_left(input_ids, attention_mask))
        loss = torch.tensor(-1.0) if labels is not None else None
        return CausalLMOutput(loss=loss, logits=logits)

    __call__ = forward
