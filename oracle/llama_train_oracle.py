"""ORACLE for the ranker's LoRA training step. TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A float64 CPU restatement, one UNPADDED prompt at a time (left padding is masked in the reference, so packed
execution computes the same numbers -- SURVEY.md 8(a) a15), with torch autograd standing in for the reference's own
autograd. Restates:
  patched LlamaForCausalLM.forward, training branch: logits of every position, shift by one, CrossEntropyLoss
      (ignore_index -100, mean over the labelled tokens of the whole micro-batch)            model/llm.py:113-127
  labels: `labels[:-2] = -100` -- only the answer letter and EOS are labelled          dataloader/llm.py:55-58
  LlamaModel body: transformers' modeling_llama.py (pinned 4.42.3, environment.yml:304), see oracle/llama_oracle.py
  LoRA on q_proj / v_proj: peft 0.11.1 (environment.yml:248, absent here): y = W x + (alpha/r) B A dropout(x);
      r = 8, alpha = 32, dropout 0.05                                  config.py:257-260, train_ranker.py:71-79
  clipping + AdamW: HF TrainingArguments defaults (max_grad_norm 1.0, betas 0.9/0.999, eps 1e-8, weight_decay 0;
      the reference's 8-bit bitsandbytes state is not restated)                             trainer/llm.py:103-136
Pinned by tests/golden/llama_lora_train_*.npz (tests/gen_goldens_rank_train.py runs the reference's forward).
"""
from __future__ import annotations

import numpy as np
import torch


def _rms(x, w, eps):
    return w * (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps))


def _rope(x, cos, sin):
    h = x.shape[-1] // 2
    x1, x2 = x[..., :h], x[..., h:]
    c, s = cos[:, None, :], sin[:, None, :]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)


def loss_and_grads(sd, cfg, lora, seqs, labels, r=8, alpha=32, dtype=torch.float64, drop_masks=None):
    """sd: HF-named float arrays of the frozen base; lora: {"layers.{l}.{q,v}_proj.lora_{A,B}": array};
    seqs / labels: lists of int lists (labels -100 = ignored). Returns (loss, {name: grad array}).
    drop_masks: optional list (one per layer) of [total_tokens, hidden] arrays holding 0 or 1 / (1 - p): the adapters'
    input dropout (peft: lora_dropout on x before lora_A), rows in packed order."""
    d, nh, nkv = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd, eps, theta = d // nh, cfg["rms_norm_eps"], cfg["rope_theta"]
    W = {k: torch.from_numpy(np.asarray(v, np.float64)).to(dtype) for k, v in sd.items()}
    P = {k: torch.from_numpy(np.asarray(v, np.float64)).to(dtype).requires_grad_(True) for k, v in lora.items()}
    scaling = alpha / r
    total, count = torch.zeros((), dtype=dtype), 0
    row0 = 0
    for ids, lab in zip(seqs, labels):
        T = len(ids)
        inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float64) / hd))
        ang = torch.arange(T, dtype=torch.float64)[:, None] * inv[None, :]
        cos, sin = torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)
        x = W["model.embed_tokens.weight"][torch.as_tensor(ids)]
        causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
        for i in range(cfg["num_hidden_layers"]):
            p = f"model.layers.{i}."
            xn = _rms(x, W[p + "input_layernorm.weight"], eps)
            xd = xn
            if drop_masks is not None:
                xd = xn * torch.from_numpy(np.asarray(drop_masks[i][row0:row0 + T], np.float64)).to(dtype)
            q = xn @ W[p + "self_attn.q_proj.weight"].T
            q = q + scaling * ((xd @ P[f"layers.{i}.q_proj.lora_A"].T) @ P[f"layers.{i}.q_proj.lora_B"].T)
            k = xn @ W[p + "self_attn.k_proj.weight"].T
            v = xn @ W[p + "self_attn.v_proj.weight"].T
            v = v + scaling * ((xd @ P[f"layers.{i}.v_proj.lora_A"].T) @ P[f"layers.{i}.v_proj.lora_B"].T)
            q, k, v = q.reshape(T, nh, hd), k.reshape(T, nkv, hd), v.reshape(T, nkv, hd)
            q, k = _rope(q, cos, sin), _rope(k, cos, sin)
            rep = nh // nkv
            k, v = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
            s = torch.einsum("qhd,khd->hqk", q, k) / np.sqrt(hd)
            s = s.masked_fill(~causal[None], float("-inf"))
            o = torch.einsum("hqk,khd->qhd", torch.softmax(s, dim=-1), v).reshape(T, nh * hd)
            x = x + o @ W[p + "self_attn.o_proj.weight"].T
            xn = _rms(x, W[p + "post_attention_layernorm.weight"], eps)
            g = xn @ W[p + "mlp.gate_proj.weight"].T
            u = xn @ W[p + "mlp.up_proj.weight"].T
            x = x + (torch.nn.functional.silu(g) * u) @ W[p + "mlp.down_proj.weight"].T
        logits = _rms(x, W["model.norm.weight"], eps) @ W["lm_head.weight"].T
        tgt = torch.as_tensor(lab[1:], dtype=torch.long)            # tokens < n predict n (model/llm.py:118-120)
        keep = tgt != -100
        if keep.any():
            lp = torch.log_softmax(logits[:-1][keep], dim=-1)
            total = total - lp[torch.arange(int(keep.sum())), tgt[keep]].sum()
            count += int(keep.sum())
        row0 += T
    loss = total / max(count, 1)
    loss.backward()
    return float(loss.detach()), {k: (v.grad.numpy().astype(np.float64) if v.grad is not None else np.zeros(v.shape))
                         for k, v in P.items()}


def clip_and_adamw(params, grads, m, v, step, lr, max_grad_norm, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """clip_grad_norm_ over all LoRA tensors, then torch.optim.AdamW's update (step counts from 1). In place on the
    float64 dicts; returns the gradient norm before clipping."""
    norm = float(np.sqrt(sum(float((g * g).sum()) for g in grads.values())))
    coef = min(1.0, max_grad_norm / (norm + 1e-6)) if max_grad_norm > 0 else 1.0
    for k in params:
        g = grads[k] * coef
        params[k] *= 1.0 - lr * weight_decay
        m[k] = beta1 * m[k] + (1 - beta1) * g
        v[k] = beta2 * v[k] + (1 - beta2) * g * g
        mh = m[k] / (1 - beta1 ** step)
        vh = v[k] / (1 - beta2 ** step)
        params[k] -= lr * mh / (np.sqrt(vh) + eps)
    return norm


# ---- the HIP step's counter-based dropout stream (csrc/llama_train.hip: lt_keep / lr_lora_drop_stream), restated so
# ---- that a test can hand the oracle the very mask the kernels use (torch's own masks are not reproducible)
def _mix32(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def drop_stream(seed, pass_no, layer):
    m = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * ((pass_no * 131 + layer + 1) & m)) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return (z ^ (z >> 31)) & 0xFFFFFFFF


def drop_mask(seed, pass_no, layer, n_rows, d, p):
    """[n_rows, d] float64: 1 / (1 - p) where the element is kept, 0 where it is dropped."""
    thresh = min(int(p * 4294967296.0), 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        rows = np.arange(n_rows, dtype=np.uint32)[:, None]
        cols = np.arange(d, dtype=np.uint32)[None, :]
        stream = np.uint32(drop_stream(seed, pass_no, layer))
        h = _mix32(_mix32(stream ^ (rows * np.uint32(0x9E3779B9))) + cols * np.uint32(0x7F4A7C15))
    return np.where(h >= np.uint32(thresh), 1.0 / (1.0 - p), 0.0)
