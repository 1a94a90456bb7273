"""numpy ORACLE for stage 2: Llama prefill -> last-position logits -> verbalizer gather.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates (paths into /root/reference and its pinned third-party dependency):
  patched LlamaForCausalLM.forward: lm_head over the hidden states, `.float()`, `logits[:, -1]`,
      eval loss = -1.0                                            model/llm.py:89-131
  LlamaModel body = transformers (pinned 4.42.3, environment.yml:304; 5.15.0 installed here),
      models/llama/modeling_llama.py: RMSNorm with fp32 statistics, rotate-half RoPE,
      causal softmax attention (scores / sqrt(head_dim)), SwiGLU MLP, pre-norm residuals.
      The reference holds no tests at this boundary; the oracle is pinned against outputs of the
      reference's patched forward on tiny random Llama configs (tests/golden/llama_*.npz).
  ManualVerbalizer.process_logits with the reference's settings == logits[:, label_ids]
      trainer/verb.py:524-544,546-586,602-614 (pinned by tests/golden/verbalizer.npz)

mode="fp32": everything in float32 (compared with the fp32 goldens at ~1e-5).
mode="bf16": weights/activations rounded to bfloat16 at the points where the HIP kernels round
      (after every projection, norm, RoPE, attention output, residual add, SwiGLU), fp32
      accumulation -- the reference's bf16_full_eval arithmetic (trainer/llm.py:113).
"""
from __future__ import annotations

import numpy as np

from llamarec_amd.synth import bf16_round


def _rms_norm(x, w, eps, rnd):
    xf = x.astype(np.float32)
    var = np.mean(xf * xf, axis=-1, keepdims=True, dtype=np.float32)
    xn = rnd(xf * (np.float32(1.0) / np.sqrt(var + np.float32(eps))))
    return rnd(w * xn)


def rope_tables(T, hd, theta, rnd):
    inv = (np.float32(1.0) / (np.float32(theta) ** (np.arange(0, hd, 2, dtype=np.float32) / np.float32(hd)))).astype(np.float32)
    ang = np.arange(T, dtype=np.float32)[:, None] * inv[None, :]
    return rnd(np.cos(ang).astype(np.float32)), rnd(np.sin(ang).astype(np.float32))


def _rope(x, cos, sin, rnd):
    # x [T, H, hd]; rotate_half convention: (x1, x2) -> (x1*c - x2*s, x2*c + x1*s)
    h = x.shape[-1] // 2
    x1, x2 = x[..., :h], x[..., h:]
    c, s = cos[:, None, :], sin[:, None, :]
    return rnd(np.concatenate([x1 * c - x2 * s, x2 * c + x1 * s], axis=-1))


def forward_hidden(sd, cfg, ids, mode="fp32"):
    """Hidden states [T, d] after the final norm for ONE unpadded prompt `ids` [T]."""
    rnd = bf16_round if mode == "bf16" else (lambda a: a.astype(np.float32))
    d, nh, nkv = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = d // nh
    eps, theta = cfg["rms_norm_eps"], cfg["rope_theta"]
    T = len(ids)
    cos, sin = rope_tables(T, hd, theta, rnd)
    x = rnd(sd["model.embed_tokens.weight"][np.asarray(ids)])
    causal = np.tril(np.ones((T, T), bool))
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        xn = _rms_norm(x, sd[p + "input_layernorm.weight"], eps, rnd)
        q = rnd(xn @ sd[p + "self_attn.q_proj.weight"].T).reshape(T, nh, hd)
        k = rnd(xn @ sd[p + "self_attn.k_proj.weight"].T).reshape(T, nkv, hd)
        v = rnd(xn @ sd[p + "self_attn.v_proj.weight"].T).reshape(T, nkv, hd)
        q, k = _rope(q, cos, sin, rnd), _rope(k, cos, sin, rnd)
        rep = nh // nkv
        o = np.empty((T, nh, hd), np.float32)
        for h in range(nh):
            kh, vh = k[:, h // rep], v[:, h // rep]
            s = (q[:, h] @ kh.T) * np.float32(1.0 / np.sqrt(hd))
            s = np.where(causal, s, np.float32(-np.inf))
            m = s.max(axis=-1, keepdims=True)
            e = np.exp(s - m).astype(np.float32)
            l = e.sum(axis=-1, keepdims=True, dtype=np.float32)
            if mode == "bf16":
                o[:, h] = (rnd(e) @ vh) / l  # flash-style: unnormalised P in bf16, fp32 row sum
            else:
                o[:, h] = (e / l) @ vh
        o = rnd(o.reshape(T, nh * hd))
        x = rnd(x + rnd(o @ sd[p + "self_attn.o_proj.weight"].T))
        xn = _rms_norm(x, sd[p + "post_attention_layernorm.weight"], eps, rnd)
        g = rnd(xn @ sd[p + "mlp.gate_proj.weight"].T)
        u = rnd(xn @ sd[p + "mlp.up_proj.weight"].T)
        act = rnd(g / (np.float32(1.0) + np.exp(-g)))
        x = rnd(x + rnd(rnd(act * u) @ sd[p + "mlp.down_proj.weight"].T))
    return _rms_norm(x, sd["model.norm.weight"], eps, rnd)


def last_logits(sd, cfg, seqs, mode="fp32"):
    """fp32 [B, vocab]: logits at each prompt's last token (model/llm.py:113-114,131)."""
    rnd = bf16_round if mode == "bf16" else (lambda a: a.astype(np.float32))
    out = []
    for ids in seqs:
        h = forward_hidden(sd, cfg, ids, mode)[-1]
        out.append(rnd(sd["lm_head.weight"] @ h).astype(np.float32))
    return np.stack(out)


def verbalize(logits, label_token_ids):
    """ManualVerbalizer.process_logits for one single-token word per class, prefix "",
    post_log_softmax=False (trainer/llm.py:93-101) == a column gather."""
    return np.ascontiguousarray(logits[:, np.asarray(label_token_ids)])


def prefill_verbalize(sd, cfg, seqs, label_token_ids, mode="bf16"):
    return verbalize(last_logits(sd, cfg, seqs, mode), label_token_ids)


def prefill_flops(cfg, T):
    """Algorithmic FLOPs of one prompt of T tokens: linear layers for every token,
    causal attention, verbalizer rows ignored."""
    d, f, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = d // nh
    lin = 2 * (d * (nh + 2 * nkv) * hd + nh * hd * d + 3 * d * f)
    attn = 2 * 2 * nh * hd * (T * (T + 1) / 2)
    return L * (T * lin + attn)
