"""CPU restatement of the reference's LRURec TRAINING step -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Path restated:
  loss      LRUTrainer.calculate_loss                 trainer/lru.py:20-28  (CE over all positions, ignore_index=0)
  forward   LRURec / LRUEmbedding / LRUModel / LRULayer / PositionwiseFeedForward   model/lru.py:38-175
            (the recursive-doubling scan in its sequential form h_t = m_{t-1} lambda h_{t-1} + u_t, which is what
             it computes on left-padded batches: dataloader/lru.py:119-131)
  backward  torch autograd on that graph (restated by hand below; complex parameters carry the gradient
            dL/dRe + i dL/dIm, which is what torch stores in .grad and what AdamW consumes via view_as_real)
  update    clip_grad_norm_ + AdamW with the reference's two parameter groups   trainer/base.py:106-112,201-246

Pinned by tests/golden/lru_train_v120.npz (made by tests/gen_goldens_train.py from the reference itself): loss,
every gradient, and every parameter after two optimizer steps. Arithmetic is float64 here (the checker), the
reference and the HIP path are float32; tests state the tolerance.

State: dict name -> numpy array with the reference's state_dict names; complex tensors as [..., 2] (re, im).
"""
from __future__ import annotations

import math

import numpy as np

LN_EPS = 1e-5
NO_DECAY = ("bias", "layer_norm")  # trainer/base.py:222


def _c(a):
    return a[..., 0].astype(np.float64) + 1j * a[..., 1].astype(np.float64)


def _pair(z):
    return np.stack([z.real, z.imag], axis=-1)


def _ln_fwd(x, w, b):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + LN_EPS)
    xhat = (x - mu) * rstd
    return xhat * w + b, (xhat, rstd)


def _ln_bwd(dy, cache, w):
    xhat, rstd = cache
    dw = (dy * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    db = dy.reshape(-1, xhat.shape[-1]).sum(0)
    dxh = dy * w
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xhat * (dxh * xhat).mean(-1, keepdims=True))
    return dx, dw, db


_erf = np.vectorize(math.erf)


def _gelu(a):
    return 0.5 * a * (1.0 + _erf(a / math.sqrt(2.0)))


def _gelu_grad(a):
    return 0.5 * (1.0 + _erf(a / math.sqrt(2.0))) + a * np.exp(-0.5 * a * a) / math.sqrt(2.0 * math.pi)


def num_blocks(state):
    n = 0
    while f"model.lru_blocks.{n}.lru_layer.params_log" in state:
        n += 1
    return n


def drop_scale(seed, site, n, p):
    """The HIP step's counter-based dropout (csrc/lru_train.hip: tr_drop_scale), restated: float64 [n] of 0 or
    1 / (1 - p) for flat element indices 0..n-1 of dropout site `site` under 64-bit `seed`."""
    C1, C2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xD6E8FEB86659FD93)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) ^ (C1 * np.uint64(site + 1)) ^ (np.arange(n, dtype=np.uint64) * C2)
        for _ in range(2):
            x ^= x >> np.uint64(32)
            x = x * C2
        x ^= x >> np.uint64(32)
    u = (x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.where(u < np.float32(p), 0.0, 1.0 / (1.0 - float(np.float32(p))))


def pass_seed(cfg_seed, pass_no):
    """Dropout seed of forward pass `pass_no` (0-based) of an engine created with `seed=cfg_seed` (tr_begin_pass)."""
    m = (1 << 64) - 1
    return ((cfg_seed * 0x9E3779B97F4A7C15) + pass_no * 0xA24BAED4963EE407) & m


def loss_and_grads(state, tokens, labels, dropout=None):
    """-> (loss, grads dict with the state's names/shapes). tokens/labels: int [B, L].
    dropout: optional (seed64, p, p_attn): apply the masks of the HIP step's stream at the reference's four dropout
    sites (model/lru.py: embedding, LRU-layer output, FFN activation, FFN output) -- torch's masks are not reproducible,
    the positions and the 1 / (1 - p) scaling are the reference's."""
    P = {k: np.asarray(v, dtype=np.float64) for k, v in state.items()}
    tokens = np.asarray(tokens)
    labels = np.asarray(labels)
    B, L = tokens.shape
    m = (tokens > 0).astype(np.float64)              # model/lru.py:51-52
    E = P["embedding.token.weight"]
    nb = num_blocks(state)
    G = {k: np.zeros_like(v) for k, v in P.items()}

    def mask(site, shape, p):
        if dropout is None or p <= 0:
            return 1.0
        return drop_scale(dropout[0], site, int(np.prod(shape)), p).reshape(shape)

    pd, pa = (dropout[1], dropout[2]) if dropout is not None else (0.0, 0.0)
    # ---- forward
    M0 = mask(0, (B, L, 64), pd)
    x, ln0 = _ln_fwd(E[tokens] * M0, P["embedding.layer_norm.weight"], P["embedding.layer_norm.bias"])
    caches = []
    for b in range(nb):
        pre = f"model.lru_blocks.{b}."
        nu, th, ga = np.exp(P[pre + "lru_layer.params_log"])     # model/lru.py:151
        lam = np.exp(-nu + 1j * th)                               # :152
        Win, bin_ = _c(P[pre + "lru_layer.in_proj.weight"]), _c(P[pre + "lru_layer.in_proj.bias"])
        Wout, bout = _c(P[pre + "lru_layer.out_proj.weight"]), _c(P[pre + "lru_layer.out_proj.bias"])
        p = x @ Win.T + bin_                                      # [B, L, 128] complex
        u = p * ga                                                # :153
        h = np.zeros_like(u)
        for t in range(L):
            h[:, t] = u[:, t] + (lam * h[:, t - 1] * m[:, t - 1, None] if t > 0 else 0.0)
        Mo, Mg, Mz = mask(10 + 4 * b, (B, L, 64), pa), mask(11 + 4 * b, (B, L, 256), pd), mask(12 + 4 * b, (B, L, 64), pd)
        o = (h @ Wout.T + bout).real * Mo                         # :160, dropout
        y, ln1 = _ln_fwd(o + x, P[pre + "lru_layer.layer_norm.weight"], P[pre + "lru_layer.layer_norm.bias"])
        a = y @ P[pre + "feed_forward.w_1.weight"].T + P[pre + "feed_forward.w_1.bias"]
        g = _gelu(a) * Mg
        z0 = (g @ P[pre + "feed_forward.w_2.weight"].T + P[pre + "feed_forward.w_2.bias"]) * Mz + y
        xn, ln2 = _ln_fwd(z0, P[pre + "feed_forward.layer_norm.weight"], P[pre + "feed_forward.layer_norm.bias"])
        caches.append(dict(x=x, p=p, h=h, lam=lam, nu=nu, th=th, ga=ga, Win=Win, Wout=Wout, ln1=ln1, y=y, a=a, g=g, ln2=ln2,
                           Mo=Mo, Mg=Mg, Mz=Mz))
        x = xn
    scores = x @ E.T + P["model.bias"]                            # model/lru.py:85
    valid = labels != 0                                           # CrossEntropyLoss(ignore_index=0)
    n_valid = max(int(valid.sum()), 1)
    mx = scores.max(-1, keepdims=True)
    ex = np.exp(scores - mx)
    lse = np.log(ex.sum(-1)) + mx[..., 0]
    picked = np.take_along_axis(scores, labels[..., None], -1)[..., 0]
    loss = float(((lse - picked) * valid).sum() / n_valid)

    # ---- backward
    ds = ex / ex.sum(-1, keepdims=True)
    np.put_along_axis(ds, labels[..., None], np.take_along_axis(ds, labels[..., None], -1) - 1.0, -1)
    ds *= (valid / n_valid)[..., None]
    G["model.bias"] = ds.reshape(-1, ds.shape[-1]).sum(0)
    G["embedding.token.weight"] += ds.reshape(-1, ds.shape[-1]).T @ x.reshape(-1, 64)
    dx = ds @ E
    for b in reversed(range(nb)):
        pre = f"model.lru_blocks.{b}."
        c = caches[b]
        dz0, G[pre + "feed_forward.layer_norm.weight"], G[pre + "feed_forward.layer_norm.bias"] = _ln_bwd(
            dx, c["ln2"], P[pre + "feed_forward.layer_norm.weight"])
        W1, W2 = P[pre + "feed_forward.w_1.weight"], P[pre + "feed_forward.w_2.weight"]
        dzw = dz0 * c["Mz"]                                        # the W2 branch sits behind a dropout
        G[pre + "feed_forward.w_2.weight"] = dzw.reshape(-1, 64).T @ c["g"].reshape(-1, 256)
        G[pre + "feed_forward.w_2.bias"] = dzw.reshape(-1, 64).sum(0)
        da = (dzw @ W2) * c["Mg"] * _gelu_grad(c["a"])
        G[pre + "feed_forward.w_1.weight"] = da.reshape(-1, 256).T @ c["y"].reshape(-1, 64)
        G[pre + "feed_forward.w_1.bias"] = da.reshape(-1, 256).sum(0)
        dy = dz0 + da @ W1
        dy0, G[pre + "lru_layer.layer_norm.weight"], G[pre + "lru_layer.layer_norm.bias"] = _ln_bwd(
            dy, c["ln1"], P[pre + "lru_layer.layer_norm.weight"])
        # o = Re(W h + b): gradient pairs (d/dRe + i d/dIm)
        Wout, h = c["Wout"], c["h"]
        dres = dy0                                                # the residual branch is not dropped
        dy0 = dy0 * c["Mo"]
        do = dy0.reshape(-1, 64)
        hf = h.reshape(-1, 128)
        G[pre + "lru_layer.out_proj.weight"] = _pair(do.T @ hf.real - 1j * (do.T @ hf.imag))
        G[pre + "lru_layer.out_proj.bias"] = _pair(do.sum(0) + 0j)
        gh = dy0 @ Wout.real - 1j * (dy0 @ Wout.imag)            # direct gradient of every h_t
        lam = c["lam"]
        Gt = np.zeros((B, 128), dtype=np.complex128)
        du = np.zeros_like(h)
        dlam = np.zeros(128, dtype=np.complex128)
        for t in reversed(range(L)):
            Gt = gh[:, t] + (np.conj(lam) * Gt * m[:, t, None] if t < L - 1 else 0.0)  # via h_{t+1} = .. + lam h_t m_t
            du[:, t] = Gt
            if t > 0:
                dlam += (np.conj(h[:, t - 1]) * Gt * m[:, t - 1, None]).sum(0)
        dnu = -(dlam * np.conj(lam)).real                         # d lam / d nu = -lam
        dth = (dlam * np.conj(1j * lam)).real                     # d lam / d theta = i lam
        dga = (du * np.conj(c["p"])).real.reshape(-1, 128).sum(0)
        G[pre + "lru_layer.params_log"] = np.stack([dnu * c["nu"], dth * c["th"], dga * c["ga"]])
        dp = (du * c["ga"]).reshape(-1, 128)
        xf = c["x"].reshape(-1, 64)
        G[pre + "lru_layer.in_proj.weight"] = _pair(dp.real.T @ xf + 1j * (dp.imag.T @ xf))
        G[pre + "lru_layer.in_proj.bias"] = _pair(dp.sum(0))
        Win = c["Win"]
        dx = dres + (dp.real @ Win.real + dp.imag @ Win.imag).reshape(B, L, 64)
    de, G["embedding.layer_norm.weight"], G["embedding.layer_norm.bias"] = _ln_bwd(dx, ln0, P["embedding.layer_norm.weight"])
    np.add.at(G["embedding.token.weight"], tokens.reshape(-1), (de * M0).reshape(-1, 64))
    return loss, G


class AdamW:
    """torch.optim.AdamW (decoupled decay, bias-corrected) over the reference's two groups; complex tensors are
    updated component-wise like torch's view_as_real path."""

    def __init__(self, state, lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-9):
        self.lr, self.wd, self.b1, self.b2, self.eps, self.t = lr, weight_decay, betas[0], betas[1], eps, 0
        self.m = {k: np.zeros(np.shape(v), np.float64) for k, v in state.items()}
        self.v = {k: np.zeros(np.shape(v), np.float64) for k, v in state.items()}

    def step(self, state, grads, max_grad_norm=5.0):
        """-> (new state, total gradient norm before clipping). trainer/base.py:109-110."""
        norm = math.sqrt(sum(float((np.asarray(g, np.float64) ** 2).sum()) for g in grads.values()))
        coef = min(1.0, max_grad_norm / (norm + 1e-6))            # torch.nn.utils.clip_grad_norm_
        self.t += 1
        out = {}
        for k, p in state.items():
            p = np.asarray(p, np.float64)
            g = np.asarray(grads[k], np.float64) * coef
            if not any(nd in k for nd in NO_DECAY):
                p = p * (1.0 - self.lr * self.wd)
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            denom = np.sqrt(self.v[k]) / math.sqrt(1 - self.b2 ** self.t) + self.eps
            out[k] = p - (self.lr / (1 - self.b1 ** self.t)) * self.m[k] / denom
        return out, norm
