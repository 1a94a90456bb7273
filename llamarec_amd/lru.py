"""LRURec retriever -- host-side mirror of the reference's `model.lru.LRURec` for the SCORING path,
backed by the HIP kernels in libllamarec_mi355x.so (no torch compute, no CPU fallback).

Reference interface being replaced (paths into the reference tree):
  LRURec(args).forward(x: int64[B,L]) -> fp32[B,L,V+1]        model/lru.py:38-41
  consumers take `[:, -1, :]`, mask the history to -1e9, top-k  trainer/lru.py:33-38,67-84,105-126,
                                                                 demo/inference.py:46-53
Weights: the reference state_dict (names/dtypes: SURVEY.md 8(a) a-W), e.g. the
`model_state_dict` entry of experiments/lru/<dataset>/models/best_acc_model.pth
(trainer/base.py:158-161, config.py:7).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _abi as A
from ._lib import check, lib, stream_ptr

MAX_TOPK = A.LR_MAX_TOPK


def init_lru_state_dict(num_items: int, seed: int = 42, num_blocks: int = 2) -> dict:
    """Random LRURec weights by the reference's init rule (model/lru.py:16-36,113-119):
    truncated normal (std 0.02, +-0.04) for everything except LayerNorm (ones / zeros) and
    params_log (ring init r in [0.8, 0.99], theta in [0, 2pi))."""
    from scipy.special import erfinv

    rng = np.random.default_rng(seed)
    std, lo, up = 0.02, -0.04, 0.04
    from math import erf, sqrt

    l = (1.0 + erf((lo / std) / sqrt(2.0))) / 2.0
    u = (1.0 + erf((up / std) / sqrt(2.0))) / 2.0

    def tn(*shape):
        x = rng.uniform(2 * l - 1, 2 * u - 1, size=shape)
        return (erfinv(x) * (std * sqrt(2.0))).astype(np.float32)

    def ctn(*shape):
        return (tn(*shape) + 1j * tn(*shape)).astype(np.complex64)

    sd = {
        "embedding.token.weight": tn(num_items + 1, 64),
        "embedding.layer_norm.weight": np.ones(64, np.float32),
        "embedding.layer_norm.bias": np.zeros(64, np.float32),
        "model.bias": tn(num_items + 1),
    }
    r_min, r_max = 0.8, 0.99
    for b in range(num_blocks):
        p = f"model.lru_blocks.{b}."
        u1, u2 = rng.uniform(size=128), rng.uniform(size=128)
        nu_log = np.log(-0.5 * np.log(u1 * (r_max**2 - r_min**2) + r_min**2))
        theta_log = np.log(u2 * np.pi * 2)
        lam_abs = np.exp(-np.exp(nu_log))
        gamma_log = np.log(np.sqrt(1 - lam_abs**2))
        sd[p + "lru_layer.params_log"] = np.vstack([nu_log, theta_log, gamma_log]).astype(np.float32)
        sd[p + "lru_layer.in_proj.weight"] = ctn(128, 64)
        sd[p + "lru_layer.in_proj.bias"] = ctn(128)
        sd[p + "lru_layer.out_proj.weight"] = ctn(64, 128)
        sd[p + "lru_layer.out_proj.bias"] = ctn(64)
        sd[p + "lru_layer.layer_norm.weight"] = np.ones(64, np.float32)
        sd[p + "lru_layer.layer_norm.bias"] = np.zeros(64, np.float32)
        sd[p + "feed_forward.w_1.weight"] = tn(256, 64)
        sd[p + "feed_forward.w_1.bias"] = tn(256)
        sd[p + "feed_forward.w_2.weight"] = tn(64, 256)
        sd[p + "feed_forward.w_2.bias"] = tn(64)
        sd[p + "feed_forward.layer_norm.weight"] = np.ones(64, np.float32)
        sd[p + "feed_forward.layer_norm.bias"] = np.zeros(64, np.float32)
    return sd


def pack_state_dict(state_dict) -> tuple[np.ndarray, int, int]:
    """Reference state_dict -> (packed float32 image for the kernels, num_items, num_blocks).
    Pure CPU (lr_lru_pack)."""
    desc, keep = A.lru_desc_from_state_dict(state_dict)
    nbytes = lib().lr_lru_packed_bytes(desc.num_items, desc.num_blocks)
    if nbytes == 0:
        raise ValueError("unsupported LRURec shape")
    img = np.empty(nbytes // 4, np.float32)
    check(lib().lr_lru_pack(C.byref(desc), img.ctypes.data, img.nbytes), "lr_lru_pack")
    del keep
    return img, int(desc.num_items), int(desc.num_blocks)


class LRURec:
    """Scoring-only LRURec on one MI355X.

    >>> model = LRURec.from_state_dict(torch.load(path)["model_state_dict"])
    >>> scores = model(seqs)[:, -1, :]                    # reference idiom, works unchanged
    >>> idx, sc = model.retrieve_topk(seqs, 20)           # fused path: scores never materialised
    """

    def __init__(self, args=None, state_dict=None, device="cuda:0"):
        self.args = args
        self.device = torch.device(device)
        self._h = C.c_void_p()
        self._img = None
        self._ws = None
        self._sd = None
        self.training = False
        if state_dict is None:
            if args is None:
                raise ValueError("LRURec needs args (num_items) or a state_dict")
            if getattr(args, "bert_hidden_units", 64) != 64:
                raise NotImplementedError("only bert_hidden_units=64 is implemented (config.py:212)")
            state_dict = init_lru_state_dict(args.num_items, getattr(args, "seed", 42),
                                             getattr(args, "bert_num_blocks", 2))
        self.load_state_dict(state_dict)

    # -- construction ------------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, state_dict, device="cuda:0"):
        return cls(state_dict=state_dict, device=device)

    @classmethod
    def from_checkpoint(cls, path, device="cuda:0"):
        """Reads `best_acc_model.pth` as written by the reference (trainer/base.py:326-330)."""
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        return cls(state_dict=ckpt.get("model_state_dict", ckpt), device=device)

    def load_state_dict(self, state_dict, strict=True):
        img, v, nb = pack_state_dict(state_dict)
        if not torch.cuda.is_available():
            raise RuntimeError("LRURec needs a GPU (MI355X); there is no CPU fallback")
        self._release()
        self._sd = state_dict
        self.num_items, self.num_blocks = v, nb
        with torch.cuda.device(self.device):
            self._img = torch.from_numpy(img).to(self.device)
            torch.cuda.synchronize()
            h = C.c_void_p()
            check(lib().lr_lru_create(self._img.data_ptr(), self._img.numel() * 4, v, nb, C.byref(h)),
                  "lr_lru_create")
        self._h = h
        return self

    def state_dict(self):
        return self._sd

    def eval(self):
        self.training = False
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            self.device = torch.device(device)
            self.load_state_dict(self._sd)
        return self

    def _release(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().lr_lru_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # -- helpers -----------------------------------------------------------------------------
    def _workspace(self, B, K, L=1):
        need = lib().lr_lru_workspace_bytes(self._h, B, K, L)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _ids(self, x):
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x))
        if x.dim() != 2:
            raise ValueError(f"expected ids of shape [B, L], got {tuple(x.shape)}")
        return x.to(device=self.device, dtype=torch.int64).contiguous()

    # -- scoring API -------------------------------------------------------------------------
    def encode_last(self, x):
        ids = self._ids(x)
        B, L = ids.shape
        q = torch.empty((B, 64), dtype=torch.float32, device=self.device)
        if B == 0:
            return q
        ws = self._workspace(B, 1, L)
        with torch.cuda.device(self.device):
            check(lib().lr_lru_encode_last(self._h, ids.data_ptr(), B, L, q.data_ptr(), ws.data_ptr(), ws.numel(),
                                           stream_ptr()), "lr_lru_encode_last")
        return q

    def scores_last(self, x, exclude_history=False):
        ids = self._ids(x)
        B, L = ids.shape
        out = torch.empty((B, self.num_items + 1), dtype=torch.float32, device=self.device)
        if B == 0:
            return out
        ws = self._workspace(B, 1, L)
        with torch.cuda.device(self.device):
            check(lib().lr_lru_scores_last(self._h, ids.data_ptr(), B, L, int(bool(exclude_history)),
                                           out.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr()),
                  "lr_lru_scores_last")
        return out

    def retrieve_topk(self, x, k, exclude_history=True):
        """Ordered top-k item ids (score desc, ties -> lower id) and their scores.
        Fuses `model(seqs)[:, -1, :]`, the -1e9 masking loop and torch.topk / argsort
        (trainer/lru.py:33-38,82-84,113-115)."""
        if not 1 <= k <= MAX_TOPK:
            raise ValueError(f"k must be in 1..{MAX_TOPK}")
        ids = self._ids(x)
        B, L = ids.shape
        idx = torch.empty((B, k), dtype=torch.int32, device=self.device)
        sc = torch.empty((B, k), dtype=torch.float32, device=self.device)
        if B == 0:
            return idx, sc
        ws = self._workspace(B, k, L)
        with torch.cuda.device(self.device):
            check(lib().lr_lru_retrieve_topk(self._h, ids.data_ptr(), B, L, k, int(bool(exclude_history)),
                                             idx.data_ptr(), sc.data_ptr(), ws.data_ptr(), ws.numel(),
                                             stream_ptr()), "lr_lru_retrieve_topk")
        return idx, sc

    def set_encoder_pipeline(self, enable):
        """Diagnostic (lr_lru_set_encoder_pipeline): the encoder's LRU layer on the pipelined kernel (default) or on the
        one-tile-at-a-time kernel; same bits."""
        check(lib().lr_lru_set_encoder_pipeline(self._h, int(bool(enable))), "lr_lru_set_encoder_pipeline")

    def last_topk_path(self, B, L, k, exclude_history=True):
        """Diagnostic (lr_lru_topk_path): which path the last retrieve_topk call of exactly this shape took -- 0 exact full
        pass, 1 bound -> candidates -> rescoring, 2 that path overflowed and the exact pass redid the call."""
        import ctypes as C

        ws = self._workspace(B, k, L)
        out = C.c_int32(-1)
        with torch.cuda.device(self.device):
            check(lib().lr_lru_topk_path(self._h, B, L, k, int(bool(exclude_history)), ws.data_ptr(), ws.numel(), C.byref(out),
                                         stream_ptr()), "lr_lru_topk_path")
        return int(out.value)

    def forward(self, x):
        """Reference-compatible call: returns fp32 [B, 1, V+1] holding the LAST position's scores,
        so the reference idiom `model(seqs)[:, -1, :]` (trainer/lru.py:33,67,105,
        demo/inference.py:48) is unchanged. Other positions are never consumed on the scoring path
        and are not computed (the reference spends ~85% of its time on them)."""
        return self.scores_last(x, exclude_history=False).unsqueeze(1)

    __call__ = forward
