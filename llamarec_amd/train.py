"""Retriever training on MI355X -- host-side mirror of the reference's training seam, backed by the HIP
kernels of csrc/lru_train.hip (no torch autograd, no CPU fallback).

Reference interface being replaced (paths into the reference tree):
  LRUTrainer.calculate_loss(batch)                  trainer/lru.py:20-28   CE over all positions, ignore_index=0
  BaseTrainer.train_one_epoch: zero_grad, backward,  trainer/base.py:84-132
      clip_gradients(max_grad_norm), optimizer.step, lr_scheduler.step
  BaseTrainer._create_optimizer (AdamW, two groups)   trainer/base.py:219-246
  checkpoints: {"model_state_dict": ...} in models/best_acc_model.pth     trainer/base.py:326-330, config.py:7
Batches are what LRUTrainDataset yields (dataloader/lru.py:119-131): int64 tokens / labels [B, L], left-padded
with 0; label 0 = ignored.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _abi as A
from ._lib import check, lib, stream_ptr

_COMPLEX = ("in_proj.weight", "in_proj.bias", "out_proj.weight", "out_proj.bias")


class LRUTrainEngine:
    """Parameters, gradients and AdamW state of one LRURec on one GPU.

    >>> eng = LRUTrainEngine(init_lru_state_dict(num_items), lr=1e-3)
    >>> loss = eng.train_step(tokens, labels)          # forward + backward + clip + AdamW
    >>> torch.save({"model_state_dict": eng.state_dict()}, "best_acc_model.pth")   # reference format
    """

    def __init__(self, state_dict, lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-9, max_grad_norm=5.0,
                 dropout=0.2, attn_dropout=0.2, seed=42, device="cuda:0", use_graph=False, ce_mode=0):
        if not torch.cuda.is_available():
            raise RuntimeError("LRUTrainEngine needs a GPU (MI355X); there is no CPU fallback")
        self.device = torch.device(device)
        self.lr = float(lr)
        sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in state_dict.items()}
        # logical shapes; complex tensors may arrive as complex64 [...] or as float pairs [..., 2]
        self._shapes = {k: (tuple(v.shape[:-1]) if (any(k.endswith(c) for c in _COMPLEX) and not np.iscomplexobj(v))
                            else tuple(v.shape)) for k, v in sd.items()}
        desc, keep = A.lru_desc_from_state_dict(sd)
        self.num_items, self.num_blocks = int(desc.num_items), int(desc.num_blocks)
        cfg = A.LrLruTrainConfig(weight_decay=weight_decay, beta1=betas[0], beta2=betas[1], eps=eps,
                                 max_grad_norm=max_grad_norm, dropout=dropout, attn_dropout=attn_dropout, seed=seed,
                                 ce_mode=ce_mode)
        nbytes = lib().lr_lru_train_state_bytes(self.num_items, self.num_blocks)
        if nbytes == 0:
            raise ValueError("unsupported LRURec shape")
        with torch.cuda.device(self.device):
            self._state = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            torch.cuda.synchronize()
            h = C.c_void_p()
            check(lib().lr_lru_train_create(C.byref(desc), C.byref(cfg), self._state.data_ptr(), nbytes, C.byref(h)),
                  "lr_lru_train_create")
        del keep
        self._h = h
        p, g, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        check(lib().lr_lru_train_buffers(self._h, C.byref(p), C.byref(g), C.byref(n)), "lr_lru_train_buffers")
        self._n = int(n.value)
        base = self._state.data_ptr()
        f32 = self._state.view(torch.float32)
        self.params = f32[(p.value - base) // 4:(p.value - base) // 4 + self._n]   # views into the state buffer
        self.grads = f32[(g.value - base) // 4:(g.value - base) // 4 + self._n]
        self._ws = None
        self._out = torch.zeros(4, dtype=torch.float32, device=self.device)  # loss, n_valid, n_bad_labels, grad norm
        # the step runs on its own stream (a hipGraph cannot be captured on the default stream) from persistent
        # token / label buffers, so every step after the first is one graph replay per half
        self._stream = torch.cuda.Stream(device=self.device)
        self._tok = self._lab = None
        check(lib().lr_lru_train_set_graph(self._h, int(bool(use_graph))), "lr_lru_train_set_graph")

    def set_fused(self, enable):
        """Row-panel kernels for the LRU blocks (default) or one generic GEMM launch per product (the cross-check)."""
        check(lib().lr_lru_train_set_fused(self._h, int(bool(enable))), "lr_lru_train_set_fused")

    def set_deterministic(self, enable=True):
        """Run-to-run identical bits (include/llamarec_mi355x.h, lr_lru_train_set_deterministic): the pass's fp32 atomics become
        64-bit fixed-point adds into shadow buffers that are folded back at fixed points. Needs the row-panel kernels (the
        default) and a larger workspace (re-allocated on the next step). One deterministic engine per process at a time."""
        check(lib().lr_lru_train_set_deterministic(self._h, int(bool(enable))), "lr_lru_train_set_deterministic")
        self._tok = self._lab = self._ws = None
        return self

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().lr_lru_train_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # -- one step ----------------------------------------------------------------------------
    def _batch(self, tokens, labels):
        t = torch.as_tensor(np.asarray(tokens) if not isinstance(tokens, torch.Tensor) else tokens)
        l = torch.as_tensor(np.asarray(labels) if not isinstance(labels, torch.Tensor) else labels)
        if t.dim() != 2 or t.shape != l.shape:
            raise ValueError(f"tokens {tuple(t.shape)} and labels {tuple(l.shape)} must both be [B, L]")
        return (t.to(device=self.device, dtype=torch.int64).contiguous(),
                l.to(device=self.device, dtype=torch.int64).contiguous())

    def _on_stream(self):
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        return cur

    def loss_and_grads(self, tokens, labels):
        """trainer/lru.py:20-28 + loss.backward(): fills `self.grads`, returns the loss as a 0-dim device tensor."""
        t, l = self._batch(tokens, labels)
        B, L = t.shape
        if self._tok is None or tuple(self._tok.shape) != (B, L):
            self._tok, self._lab = torch.empty_like(t), torch.empty_like(l)
            self._ws = torch.empty(lib().lr_lru_train_workspace_bytes(self._h, B, L), dtype=torch.uint8, device=self.device)
        cur = self._on_stream()
        with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
            self._tok.copy_(t, non_blocking=True)
            self._lab.copy_(l, non_blocking=True)
            check(lib().lr_lru_train_loss_grad(self._h, self._tok.data_ptr(), self._lab.data_ptr(), B, L,
                                               self._out.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                               self._stream.cuda_stream), "lr_lru_train_loss_grad")
        cur.wait_stream(self._stream)
        return self._out[0]

    @property
    def bad_labels(self):
        """Labels outside [0, num_items] seen by the last loss_and_grads (they were ignored)."""
        return int(self._out[2])

    def apply(self, lr=None, max_grad_norm=None):
        """clip_gradients(limit) + optimizer.step (trainer/base.py:109-110,201-202); returns the pre-clip gradient
        norm (0-dim device tensor)."""
        cur = self._on_stream()
        with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
            check(lib().lr_lru_train_apply(self._h, float(self.lr if lr is None else lr),
                                           float(0.0 if max_grad_norm is None else max_grad_norm),
                                           self._out[3:].data_ptr(), self._stream.cuda_stream), "lr_lru_train_apply")
        cur.wait_stream(self._stream)
        return self._out[3]

    def train_step(self, tokens, labels, lr=None, all_reduce=None):
        """One optimizer step. all_reduce: optional callable(tensor) that averages the flat gradient buffer across
        data-parallel ranks (llamarec_amd.dist.average_)."""
        loss = self.loss_and_grads(tokens, labels)
        if all_reduce is not None:
            all_reduce(self.grads)
        self.apply(lr)
        return loss

    # -- parameters by their reference names -------------------------------------------------
    def _range(self, name):
        off, cnt = C.c_size_t(), C.c_size_t()
        check(lib().lr_lru_train_param_range(self._h, name.encode(), C.byref(off), C.byref(cnt)), "lr_lru_train_param_range")
        return int(off.value), int(cnt.value)

    def _named(self, flat):
        out = {}
        for name, cshape in self._shapes.items():
            off, cnt = self._range(name)
            a = flat[off:off + cnt].detach().cpu().numpy().copy()
            if any(name.endswith(c) for c in _COMPLEX):
                out[name] = a.view(np.complex64).reshape(cshape)
            else:
                out[name] = a.reshape(cshape)
        return out

    def state_dict(self):
        """Reference-format state_dict (numpy; complex64 where the reference's tensors are complex): feeds
        LRURec.from_state_dict and `torch.save({"model_state_dict": ...})`."""
        return self._named(self.params)

    def grad_dict(self):
        return self._named(self.grads)


def average_gradients_(flat: torch.Tensor) -> torch.Tensor:
    """Data-parallel gradient exchange: ONE all-reduce of the flat gradient buffer (RCCL over xGMI with backend
    "nccl", gloo in the CPU tests), then the mean -- what DDP does bucket by bucket for the reference. No-op for a
    single process."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():   # a 1-rank group too (rehearsal of the collective on one GPU)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if dist.get_world_size() > 1:
            flat.mul_(1.0 / dist.get_world_size())
    return flat


def lr_lambda_from_args(args, num_training_steps=None):
    """The reference's optional schedules (trainer/base.py:45-57,248-260): linear warm-up + linear decay, or StepLR."""
    if not getattr(args, "enable_lr_schedule", False):
        return None
    if getattr(args, "enable_lr_warmup", False):
        w, n = args.warmup_steps, max(int(num_training_steps or 0), args.warmup_steps + 1)
        return lambda it: it / max(1, w) if it < w else max(0.0, (n - it) / max(1, n - w))
    step = args.decay_step or 10000   # the reference leaves --decay_step unset (config.py); StepLR needs a period
    return lambda it: args.gamma ** (it // step)


class LRUTrainer:
    """Mirror of the reference's LRUTrainer training surface (trainer/lru.py:13-28, trainer/base.py:60-154,
    trainer/loggers.py:102-125): `calculate_loss(batch)`, `train()` with validation every `val_iterations`
    optimizer steps (or every epoch), best-`best_metric` checkpointing to models/best_acc_model.pth and early
    stopping after `early_stopping_patience` validations without improvement. Validation itself is the scoring
    path (llamarec_amd.retrieve.LRUEvaluator on the exported weights, history not masked: trainer/base.py:141-143)."""

    def __init__(self, args, state_dict=None, device="cuda:0", export_root=None, rank=0, world=1):
        from .lru import init_lru_state_dict

        self.args = args
        if state_dict is None:
            state_dict = init_lru_state_dict(args.num_items, getattr(args, "seed", 42), getattr(args, "bert_num_blocks", 2))
        self.engine = LRUTrainEngine(
            state_dict, lr=getattr(args, "lr", 1e-3), weight_decay=getattr(args, "weight_decay", 1e-2),
            eps=getattr(args, "adam_epsilon", 1e-9), max_grad_norm=getattr(args, "max_grad_norm", 5.0),
            dropout=getattr(args, "bert_dropout", 0.2), attn_dropout=getattr(args, "bert_attn_dropout", 0.2),
            seed=getattr(args, "seed", 42) + 7919 * rank, device=device)
        if getattr(args, "deterministic", False):   # --deterministic: run-to-run identical bits (LRUTrainEngine.set_deterministic)
            self.engine.set_deterministic(True)
        self.device = device
        self.export_root = export_root
        self.rank, self.world = rank, world
        self.iterations = 0
        self.best_metric, self.patience_counter = 0.0, 0
        self.history = []

    def calculate_loss(self, batch):
        seqs, labels = batch
        return self.engine.loss_and_grads(seqs, labels)

    def train_one_epoch(self, batches, lr_lambda=None, validate=None):
        """batches: iterable of (tokens, labels). Returns (mean loss, stop flag) -- trainer/base.py:98-132."""
        total, n = 0.0, 0
        reduce_ = average_gradients_ if self.world > 1 else None
        max_it = getattr(self.args, "max_train_iterations", None)
        for tokens, labels in batches:
            lr = self.engine.lr * (lr_lambda(self.iterations) if lr_lambda else 1.0)
            loss = self.engine.train_step(tokens, labels, lr=lr, all_reduce=reduce_)
            total += float(loss)
            n += 1
            self.iterations += 1
            if validate and getattr(self.args, "val_strategy", "iteration") == "iteration" and \
                    self.iterations % self.args.val_iterations == 0 and validate():
                return total / max(n, 1), True
            if max_it and self.iterations >= max_it:
                return total / max(n, 1), True
        return total / max(n, 1), False

    def _log_val(self, metrics):
        """BestModelLogger.log + LoggerService.log_val (trainer/loggers.py:26-35,116-125): True = stop."""
        cur = metrics.get(getattr(self.args, "best_metric", "Recall@10"), 0.0)
        self.history.append(dict(iteration=self.iterations, **metrics))
        if self.best_metric < cur:
            self.best_metric, self.patience_counter = cur, 0
            if self.export_root and self.rank == 0:
                import os

                os.makedirs(os.path.join(self.export_root, "models"), exist_ok=True)
                sd = {k: torch.from_numpy(v) for k, v in self.engine.state_dict().items()}
                torch.save({"model_state_dict": sd}, os.path.join(self.export_root, "models", "best_acc_model.pth"))
        else:
            self.patience_counter += 1
        return self.patience_counter >= getattr(self.args, "early_stopping_patience", 20)

    def validate(self, val_loader):
        from .lru import LRURec
        from .retrieve import LRUEvaluator

        model = LRURec.from_state_dict(self.engine.state_dict(), device=self.device)
        metrics = LRUEvaluator(self.args, model, val_loader, []).validate()
        return self._log_val(metrics)

    def train(self, all_seqs, val_loader, rng=None):
        """trainer/base.py:60-82: validate, then epochs of shuffled batches until early stopping / num_epochs."""
        from . import data as D

        rng = rng or np.random.default_rng(getattr(self.args, "seed", 42))
        L, bs = self.args.bert_max_len, self.args.train_batch_size
        steps_per_epoch = (len(all_seqs) + bs * self.world - 1) // (bs * self.world)
        lr_lambda = lr_lambda_from_args(self.args, steps_per_epoch * self.args.num_epochs)
        val = (lambda: self.validate(val_loader)) if val_loader is not None else None
        stop = val() if val else False
        losses = []
        for _epoch in range(self.args.num_epochs):
            if stop:
                break
            loss, stop = self.train_one_epoch(D.train_batches(all_seqs, bs, L, rng, self.rank, self.world), lr_lambda, val)
            losses.append(loss)
            if not stop and val and self.args.val_strategy == "epoch":
                stop = val()
        return losses

    def state_dict(self):
        return self.engine.state_dict()
