"""Ranker LoRA fine-tuning -- host-side mirror of the reference's `LLMTrainer.train()` path
(trainer/llm.py:76-136 over the patched LlamaForCausalLM of model/llm.py:89-127 with peft LoRA on
q_proj / v_proj, train_ranker.py:71-79), backed by the HIP training step of
`csrc/api_llama_train.hip` (no torch compute, no fallback).

  LoraTrainEngine   one replica: frozen bf16 base (a LlamaRanker built WITHOUT a merged adapter), fp32 LoRA
                    parameters / gradients / Adam moments in one flat buffer each, loss_and_grads / apply / scores
  lora_samples      LLMTrainDataset.__getitem__ (dataloader/llm.py:236-283): 19 sampled negatives + shuffle, prompt with
                    the answer letter appended, labels[:-2] = -100
  LoraRankerTrainer the Trainer loop: gradient accumulation (train_batch_size / lora_micro_batch_size), linear warm-up
                    and decay, clipping at 1.0, validation every lora_val_iterations with best-adapter checkpoint and
                    early stopping on rerank_best_metric, adapter saved in PEFT's on-disk format
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os

import numpy as np
import torch

from . import _abi as A
from ._lib import check, lib, stream_ptr
from .llm import LlamaRanker, pack_prompts

IGNORE = -100
PEFT_KEY = "base_model.model.model.layers.{l}.self_attn.{p}_proj.lora_{ab}.weight"


def loss_rows_and_targets(seqs, labels):
    """Shift of model/llm.py:118-120 on packed prompts: row p of a prompt predicts token p+1, so the rows that enter the
    loss are those whose NEXT token carries a label. Returns (rows int32 [m], targets int32 [m])."""
    rows, tgts, base = [], [], 0
    for s, lab in zip(seqs, labels):
        lab = np.asarray(lab)
        if len(lab) != len(s):
            raise ValueError("labels and input_ids differ in length")
        p = np.nonzero(lab[1:] != IGNORE)[0]
        rows.append(base + p)
        tgts.append(lab[1:][p])
        base += len(s)
    return np.concatenate(rows).astype(np.int32), np.concatenate(tgts).astype(np.int32)


class LoraTrainEngine:
    def __init__(self, ranker: LlamaRanker, r=8, alpha=32, dropout=0.05, seed=42, beta1=0.9, beta2=0.999, eps=1e-8,
                 weight_decay=0.0, init=None):
        self.ranker, self.device = ranker, ranker.device
        self.r, self.alpha = int(r), float(alpha)
        c = ranker.config
        self.L = c["num_hidden_layers"]
        T, L_ = ranker._tensors, lib()
        self._t = {}
        with torch.cuda.device(self.device):
            def transposed(w):
                out = torch.empty((w.shape[1], w.shape[0]), dtype=torch.bfloat16, device=self.device)
                check(L_.lr_transpose_bf16(w.data_ptr(), w.shape[0], w.shape[1], out.data_ptr(), stream_ptr()),
                      "lr_transpose_bf16")
                return out

            arr = (A.LrLlamaLayerWeightsT * self.L)()
            for i in range(self.L):
                for f in ("wqkv", "wo", "wgu", "wdown"):
                    self._t[f"{i}.{f}_t"] = transposed(T[f"{i}.{f}"])
                    setattr(arr[i], f + "_t", self._t[f"{i}.{f}_t"].data_ptr())
            self._t["lm_head_t"] = transposed(T["lm_head"])
            desc = A.LrLlamaWeightsTDesc(layers=arr, lm_head_t=self._t["lm_head_t"].data_ptr())
            cfg = A.LrLoraTrainConfig(r=self.r, alpha=self.alpha, dropout=float(dropout), beta1=beta1, beta2=beta2,
                                      eps=eps, weight_decay=weight_decay, seed=int(seed))
            nbytes = L_.lr_llama_lora_state_bytes(ranker._h, C.byref(cfg))
            if nbytes == 0:
                check(1, "lr_llama_lora_state_bytes")
            self._state = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            h = C.c_void_p()
            check(L_.lr_llama_lora_create(ranker._h, C.byref(desc), C.byref(cfg), self._state.data_ptr(), nbytes,
                                          stream_ptr(), C.byref(h)), "lr_llama_lora_create")
            self._h, self._arr = h, arr
            ptrs = [C.c_void_p() for _ in range(4)]
            n = C.c_size_t()
            check(L_.lr_llama_lora_buffers(h, *[C.byref(p) for p in ptrs], C.byref(n)), "lr_llama_lora_buffers")
            base = self._state.data_ptr()

            def view(p):
                off = p.value - base
                return self._state[off:off + 4 * n.value].view(torch.float32)

            self.params, self.grads, self.m, self.v = (view(p) for p in ptrs)
        self._ws = self._eval_ws = None
        self._out = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.load(init if init is not None else self.peft_init(seed))

    # -- parameters -------------------------------------------------------------------------------------
    def _range(self, layer, which, ab):
        off, cnt = C.c_size_t(), C.c_size_t()
        check(lib().lr_llama_lora_param_range(self._h, layer, which, ab, C.byref(off), C.byref(cnt)),
              "lr_llama_lora_param_range")
        return off.value, cnt.value

    def shapes(self):
        c = self.ranker.config
        d, hd = c["hidden_size"], self.ranker.hd
        return {("q", "A"): (self.r, d), ("q", "B"): (c["num_attention_heads"] * hd, self.r),
                ("v", "A"): (self.r, d), ("v", "B"): (c["num_key_value_heads"] * hd, self.r)}

    def named(self, buf=None):
        """{"layers.{l}.{q,v}_proj.lora_{A,B}": view into the flat buffer (default: parameters)}."""
        buf = self.params if buf is None else buf
        out, sh = {}, self.shapes()
        for l in range(self.L):
            for wi, p in enumerate("qv"):
                for ai, ab in enumerate("AB"):
                    off, cnt = self._range(l, wi, ai)
                    out[f"layers.{l}.{p}_proj.lora_{ab}"] = buf[off:off + cnt].view(sh[(p, ab)])
        return out

    def peft_init(self, seed):
        """peft's default: lora_A ~ kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)), lora_B = 0."""
        g = torch.Generator().manual_seed(int(seed))
        out = {}
        for (p, ab), shape in self.shapes().items():
            for l in range(self.L):
                if ab == "A":
                    bound = 1.0 / math.sqrt(shape[1])
                    out[f"layers.{l}.{p}_proj.lora_A"] = (torch.rand(shape, generator=g) * 2 - 1) * bound
                else:
                    out[f"layers.{l}.{p}_proj.lora_B"] = torch.zeros(shape)
        return out

    def load(self, weights):
        views = self.named()
        for k, v in views.items():
            src = weights[k]
            src = torch.from_numpy(np.ascontiguousarray(src)) if not isinstance(src, torch.Tensor) else src
            v.copy_(src.to(torch.float32).reshape(v.shape))
        return self

    def export(self):
        """PEFT-named CPU tensors (adapter_model.safetensors keys)."""
        out = {}
        for k, v in self.named().items():
            parts = k.split(".")   # layers.{l}.{q,v}_proj.lora_{A,B}
            out[PEFT_KEY.format(l=parts[1], p=parts[2][0], ab=parts[3][-1])] = v.detach().cpu().clone()
        return out

    def save_adapter(self, path, base_model=""):
        """adapter_config.json + adapter_model.safetensors, loadable by peft and by LlamaRanker.from_pretrained."""
        from safetensors.torch import save_file

        os.makedirs(path, exist_ok=True)
        json.dump({"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": self.r, "lora_alpha": self.alpha,
                   "lora_dropout": 0.0, "bias": "none", "target_modules": ["q_proj", "v_proj"],
                   "base_model_name_or_path": base_model, "fan_in_fan_out": False, "inference_mode": True},
                  open(os.path.join(path, "adapter_config.json"), "w"), indent=1)
        save_file(self.export(), os.path.join(path, "adapter_model.safetensors"))

    # -- the step ---------------------------------------------------------------------------------------
    def reserve(self, shapes):
        """Size the training workspace ONCE for the largest of `shapes` = [(tokens, prompts, labelled rows), ...]:
        a workspace is ~100 KB per token and layer (tens of GB for Llama-2-7b), so growing it when a larger
        micro-batch arrives means a hipFree + hipMalloc of that size in the middle of the step loop (round 2's
        driver line: 628 ms per step instead of 290)."""
        need = max(int(lib().lr_llama_lora_workspace_bytes(self._h, int(n), int(B), int(m))) for n, B, m in shapes)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self.ws_allocations += 1
        return self._ws

    ws_allocations = 0   # how often the workspace was (re)allocated: 1 after a reserve() that covered the run

    def _workspace(self, n, B, m):
        return self.reserve([(n, B, m)])

    merged = False

    def loss_and_grads(self, seqs, labels, grad_scale=1.0, accumulate=False):
        """One micro-batch. Returns the loss as a device scalar (mean over the labelled tokens); gradients * grad_scale
        are written to (or, with accumulate, added to) `self.grads`."""
        if self.merged:
            raise RuntimeError("the adapter was merged into the base weights: build a new engine to train further")
        ids, cu = pack_prompts(seqs)
        rows, tgts = loss_rows_and_targets(seqs, labels)
        if len(rows) == 0:
            raise ValueError("no labelled token in the micro-batch")
        dev = self.device
        ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
        rows_d, tgts_d = torch.from_numpy(rows).to(dev), torch.from_numpy(tgts).to(dev)
        B, n, m = len(seqs), int(cu[-1]), len(rows)
        ws = self._workspace(n, B, m)
        with torch.cuda.device(dev):
            check(lib().lr_llama_lora_loss_grad(self._h, ids_d.data_ptr(), cu_d.data_ptr(), cu.ctypes.data, B,
                                                rows_d.data_ptr(), tgts_d.data_ptr(), m, float(grad_scale),
                                                int(bool(accumulate)), self._out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                stream_ptr()), "lr_llama_lora_loss_grad")
        self._keep = (ids_d, cu_d, rows_d, tgts_d)
        return self._out[0]

    @property
    def bad_targets(self):
        return int(self._out[2].item())

    def apply(self, lr, max_grad_norm=1.0):
        """clip_grad_norm_ + AdamW; returns the gradient norm before clipping (device scalar)."""
        with torch.cuda.device(self.device):
            check(lib().lr_llama_lora_apply(self._h, float(lr), float(max_grad_norm), self._out[3:].data_ptr(),
                                            stream_ptr()), "lr_llama_lora_apply")
        return self._out[3]

    def scores(self, seqs, label_token_ids):
        """fp32 [B, C] verbalizer scores with the adapters as they are now (no dropout)."""
        ids, cu = pack_prompts(seqs)
        dev = self.device
        ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
        lab = torch.as_tensor(np.asarray(label_token_ids, dtype=np.int32)).to(dev)
        B, n = len(seqs), int(cu[-1])
        need = lib().lr_llama_lora_eval_workspace_bytes(self._h, n, B)
        if self._eval_ws is None or self._eval_ws.numel() < need:
            self._eval_ws = None
            self._eval_ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((B, lab.numel()), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(lib().lr_llama_lora_prefill_verbalize(self._h, ids_d.data_ptr(), cu_d.data_ptr(), cu.ctypes.data, B,
                                                        lab.data_ptr(), lab.numel(), out.data_ptr(),
                                                        self._eval_ws.data_ptr(), self._eval_ws.numel(), stream_ptr()),
                  "lr_llama_lora_prefill_verbalize")
        return out

    prefill_verbalize = scores   # the evaluator's call (llamarec_amd.rerank.LLMEvaluator)

    def merge_into_base_(self):
        """W_q, W_v += (alpha/r) B A in the ranker's packed bf16 weights (what LlamaRanker.from_state_dict(lora=...) does
        at load) so that the plain scoring path serves the tuned model. The engine must not be trained further: its
        transposed copies and the now-merged base no longer describe 'base + adapter'."""
        rk, s = self.ranker, self.alpha / self.r
        c = rk.config
        nq, nkv = c["num_attention_heads"] * rk.hd, c["num_key_value_heads"] * rk.hd
        p = self.named()
        for l in range(self.L):
            w = rk._tensors[f"{l}.wqkv"]
            dq = rk._interleave_rope_rows(s * (p[f"layers.{l}.q_proj.lora_B"] @ p[f"layers.{l}.q_proj.lora_A"]))
            dv = s * (p[f"layers.{l}.v_proj.lora_B"] @ p[f"layers.{l}.v_proj.lora_A"])
            w[:nq] = (w[:nq].float() + dq).to(torch.bfloat16)
            w[nq + nkv:] = (w[nq + nkv:].float() + dv).to(torch.bfloat16)
        if rk.fold_norms:
            rk.set_fold_norms(True)   # the scoring path's wqkv * diag(input_norm) copies follow the merged weights
        self.merged = True
        return rk

    def __del__(self):
        try:
            if self._h.value:
                lib().lr_llama_lora_destroy(self._h)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# training samples
# ---------------------------------------------------------------------------------------------
class LLMTrainSamples:
    """LLMTrainDataset (dataloader/llm.py:236-283): one sample per prefix seq[:i], i >= 2, of every user's training
    sequence (users in sorted order); the answer is the prefix's last item, the history the llm_max_history items
    before it; llm_negative_sample_size negatives drawn from 5x as many uniform item ids (skipping the user's history
    and the answer), candidates shuffled. `rng` needs numpy's legacy `randint` / `shuffle` (the reference passes the
    `np.random` module; a `np.random.RandomState` reproduces it under a seed)."""

    def __init__(self, args, u2seq, text_dict, tokenizer, prompter=None, rng=None):
        from . import prompt as P

        self.P, self.args = P, args
        self.max_len = getattr(args, "llm_max_history", P.LLM_MAX_HISTORY)
        self.num_items = args.num_items
        self.neg = getattr(args, "llm_negative_sample_size", 19)
        self.rng = rng if rng is not None else np.random
        self.text_dict, self.tokenizer = text_dict, tokenizer
        self.prompter = prompter or P.Prompter()
        self.kw = dict(max_title_len=getattr(args, "llm_max_title_len", P.LLM_MAX_TITLE_LEN),
                       max_text_len=getattr(args, "llm_max_text_len", P.LLM_MAX_TEXT_LEN),
                       system_template=getattr(args, "llm_system_template", None) or P.DEFAULT_SYSTEM_TEMPLATE,
                       input_template=getattr(args, "llm_input_template", None) or P.DEFAULT_INPUT_TEMPLATE,
                       train_on_inputs=bool(getattr(args, "llm_train_on_inputs", False)))
        self.all_seqs = []
        for u in sorted(u2seq.keys()):
            seq = u2seq[u]
            for i in range(2, len(seq) + 1):
                self.all_seqs.append(seq[:i])

    def __len__(self):
        return len(self.all_seqs)

    def candidates(self, index):
        tokens = self.all_seqs[index]
        answer, original_seq = tokens[-1], tokens[:-1]
        seq = original_seq[-self.max_len:]
        cur, cands = 0, [answer]
        samples = self.rng.randint(1, self.num_items + 1, size=5 * self.neg)
        while len(cands) < self.neg + 1:
            item = samples[cur]
            cur += 1
            if item in original_seq or item == answer:
                continue
            cands.append(item)
        self.rng.shuffle(cands)
        return seq, cands, answer

    def __getitem__(self, index):
        seq, cands, answer = self.candidates(index)
        return self.P.seq_to_token_ids_train(seq, cands, answer, self.text_dict, self.tokenizer, self.prompter,
                                             **self.kw)


def linear_schedule(warmup_steps, total_steps):
    """transformers.get_linear_schedule_with_warmup (HF Trainer's default lr_scheduler_type="linear")."""
    def f(step):
        if step < warmup_steps:
            return step / max(1, warmup_steps)
        return max(0.0, (total_steps - step) / max(1, total_steps - warmup_steps))
    return f


# ---------------------------------------------------------------------------------------------
# the Trainer loop
# ---------------------------------------------------------------------------------------------
class LoraRankerTrainer:
    """Mirror of LLMTrainer(...).train() (trainer/llm.py:76-136, HF Trainer semantics for the arguments it sets):
    micro-batches of lora_micro_batch_size, train_batch_size // lora_micro_batch_size of them per optimizer step,
    AdamW at lora_lr with linear warm-up over warmup_steps and linear decay to 0, clipping at HF's default 1.0,
    lora_num_epochs epochs (or lora_max_steps), validation every lora_val_iterations steps on the retriever's
    validation candidates with `rerank_best_metric` deciding the kept adapter, EarlyStoppingCallback patience,
    load_best_model_at_end. Data parallel: rank r takes micro-batches r, r + world, ... of every step and the flat
    gradient buffer is averaged with ONE all-reduce before the optimizer step (DDP's semantics)."""

    MAX_GRAD_NORM = 1.0   # TrainingArguments default

    def __init__(self, args, engine: LoraTrainEngine, train_samples, val_items, verbalizer, export_root=None, rank=0,
                 world=1, log=print):
        self.args, self.engine, self.samples, self.val_items = args, engine, train_samples, val_items
        self.verbalizer, self.export_root, self.rank, self.world, self.log = verbalizer, export_root, rank, world, log
        self.micro = args.lora_micro_batch_size
        self.accum = max(1, args.train_batch_size // args.lora_micro_batch_size)
        per_step = self.micro * self.accum * world
        if len(train_samples) < per_step:
            # HF's dataloader would yield one short batch; this trainer only takes whole optimizer steps, and train()
            # would otherwise spin over empty epochs forever
            raise ValueError(f"{len(train_samples)} training samples are fewer than one optimizer step "
                             f"(micro batch {self.micro} x accumulation {self.accum} x world {world} = {per_step}): "
                             f"lower --train_batch_size / --lora_micro_batch_size or the number of ranks")
        steps_per_epoch = max(1, len(train_samples) // per_step)
        self.total_steps = args.lora_max_steps if args.lora_max_steps and args.lora_max_steps > 0 \
            else steps_per_epoch * args.lora_num_epochs
        self.schedule = linear_schedule(args.warmup_steps, self.total_steps)
        self.ks = list(getattr(args, "rerank_metric_ks", [1, 5, 10]))
        self.best_metric, self.best_state, self.bad_evals = None, None, 0
        self.history = []
        # tokens per forward/backward pass when an optimizer step's samples are regrouped (repack); None = the reference's
        # micro-batches as they come. 16 384 rows = 64 row tiles: whole 256-CU rounds for N = 4096 / 12288 (packing.py)
        self.train_token_budget = getattr(args, "lora_token_budget", 16384) or None

    def _order(self, epoch):
        return np.random.RandomState(self.args.seed + epoch).permutation(len(self.samples))

    def repack(self, micro):
        """An optimizer step's micro-batches [(seqs, labels), ...] of this rank -> [(seqs, labels, grad_scale), ...] with
        the SAME summed gradient in fewer, token-budget-sized passes. The reference's step gradient is
        (1 / accum) * sum_k mean over micro-batch k's labelled tokens (HF Trainer's accumulation over the loss of
        model/llm.py:116-127). When every micro-batch carries the same number m of labelled tokens (always so with
        train_on_inputs off: the answer letter + EOS per sample, dataloader/llm.py:53-58) each labelled token weighs
        1 / (accum * m) wherever it sits, so the samples may be regrouped freely: a pass over m_j labelled tokens runs
        with grad_scale m_j / (accum * m). The GEMMs then see one 11.8 k-row pass instead of two 5.9 k-row ones on Beauty
        (reference micro batch 8, config.py:96: 2.9 instead of 2 x 1.4 rounds of 256 x 256 tiles on N = 4096).
        Unequal label counts (train_on_inputs) keep the reference's grouping."""
        counts = [sum(int((np.asarray(l)[1:] != IGNORE).sum()) for l in labels) for _, labels in micro]
        if self.train_token_budget is None or len(micro) < 2 or len(set(counts)) != 1 or counts[0] == 0:
            return [(seqs, labels, 1.0 / len(micro)) for seqs, labels in micro]
        from .packing import token_budget_steps

        seqs = [s for sq, _ in micro for s in sq]
        labels = [l for _, lb in micro for l in lb]
        lens = [len(s) for s in seqs]
        total = float(sum(counts))
        out = []
        for idx in token_budget_steps(lens, max(self.train_token_budget, max(lens)), window=len(seqs)):
            lab = [labels[int(i)] for i in idx]
            mj = sum(int((np.asarray(l)[1:] != IGNORE).sum()) for l in lab)
            out.append(([seqs[int(i)] for i in idx], lab, mj / total))
        return out

    def evaluate(self):
        """Verbalizer scores with the live adapters -> Recall/MRR/NDCG@ks on the validation prompts (rank 0's shard of
        nothing: every rank evaluates the same items, cheap next to training)."""
        from . import metrics as M
        from . import prompt as P

        items = self.val_items
        n = getattr(self.args, "lora_max_val_samples", None)
        if n:
            items = items[:n]
        ncls = self.verbalizer.num_classes
        hist = torch.zeros(ncls + 1, dtype=torch.int64, device=self.engine.device)
        bs = getattr(self.args, "val_batch_size", None) or 16
        for i in range(0, len(items), bs):
            seqs, labels = P.eval_pack(items[i:i + bs], getattr(self.args, "llm_max_text_len", P.LLM_MAX_TEXT_LEN))
            scores = self.engine.scores(seqs, self.verbalizer.label_token_ids)
            M.rank_histogram(M.rank_classes(scores), torch.from_numpy(labels).to(self.engine.device), hist)
        return M.metrics_from_histogram(hist, self.ks) if len(items) else {}

    def _maybe_validate(self, step):
        a = self.args
        if not self.val_items or step % a.lora_val_iterations != 0 or step < a.lora_val_delay:
            return False
        m = self.evaluate()
        cur = m.get(a.rerank_best_metric, 0.0)
        self.history.append({"step": step, **{"eval_" + k: v for k, v in m.items()}})
        self.log(f"[step {step}] eval {a.rerank_best_metric} = {cur:.4f}")
        if self.best_metric is None or cur > self.best_metric:
            self.best_metric, self.bad_evals = cur, 0
            self.best_state = self.engine.params.clone()
            if self.export_root and self.rank == 0:
                self.engine.save_adapter(os.path.join(self.export_root, "best_adapter"),
                                         getattr(a, "llm_base_model", ""))
        else:
            self.bad_evals += 1
        return self.bad_evals >= a.lora_early_stopping_patience

    def train(self):
        from . import prompt as P
        from .train import average_gradients_

        a, eng = self.args, self.engine
        step, epoch, losses = 0, 0, []
        eos = getattr(getattr(self.samples, "tokenizer", None), "eos_token_id", 2)
        stop = False
        while step < self.total_steps and not stop:
            order = self._order(epoch)
            per_step = self.micro * self.accum * self.world
            for s0 in range(0, len(order) - per_step + 1, per_step):
                micro = []
                for k in range(self.accum):
                    lo = s0 + (k * self.world + self.rank) * self.micro
                    batch = [self.samples[int(i)] for i in order[lo:lo + self.micro]]
                    micro.append(P.train_pack(batch, getattr(a, "llm_max_text_len", P.LLM_MAX_TEXT_LEN), eos))
                step_loss = 0.0
                for j, (seqs, labels, scale) in enumerate(self.repack(micro)):
                    # the engine returns a view of its output scalar: scale-and-add makes a copy on the stream (no host
                    # sync); sum_j scale_j * loss_j = the mean of the reference's micro-batch losses
                    step_loss = step_loss + eng.loss_and_grads(seqs, labels, grad_scale=scale, accumulate=j > 0) * scale
                losses.append(step_loss)
                average_gradients_(eng.grads)
                eng.apply(a.lora_lr * self.schedule(step), self.MAX_GRAD_NORM)
                step += 1
                if step % 10 == 0 or step == self.total_steps:      # logging_steps=10 (trainer/llm.py:115)
                    self.log(f"[step {step}/{self.total_steps}] loss {float(torch.stack(losses).mean()):.4f} "
                             f"lr {a.lora_lr * self.schedule(step):.2e}")
                    losses = []
                if self._maybe_validate(step):
                    self.log(f"early stopping at step {step}")
                    stop = True
                if stop or step >= self.total_steps:
                    break
            epoch += 1
        if self.best_state is not None:                              # load_best_model_at_end=True
            eng.params.copy_(self.best_state)
        if self.export_root and self.rank == 0:
            eng.save_adapter(os.path.join(self.export_root, "adapter"), getattr(a, "llm_base_model", ""))
            json.dump(self.history, open(os.path.join(self.export_root, "lora_eval_history.json"), "w"), indent=1)
        return step
