"""GPU parity of the ranker's LoRA training step through the C ABI: attention backward against a
torch fp32 autograd reference of the same op, loss / LoRA gradients / AdamW steps against the goldens made by the
reference's training forward (tests/gen_goldens_rank_train.py) and against the float64 oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIGS = ["tiny_hd16", "tiny_hd128", "tiny_gqa"]


def _attention_ref(q, k, v, lens, scale):
    """q [n, nh, hd], k/v [n, nkv, hd] fp32 (requires_grad) -> o [n, nh, hd]; causal within each prompt."""
    outs, s0 = [], 0
    rep = q.shape[1] // k.shape[1]
    for T in lens:
        qs, ks, vs = q[s0:s0 + T], k[s0:s0 + T].repeat_interleave(rep, 1), v[s0:s0 + T].repeat_interleave(rep, 1)
        s = torch.einsum("qhd,khd->hqk", qs, ks) * scale
        s = s.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool, device=q.device))[None], float("-inf"))
        outs.append(torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vs))
        s0 += T
    return torch.cat(outs)


@pytest.mark.parametrize("nh,nkv,hd,variant,lens", [
    (2, 2, 128, 2, [200, 64, 1, 129, 77]),      # MFMA passes: ragged tails, a 1-token prompt, exact block multiples
    (4, 2, 128, 2, [130, 300]),                 # grouped-query: the dK/dV pass walks the heads of its group
    (4, 2, 32, 1, [37, 5, 64]),                 # generic kernel (test models)
    (2, 2, 128, 1, [70, 33]),                   # generic kernel on the MFMA shape
])
def test_attention_backward_matches_autograd(nh, nkv, hd, variant, lens):
    from llamarec_amd._lib import check, lib, stream_ptr

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    n = sum(lens)
    qw = (nh + 2 * nkv) * hd
    qkv = (torch.randn(n, qw, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    d_out = (torch.randn(n, nh * hd, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cu_d = torch.from_numpy(cu).to(dev)
    out = torch.empty(n, nh * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(n, nh, dtype=torch.float32, device=dev)
    dqkv = torch.full((n, qw), float("nan"), dtype=torch.bfloat16, device=dev)
    L = lib()
    check(L.lr_attention_varlen_lse(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), cu_d.data_ptr(), cu.ctypes.data,
                                    len(lens), nh, nkv, hd, variant, stream_ptr()), "lr_attention_varlen_lse")
    sb = L.lr_attention_bwd_scratch_bytes(n, nh, nkv, hd)
    scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
    check(L.lr_attention_varlen_bwd(qkv.data_ptr(), out.data_ptr(), d_out.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                    cu_d.data_ptr(), cu.ctypes.data, len(lens), nh, nkv, hd, variant,
                                    scratch.data_ptr(), sb, stream_ptr()), "lr_attention_varlen_bwd")
    torch.cuda.synchronize()
    x = qkv.float()
    q = x[:, :nh * hd].reshape(n, nh, hd).clone().requires_grad_(True)
    k = x[:, nh * hd:(nh + nkv) * hd].reshape(n, nkv, hd).clone().requires_grad_(True)
    v = x[:, (nh + nkv) * hd:].reshape(n, nkv, hd).clone().requires_grad_(True)
    o = _attention_ref(q, k, v, lens, 1.0 / np.sqrt(hd))
    # the statistics the backward consumes
    assert torch.allclose(out.float(), o.detach().reshape(n, -1), atol=2e-2, rtol=2e-2)
    o.backward(d_out.float().reshape(n, nh, hd))
    ref = torch.cat([q.grad.reshape(n, -1), k.grad.reshape(n, -1), v.grad.reshape(n, -1)], 1)
    got = dqkv.float()
    assert torch.isfinite(got).all()
    for name, a, b in (("dq", 0, nh * hd), ("dk", nh * hd, (nh + nkv) * hd), ("dv", (nh + nkv) * hd, qw)):
        err = (got[:, a:b] - ref[:, a:b]).abs().max().item()
        scale = ref[:, a:b].abs().max().item()
        assert err <= 2e-2 * scale, (name, err, scale)   # bf16 P / dS operands and bf16 outputs


def _load(golden_dir, name):
    from llamarec_amd.synth import synth_llama_state

    z = np.load(os.path.join(golden_dir, f"llama_lora_train_{name}.npz"))
    cfg = json.loads(str(z["config"]))
    sd = synth_llama_state(cfg, int(z["weight_seed"]))
    names = [str(n) for n in z["param_names"]]
    return z, cfg, sd, names


def _unpack(z, step):
    lens = z[f"step{step}/lens"]
    ids, lab = z[f"step{step}/packed_ids"], z[f"step{step}/packed_labels"]
    cu = np.concatenate([[0], np.cumsum(lens)])
    return ([ids[cu[i]:cu[i + 1]] for i in range(len(lens))], [lab[cu[i]:cu[i + 1]] for i in range(len(lens))])


def _engine(z, cfg, sd, names, **kw):
    from llamarec_amd.llm import LlamaRanker
    from llamarec_amd.rank_train import LoraTrainEngine

    ranker = LlamaRanker.from_state_dict(sd, cfg)
    init = {n: z["init/" + n] for n in names}
    kw.setdefault("dropout", 0.0)
    return LoraTrainEngine(ranker, r=int(z["lora_r"]), alpha=float(z["lora_alpha"]), init=init, **kw)


def _rel(a, b):
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("name", CONFIGS)
def test_loss_and_lora_gradients_match_reference(golden_dir, name):
    """bf16 HIP step vs the reference's fp32 run. The reference's own bf16-autocast run differs from its fp32 run by
    0.6 % (gradient, relative L2, stored in the goldens): the bar here is 3 % per tensor and 1.5 % overall."""
    z, cfg, sd, names = _load(golden_dir, name)
    eng = _engine(z, cfg, sd, names)
    seqs, labels = _unpack(z, 0)
    loss = float(eng.loss_and_grads(seqs, labels))
    assert eng.bad_targets == 0
    assert abs(loss - float(z["step0/loss"])) < 4e-3
    got = {k: v.detach().cpu().numpy() for k, v in eng.named(eng.grads).items()}
    for n in names:
        assert _rel(got[n], z["step0/grad/" + n]) < 3e-2, (n, _rel(got[n], z["step0/grad/" + n]))
    allg = np.concatenate([got[n].ravel() for n in names])
    allr = np.concatenate([z["step0/grad/" + n].ravel() for n in names])
    ref_bf16 = np.concatenate([z["bf16/grad/" + n].ravel() for n in names])
    assert _rel(allg, allr) < 1.5e-2
    assert _rel(allg, allr) < 3 * _rel(ref_bf16, allr) + 2e-3     # as close as the reference's own bf16 arithmetic


@pytest.mark.parametrize("name", ["tiny_hd128"])
def test_gradients_match_float64_oracle(golden_dir, name):
    from oracle import llama_train_oracle as LO

    z, cfg, sd, names = _load(golden_dir, name)
    eng = _engine(z, cfg, sd, names)
    seqs, labels = _unpack(z, 1)
    loss = float(eng.loss_and_grads(seqs, labels))
    ol, og = LO.loss_and_grads(sd, cfg, {n: z["init/" + n] for n in names}, [s.tolist() for s in seqs],
                               [l.tolist() for l in labels], int(z["lora_r"]), int(z["lora_alpha"]))
    assert abs(loss - ol) < 4e-3
    got = eng.named(eng.grads)
    for n in names:
        assert _rel(got[n].cpu().numpy(), og[n]) < 3e-2, n


@pytest.mark.parametrize("name", CONFIGS)
def test_two_adamw_steps_follow_reference(golden_dir, name):
    z, cfg, sd, names = _load(golden_dir, name)
    eng = _engine(z, cfg, sd, names)
    lr = 2e-4
    for step in range(2):
        seqs, labels = _unpack(z, step)
        loss = float(eng.loss_and_grads(seqs, labels))
        assert abs(loss - float(z[f"step{step}/loss"])) < 1e-2   # bf16 logits, 6-8 labelled tokens
        norm = float(eng.apply(lr, float(z[f"step{step}/clip_limit"])))
        assert abs(norm - float(z[f"step{step}/grad_norm"])) < 2e-2 * float(z[f"step{step}/grad_norm"])
        p = eng.named()
        for n in names:
            ref = z[f"step{step}/param/" + n]
            diff = np.abs(p[n].cpu().numpy() - ref)
            g = np.abs(z["step0/grad/" + n])
            if step == 0:
                # Adam's first step is lr * g / (|g| + eps): where the gradient is well above the bf16 noise the update
                # agrees closely; elsewhere it is bounded by 2 lr (a sign flip)
                big = g > 0.05 * g.max()
                assert diff[big].max() < 0.1 * lr, n
            assert diff.max() <= 2.05 * lr * (step + 1), n
            assert diff.mean() < 0.25 * lr, n


def test_gradient_accumulation_and_scale(golden_dir):
    """HF Trainer's accumulation: two micro-batches at grad_scale 1/2, the second added to the first."""
    z, cfg, sd, names = _load(golden_dir, "tiny_hd16")
    eng = _engine(z, cfg, sd, names)
    s0, l0 = _unpack(z, 0)
    s1, l1 = _unpack(z, 1)
    eng.loss_and_grads(s0, l0)
    g0 = eng.grads.clone()
    eng.loss_and_grads(s1, l1)
    g1 = eng.grads.clone()
    eng.loss_and_grads(s0, l0, grad_scale=0.5)
    eng.loss_and_grads(s1, l1, grad_scale=0.5, accumulate=True)
    want = 0.5 * (g0 + g1)
    assert torch.allclose(eng.grads, want, rtol=2e-2, atol=2e-3 * want.abs().max().item())


def test_live_adapter_scores_equal_merged_inference(golden_dir):
    """Validation during training scores with the adapters live; the inference path merges them at load
    (W + (alpha/r) B A). Same scores to bf16 resolution."""
    from llamarec_amd.llm import LlamaRanker

    z, cfg, sd, names = _load(golden_dir, "tiny_hd128")
    eng = _engine(z, cfg, sd, names)
    seqs, _ = _unpack(z, 0)
    label_ids = np.arange(10, 30, dtype=np.int32)
    live = eng.scores(seqs, label_ids).cpu().numpy()
    weights = {}
    for n in names:
        _, l, proj, ab = n.split(".")
        weights[f"model.layers.{l}.self_attn.{proj}.{ab}.weight"] = z["init/" + n]
    merged = LlamaRanker.from_state_dict(sd, cfg, lora=dict(r=int(z["lora_r"]), alpha=float(z["lora_alpha"]),
                                                            weights=weights))
    ref = merged.prefill_verbalize(seqs, label_ids).cpu().numpy()
    assert np.abs(live - ref).max() < 3e-2 * max(1.0, np.abs(ref).max())


def test_dropout_changes_the_adapter_path_only_and_is_reproducible(golden_dir):
    z, cfg, sd, names = _load(golden_dir, "tiny_hd16")
    seqs, labels = _unpack(z, 0)
    base = float(_engine(z, cfg, sd, names).loss_and_grads(seqs, labels))
    a = _engine(z, cfg, sd, names, dropout=0.3, seed=9)
    b = _engine(z, cfg, sd, names, dropout=0.3, seed=9)
    la, lb = float(a.loss_and_grads(seqs, labels)), float(b.loss_and_grads(seqs, labels))
    assert la == lb and torch.equal(a.grads, b.grads) or torch.allclose(a.grads, b.grads, rtol=1e-3, atol=1e-6)
    assert la != base and abs(la - base) < 0.2          # only the rank-8 path is perturbed
    l2 = float(a.loss_and_grads(seqs, labels))          # next pass: another mask
    assert l2 != la


def test_bad_arguments_fail_loudly(golden_dir):
    from llamarec_amd._lib import LlamaRecError
    from llamarec_amd.llm import LlamaRanker
    from llamarec_amd.rank_train import LoraTrainEngine

    z, cfg, sd, names = _load(golden_dir, "tiny_hd16")
    ranker = LlamaRanker.from_state_dict(sd, cfg)
    with pytest.raises(LlamaRecError):
        LoraTrainEngine(ranker, r=17)
    eng = _engine(z, cfg, sd, names)
    seqs, labels = _unpack(z, 0)
    with pytest.raises(ValueError):
        eng.loss_and_grads(seqs, [np.full(len(s), -100) for s in seqs])
    bad = [l.copy() for l in labels]
    bad[0][-1] = 10 ** 6                                  # not a token id: ignored and counted, no fault
    eng.loss_and_grads(seqs, bad)
    assert eng.bad_targets == 1
    assert torch.isfinite(eng.grads).all()


def test_training_reduces_the_loss_from_peft_init(golden_dir):
    """peft's init (B = 0): the first step moves only B, then both; 40 steps on one batch must fit it."""
    from llamarec_amd.llm import LlamaRanker
    from llamarec_amd.rank_train import LoraTrainEngine

    z, cfg, sd, names = _load(golden_dir, "tiny_hd128")
    eng = LoraTrainEngine(LlamaRanker.from_state_dict(sd, cfg), dropout=0.05, seed=3)
    seqs, labels = _unpack(z, 0)
    first = float(eng.loss_and_grads(seqs, labels))
    g = eng.named(eng.grads)
    assert all(float(g[f"layers.{l}.{p}_proj.lora_A"].abs().max()) == 0.0 for l in range(2) for p in "qv")
    assert all(float(g[f"layers.{l}.{p}_proj.lora_B"].abs().max()) > 0.0 for l in range(2) for p in "qv")
    for _ in range(40):
        eng.apply(3e-3, 1.0)
        last = float(eng.loss_and_grads(seqs, labels))
    assert last < first - 1.0, (first, last)


def test_full_size_llama2_7b_properties():
    """BASELINE.json's model (32 layers, hidden 4096, 32 heads of 128, vocab 32000), random weights, every kernel on
    its production shape. Size-independent properties: with B = 0 the adapters are inert, so (1) the loss equals the
    cross-entropy computed from the INFERENCE path's logits (an independent code path: fused-rotary GEMM epilogue,
    fused SwiGLU, pruned last layer), (2) d A is exactly zero while d B is not; (3) gradients are linear in
    grad_scale; (4) the loss does not depend on the order of the prompts in the micro-batch."""
    from llamarec_amd.llm import LLAMA2_7B, LlamaRanker
    from llamarec_amd.rank_train import LoraTrainEngine

    ranker = LlamaRanker.random_init(LLAMA2_7B, seed=5)
    eng = LoraTrainEngine(ranker, dropout=0.0, seed=1)
    rng = np.random.default_rng(2)
    lens = [257, 130, 301, 64]
    seqs = [np.concatenate([[1], rng.integers(3, 32000, size=n - 2), [2]]).astype(np.int32) for n in lens]
    labels = []
    for s in seqs:
        l = np.full(len(s), -100, np.int64)
        l[-2:] = s[-2:]
        labels.append(l)
    loss = float(eng.loss_and_grads(seqs, labels))
    g1 = eng.grads.clone()
    assert np.isfinite(loss) and torch.isfinite(g1).all()
    # (1) inference-path logits of the prefixes that predict the two labelled tokens
    ref = 0.0
    for s in seqs:
        for cut in (2, 1):
            logits = ranker.last_logits([s[:-cut]])[0].double()
            ref += float(torch.logsumexp(logits, 0) - logits[int(s[len(s) - cut])])
    ref /= 2 * len(seqs)
    assert abs(loss - ref) < 2e-2, (loss, ref)
    # (2)
    g = eng.named(g1)
    for l in (0, 15, 31):
        for p in "qv":
            assert float(g[f"layers.{l}.{p}_proj.lora_A"].abs().max()) == 0.0
            assert float(g[f"layers.{l}.{p}_proj.lora_B"].abs().max()) > 0.0
    # (3)
    eng.loss_and_grads(seqs, labels, grad_scale=0.25)
    assert torch.allclose(eng.grads, 0.25 * g1, rtol=2e-2, atol=2e-3 * float(g1.abs().max()))
    # (4)
    perm = [2, 0, 3, 1]
    loss_p = float(eng.loss_and_grads([seqs[i] for i in perm], [labels[i] for i in perm]))
    assert abs(loss_p - loss) < 2e-3
    assert torch.allclose(eng.grads, g1, rtol=5e-2, atol=5e-3 * float(g1.abs().max()))


@pytest.mark.parametrize("r,lens", [(4, [300, 150, 33]), (16, [129, 260])])
def test_other_ranks_and_long_prompts_match_float64_oracle(golden_dir, r, lens):
    """LoRA ranks other than the reference's 8 (padding rows of the 16-wide MFMA tile must stay inert) and prompts that
    span several attention blocks (128 query rows / 64 keys per block), against the float64 oracle."""
    from llamarec_amd.llm import LlamaRanker
    from llamarec_amd.rank_train import LoraTrainEngine
    from llamarec_amd.synth import bf16_round, hash_uniform
    from oracle import llama_train_oracle as LO

    z, cfg, sd, _ = _load(golden_dir, "tiny_hd128")
    cfg = dict(cfg, max_position_embeddings=512)
    eng = LoraTrainEngine(LlamaRanker.from_state_dict(sd, cfg), r=r, alpha=2 * r, dropout=0.0)
    init = {}
    for i, (k, v) in enumerate(sorted(eng.named().items())):
        init[k] = bf16_round(hash_uniform(900 + i, tuple(v.shape), 0.05))
    eng.load(init)
    rng = np.random.default_rng(r)
    seqs = [np.concatenate([[1], rng.integers(3, cfg["vocab_size"], size=n - 2), [2]]).astype(np.int32) for n in lens]
    labels = [np.where(np.arange(len(s)) >= len(s) - 2, s, -100) for s in seqs]
    loss = float(eng.loss_and_grads(seqs, labels))
    ol, og = LO.loss_and_grads(sd, cfg, init, [s.tolist() for s in seqs], [l.tolist() for l in labels], r, 2 * r)
    assert abs(loss - ol) < 1e-2
    got = eng.named(eng.grads)
    for n in sorted(init):
        assert _rel(got[n].cpu().numpy(), og[n]) < 4e-2, (n, _rel(got[n].cpu().numpy(), og[n]))
    allg = np.concatenate([got[n].cpu().numpy().ravel() for n in sorted(init)])
    allr = np.concatenate([og[n].ravel() for n in sorted(init)])
    assert _rel(allg, allr) < 2e-2


def test_workspace_and_position_limits_are_checked(golden_dir):
    from llamarec_amd._lib import LlamaRecError, check, lib, stream_ptr

    z, cfg, sd, names = _load(golden_dir, "tiny_hd16")
    eng = _engine(z, cfg, sd, names)
    too_long = [np.concatenate([[1], np.full(cfg["max_position_embeddings"], 5), [2]]).astype(np.int32)]
    with pytest.raises(LlamaRecError, match="max_positions"):
        eng.loss_and_grads(too_long, [np.where(np.arange(len(too_long[0])) >= len(too_long[0]) - 2, too_long[0], -100)])
    # a workspace sized for 8 tokens cannot take 131
    seqs, labels = _unpack(z, 0)
    eng._ws = torch.empty(lib().lr_llama_lora_workspace_bytes(eng._h, 8, 1, 2), dtype=torch.uint8, device=eng.device)
    eng._workspace = lambda n, B, m: eng._ws
    with pytest.raises(LlamaRecError, match="workspace"):
        eng.loss_and_grads(seqs, labels)


@pytest.mark.parametrize("name", ["tiny_hd128", "tiny_gqa"])
def test_minimal_and_ragged_micro_batches_match_oracle(golden_dir, name):
    """The shortest sample the reference's collate accepts (3 tokens: labels[-3] == -100), alone and packed with
    neighbours of very different length."""
    from oracle import llama_train_oracle as LO

    z, cfg, sd, names = _load(golden_dir, name)
    init = {n: z["init/" + n] for n in names}
    rng = np.random.default_rng(4)
    for lens in ([3], [3, 70, 4, 129]):
        eng = _engine(z, cfg, sd, names)
        seqs = [np.concatenate([[1], rng.integers(3, cfg["vocab_size"], size=n - 2), [2]]).astype(np.int32) for n in lens]
        labels = [np.where(np.arange(len(s)) >= len(s) - 2, s, -100) for s in seqs]
        loss = float(eng.loss_and_grads(seqs, labels))
        ol, og = LO.loss_and_grads(sd, cfg, init, [s.tolist() for s in seqs], [l.tolist() for l in labels],
                                   int(z["lora_r"]), int(z["lora_alpha"]))
        assert abs(loss - ol) < 1.5e-2, (lens, loss, ol)
        got = eng.named(eng.grads)
        allg = np.concatenate([got[n].cpu().numpy().ravel() for n in names])
        allr = np.concatenate([og[n].ravel() for n in names])
        assert _rel(allg, allr) < 3e-2, (lens, _rel(allg, allr))


def test_gradients_with_dropout_match_the_oracle_under_the_same_mask(golden_dir):
    """The adapters' input dropout (config.py:259, p = 0.05; here 0.3 so that it matters): the oracle is handed the
    mask of the kernels' counter-based stream (restated in oracle/llama_train_oracle.py), so forward AND backward must
    have used that same mask in every layer for the gradients to agree."""
    from oracle import llama_train_oracle as LO

    z, cfg, sd, names = _load(golden_dir, "tiny_hd128")
    p, seed = 0.3, 21
    eng = _engine(z, cfg, sd, names, dropout=p, seed=seed)
    seqs, labels = _unpack(z, 0)
    n = sum(len(s) for s in seqs)
    for pass_no in (1, 2):                                  # the stream moves on with every micro-batch
        loss = float(eng.loss_and_grads(seqs, labels))
        masks = [LO.drop_mask(seed, pass_no, l, n, cfg["hidden_size"], p) for l in range(cfg["num_hidden_layers"])]
        assert 0.25 < 1.0 - (masks[0] > 0).mean() < 0.35
        ol, og = LO.loss_and_grads(sd, cfg, {k: z["init/" + k] for k in names}, [s.tolist() for s in seqs],
                                   [l.tolist() for l in labels], int(z["lora_r"]), int(z["lora_alpha"]), drop_masks=masks)
        assert abs(loss - ol) < 1e-2, (pass_no, loss, ol)
        got = eng.named(eng.grads)
        allg = np.concatenate([got[k].cpu().numpy().ravel() for k in names])
        allr = np.concatenate([og[k].ravel() for k in names])
        assert _rel(allg, allr) < 2e-2, (pass_no, _rel(allg, allr))
        # and the mask matters: without it the oracle's gradient is far from the engine's
        _, og0 = LO.loss_and_grads(sd, cfg, {k: z["init/" + k] for k in names}, [s.tolist() for s in seqs],
                                   [l.tolist() for l in labels], int(z["lora_r"]), int(z["lora_alpha"]))
        assert _rel(allg, np.concatenate([og0[k].ravel() for k in names])) > 0.1


def test_eight_step_loss_trajectory_follows_the_oracle(golden_dir):
    """Eight clipped AdamW steps at a learning rate large enough to move the loss by > 1: the bf16 HIP engine stays on
    the float64 oracle's loss curve (optimizer state, bias corrections, clipping and the adapter refresh every step)."""
    from oracle import llama_train_oracle as LO

    z, cfg, sd, names = _load(golden_dir, "tiny_hd128")
    eng = _engine(z, cfg, sd, names)
    seqs, labels = _unpack(z, 0)
    sl, ll = [s.tolist() for s in seqs], [l.tolist() for l in labels]
    params = {n: z["init/" + n].astype(np.float64) for n in names}
    m = {n: np.zeros_like(params[n]) for n in names}
    v = {n: np.zeros_like(params[n]) for n in names}
    lr, got, ref = 2e-3, [], []
    for step in range(8):
        got.append(float(eng.loss_and_grads(seqs, labels)))
        eng.apply(lr, 1.0)
        ol, og = LO.loss_and_grads(sd, cfg, params, sl, ll, int(z["lora_r"]), int(z["lora_alpha"]))
        LO.clip_and_adamw(params, og, m, v, step + 1, lr, 1.0)
        ref.append(ol)
    assert ref[0] - ref[-1] > 1.0, ref
    assert max(abs(a - b) for a, b in zip(got, ref)) < 3e-2, (got, ref)


def test_lds_staged_token_reductions_equal_the_gather_kernel(golden_dir, tmp_path):
    """d A / d B of every adapter come from lt_tn_lds_kernel (16-byte loads -> LDS -> ds_read_b64_tr_b16) since round 5; the 2-byte
    gather kernel it replaces builds the same MFMA fragments (incl. the dropout mask and the zero rows past a chunk's end) and is
    kept for unaligned operands. LR_TN_GATHER=1 (read once per process, hence the child) forces it: every gradient tensor must
    agree to the order of the fp32 atomics that combine the token chunks."""
    import subprocess
    import sys

    z, cfg, sd, names = _load(golden_dir, "tiny_hd128")
    out = str(tmp_path / "gather.npz")
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "from tests import test_gpu_llama_train as T\n"
        f"z, cfg, sd, names = T._load({golden_dir!r}, 'tiny_hd128')\n"
        "eng = T._engine(z, cfg, sd, names, dropout=0.3, seed=21)\n"
        "seqs, labels = T._unpack(z, 0)\n"
        "loss = float(eng.loss_and_grads(seqs, labels))\n"
        f"np.savez({out!r}, loss=loss, **{{k: v.cpu().numpy() for k, v in eng.named(eng.grads).items()}})\n")
    env = dict(os.environ, LR_TN_GATHER="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = np.load(out)
    eng = _engine(z, cfg, sd, names, dropout=0.3, seed=21)
    seqs, labels = _unpack(z, 0)
    loss = float(eng.loss_and_grads(seqs, labels))
    assert abs(loss - float(ref["loss"])) < 2e-6            # the forward touches neither kernel (the loss is a sum of fp32 atomics: 1 ulp)
    got = eng.named(eng.grads)
    worst = 0.0
    for n in names:
        g = got[n].cpu().numpy()
        assert np.abs(ref[n]).max() > 0, n
        worst = max(worst, _rel(g, ref[n]))
        assert _rel(g, ref[n]) < 2e-6, (n, _rel(g, ref[n]))
    print(f"LDS-staged vs gather token reductions: worst relative L2 distance {worst:.2e}")
