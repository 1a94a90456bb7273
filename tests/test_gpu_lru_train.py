"""Retriever training step on the GPU (csrc/lru_train.hip through the C ABI) against the float64 oracle and the
reference's own autograd / AdamW run (tests/golden/lru_train_v120.npz)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def load(golden_dir):
    z = np.load(f"{golden_dir}/lru_train_v120.npz", allow_pickle=False)
    names = [str(n) for n in z["param_names"]]
    return z, names


def as_pairs(a):
    a = np.asarray(a)
    return a.view(np.float32).reshape(a.shape + (2,)) if np.iscomplexobj(a) else a


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def make_engine(z, names, **kw):
    from llamarec_amd.train import LRUTrainEngine

    return LRUTrainEngine({n: z["init/" + n] for n in names}, dropout=0.0, attn_dropout=0.0, **kw)


@pytest.mark.parametrize("ce_mode", [1, 2])   # stored logits / fused item GEMM + cross-entropy
def test_loss_and_gradients_match_reference_and_oracle(golden_dir, ce_mode):
    from oracle import lru_train_oracle as TO

    z, names = load(golden_dir)
    eng = make_engine(z, names, ce_mode=ce_mode)
    loss = float(eng.loss_and_grads(z["tokens"], z["labels"]))
    assert abs(loss - float(z["step0/loss"])) < 2e-5
    o_loss, o_grads = TO.loss_and_grads({n: z["init/" + n] for n in names}, z["tokens"], z["labels"])
    assert abs(loss - o_loss) < 2e-5
    got = eng.grad_dict()
    assert set(got) == set(names)
    for n in names:
        g = as_pairs(got[n])
        assert g.shape == z["step0/grad/" + n].shape, n
        assert rel_err(g, o_grads[n]) < 3e-4, (n, rel_err(g, o_grads[n]))          # fp32 kernels vs float64 oracle
        assert rel_err(g, z["step0/grad/" + n]) < 5e-4, (n, rel_err(g, z["step0/grad/" + n]))  # vs torch autograd
    # parameters round-trip through the flat buffer by their reference names
    sd = eng.state_dict()
    for n in names:
        assert np.array_equal(as_pairs(sd[n]), z["init/" + n]), n


def test_two_clipped_adamw_steps_match_reference(golden_dir):
    z, names = load(golden_dir)
    eng = make_engine(z, names)
    for step in range(2):
        loss = float(eng.loss_and_grads(z["tokens"], z["labels"]))
        assert abs(loss - float(z[f"step{step}/loss"])) < 5e-5
        # the golden run clips at 5.0 (no-op) on step 0 and at 0.05 (real rescale) on step 1
        norm = float(eng.apply(max_grad_norm=float(z[f"step{step}/clip_limit"])))
        assert abs(norm - float(z[f"step{step}/grad_norm"])) < 1e-3 * float(z[f"step{step}/grad_norm"])
        sd = eng.state_dict()
        for n in names:
            ref = z[f"step{step}/param/" + n]
            solid = np.minimum(np.abs(z["step0/grad/" + n]), np.abs(z[f"step{step}/grad/" + n])) > 1e-5
            err = np.abs(as_pairs(sd[n]) - ref)
            assert err[solid].max(initial=0.0) < 2e-5 + 1e-4 * np.abs(ref).max(), (step, n)
            assert err.max() <= 2.2e-3, (step, n)


def test_trained_weights_feed_the_scoring_path(golden_dir):
    """After two steps, exporting the state_dict into the retriever gives the reference's last-position scores."""
    from llamarec_amd.lru import LRURec

    z, names = load(golden_dir)
    eng = make_engine(z, names)
    for step in range(2):
        eng.loss_and_grads(z["tokens"], z["labels"])
        eng.apply(max_grad_norm=float(z[f"step{step}/clip_limit"]))
    scores = LRURec.from_state_dict(eng.state_dict()).scores_last(z["tokens"]).cpu().numpy()
    assert np.abs(scores - z["final_scores_last"]).max() < 2e-3


@pytest.mark.parametrize("ce_mode", [1, 2])
def test_forty_step_loss_trajectory_matches_reference(golden_dir, ce_mode):
    """42 optimizer steps on the golden batch: the HIP engine stays on the reference's loss curve."""
    z, names = load(golden_dir)
    eng = make_engine(z, names, ce_mode=ce_mode)
    for step in range(2):
        eng.loss_and_grads(z["tokens"], z["labels"])
        eng.apply(max_grad_norm=float(z[f"step{step}/clip_limit"]))
    ref = z["traj_loss"]
    for i in range(len(ref)):
        loss = float(eng.train_step(z["tokens"], z["labels"]))
        assert abs(loss - float(ref[i])) < 3e-3 * (1 + i / 10), (i, loss, float(ref[i]))


def test_out_of_range_labels_are_ignored_and_counted(golden_dir):
    z, names = load(golden_dir)
    eng = make_engine(z, names)
    ref = float(eng.loss_and_grads(z["tokens"], z["labels"]))
    g_ref = eng.grad_dict()
    assert eng.bad_labels == 0
    lab = z["labels"].copy()
    tok = z["tokens"].copy()
    # two extra rows whose labels are garbage: they must not fault, not change the loss, not add gradient
    lab2 = np.concatenate([lab, np.full((2, lab.shape[1]), 10 ** 6), np.full((1, lab.shape[1]), -5)])
    tok2 = np.concatenate([tok, tok[:3]])
    loss = float(eng.loss_and_grads(tok2, lab2))
    assert eng.bad_labels == 3 * lab.shape[1]
    # the loss and the weight-gradient sums are accumulated with fp32 atomics: the extra rows change the ORDER of the terms, not the
    # terms. A row that leaked would add ~1e-3 of a tensor's scale; the order noise is ~1e-6 of it. The check is per tensor and
    # relative to the tensor's largest entry, so that a leak into a small-magnitude tensor cannot hide under an absolute bound
    # sized for the large ones.
    assert abs(loss - ref) < 1e-5
    got = eng.grad_dict()
    for n in names:
        assert rel_err(as_pairs(got[n]), as_pairs(g_ref[n])) < 2e-5, (n, rel_err(as_pairs(got[n]), as_pairs(g_ref[n])))


def test_graph_replay_equals_direct_launches(golden_dir):
    """hipGraph replay (default) and plain launches walk the same arithmetic: same losses over several steps,
    including a change of batch shape (re-capture) and of the learning rate / clip limit (device-side scalars)."""
    from llamarec_amd.train import LRUTrainEngine

    z, names = load(golden_dir)
    init = {n: z["init/" + n] for n in names}
    tok, lab = z["tokens"], z["labels"]
    runs = []
    for use_graph in (True, False):
        eng = LRUTrainEngine(init, dropout=0.1, attn_dropout=0.1, seed=11, use_graph=use_graph)
        out = []
        for i in range(6):
            t, l = (tok, lab) if i != 3 else (tok[:4, 2:], lab[:4, 2:])       # another shape in the middle
            out.append(float(eng.train_step(t, l, lr=1e-3 * (1 + i))))
            out.append(float(eng.apply(lr=0.0, max_grad_norm=0.01 * (i + 1))))  # norm read-back; lr 0 = no change but decay
        runs.append(out)
    assert np.allclose(runs[0], runs[1], rtol=2e-4, atol=1e-5), (runs[0], runs[1])


def test_loss_decreases_and_dropout_is_deterministic(golden_dir):
    from llamarec_amd.train import LRUTrainEngine

    z, names = load(golden_dir)
    init = {n: z["init/" + n] for n in names}
    a = LRUTrainEngine(init, dropout=0.2, attn_dropout=0.2, seed=5)
    b = LRUTrainEngine(init, dropout=0.2, attn_dropout=0.2, seed=5)
    la = [float(a.train_step(z["tokens"], z["labels"])) for _ in range(30)]
    lb = [float(b.train_step(z["tokens"], z["labels"])) for _ in range(3)]
    # same seed -> same dropout masks; the sums behind parameter gradients use fp32 atomics, so runs agree to
    # rounding, not bit for bit
    assert np.allclose(la[:3], lb, rtol=0, atol=1e-4)
    assert la[-1] < la[0] - 0.5              # it learns the batch
    c = LRUTrainEngine(init, dropout=0.2, attn_dropout=0.2, seed=6)
    assert abs(float(c.train_step(z["tokens"], z["labels"])) - la[0]) > 1e-3   # another seed, other masks
    assert all(np.isfinite(la))


def test_gradients_with_dropout_match_the_oracle_under_the_same_masks(golden_dir):
    """Dropout at the reference's four sites (embedding, LRU-layer output, FFN activation, FFN output; config.py
    bert_dropout / bert_attn_dropout 0.2) with the kernels' counter-based stream restated in the oracle: forward and
    backward of every site must have used the same mask for the gradients to agree -- on two consecutive passes."""
    from oracle import lru_train_oracle as TO
    from llamarec_amd.train import LRUTrainEngine

    z, names = load(golden_dir)
    init = {n: z["init/" + n] for n in names}
    pd, pa, seed = 0.2, 0.3, 17
    eng = LRUTrainEngine(init, dropout=pd, attn_dropout=pa, seed=seed)
    for pass_no in range(2):
        loss = float(eng.loss_and_grads(z["tokens"], z["labels"]))
        o_loss, o_grads = TO.loss_and_grads(init, z["tokens"], z["labels"],
                                            dropout=(TO.pass_seed(seed, pass_no), pd, pa))
        assert abs(loss - o_loss) < 5e-5, (pass_no, loss, o_loss)
        got = eng.grad_dict()
        for n in names:
            assert rel_err(as_pairs(got[n]), o_grads[n]) < 5e-4, (pass_no, n, rel_err(as_pairs(got[n]), o_grads[n]))
    assert abs(o_loss - float(z["step0/loss"])) > 1e-3            # the masks matter


@pytest.mark.parametrize("B,L", [(3, 7), (5, 50), (64, 50), (130, 200), (340, 200)])   # 21 / 250 rows: ragged 16-row panels, 64-row slices; 26 000 / 68 000 rows: the 128- and 256-row weight-gradient slices, one item split
def test_row_panel_and_generic_block_kernels_agree(golden_dir, B, L):
    """The LRU blocks run as row-panel kernels (csrc/lru_train_blocks.hip) by default; lr_lru_train_set_fused(h, 0) selects
    one generic GEMM launch per product. Same mathematics and the same dropout masks (same seed, same (site, element)
    counters), other summation orders: loss and every gradient agree to fp32 rounding, on two consecutive passes."""
    from llamarec_amd.train import LRUTrainEngine

    z, names = load(golden_dir)
    init = {n: z["init/" + n] for n in names}
    V = z["init/embedding.token.weight"].shape[0] - 1
    rng = np.random.default_rng(B * 100 + L)
    tok = rng.integers(1, V + 1, size=(B, L)).astype(np.int64)
    lab = rng.integers(1, V + 1, size=(B, L)).astype(np.int64)
    for b in range(B):                      # left padding of random length, as the dataloader produces it
        n_pad = int(rng.integers(0, L - 1))
        tok[b, :n_pad] = 0
        lab[b, :n_pad] = 0
    engines = []
    for fused in (1, 0):
        e = LRUTrainEngine(init, dropout=0.2, attn_dropout=0.3, seed=11)
        e.set_fused(fused)
        engines.append(e)
    for pass_no in range(2):
        losses = [float(e.loss_and_grads(tok, lab)) for e in engines]
        # the mean of up to 68 000 row terms, summed in fp32 by atomics in an order that differs between the two forms AND between
        # runs: 4e-6 relative has been observed at 68 000 rows (loss 4.81), so the bound is relative, not 2e-5 absolute
        assert abs(losses[0] - losses[1]) < 2e-5 * max(1.0, abs(losses[1])), (pass_no, losses)
        ga, gb = engines[0].grad_dict(), engines[1].grad_dict()
        for n in names:
            assert rel_err(as_pairs(ga[n]), as_pairs(gb[n])) < 5e-5, (pass_no, n, rel_err(as_pairs(ga[n]), as_pairs(gb[n])))


@pytest.mark.parametrize("ce_mode", [1, 2])   # row-panel score kernels over stored logits / fused item GEMM + cross-entropy
def test_deterministic_mode_is_bit_identical_run_to_run(golden_dir, ce_mode):
    """lr_lru_train_set_deterministic: every fp32 atomic of the pass becomes a 64-bit fixed-point add into
    a shadow buffer that is folded back at fixed points (csrc/lr_det.h), the gradient norm is summed by one workgroup. Two engines
    from the same state -- with dropout, over three optimizer steps, with a batch big enough that dozens of workgroups add into
    the same gradient entries -- then agree BIT FOR BIT in loss, gradient norm, every gradient tensor and every parameter; with
    hipGraph replay as without. And the mode changes no value beyond the atomics' rounding: gradients agree with the default
    mode's to 2e-5 of each tensor's scale."""
    from llamarec_amd.train import LRUTrainEngine

    z, names = load(golden_dir)
    init = {n: z["init/" + n] for n in names}
    rng = np.random.default_rng(3)
    V = int(z["init/embedding.token.weight"].shape[0]) - 1
    tok = rng.integers(1, V + 1, size=(48, 50)).astype(np.int64)
    tok[::3, :17] = 0                                     # left-padded rows
    lab = np.where(tok > 0, rng.integers(1, V + 1, size=tok.shape), 0).astype(np.int64)

    def run(det, use_graph):
        eng = LRUTrainEngine(init, dropout=0.2, attn_dropout=0.2, seed=9, use_graph=use_graph, ce_mode=ce_mode)
        if det:
            eng.set_deterministic(True)
        out = []
        for step in range(3):
            loss = eng.loss_and_grads(tok, lab)
            grads = {k: as_pairs(v).copy() for k, v in eng.grad_dict().items()}
            norm = eng.apply(lr=1e-3, max_grad_norm=0.5)
            out.append((float(loss), float(norm), grads, {k: as_pairs(v).copy() for k, v in eng.state_dict().items()}))
        return out

    a, b, c = run(True, False), run(True, False), run(True, True)
    for other in (b, c):
        for (la, na, ga, pa), (lb, nb, gb, pb) in zip(a, other):
            assert la == lb and na == nb, (la, lb, na, nb)
            for n in names:
                assert np.array_equal(ga[n], gb[n]), n
                assert np.array_equal(pa[n], pb[n]), n
    plain = run(False, False)
    assert abs(a[0][0] - plain[0][0]) < 1e-5
    for n in names:
        assert np.abs(a[0][2][n]).max() > 0, n
        assert rel_err(a[0][2][n], plain[0][2][n]) < 2e-5, (n, rel_err(a[0][2][n], plain[0][2][n]))


def test_deterministic_mode_against_the_float64_oracle_and_the_ignored_rows(golden_dir):
    """Under deterministic mode the comparisons that had to leave room for the atomics' order get their tight bounds back: loss
    and gradients against the float64 oracle, and a batch with extra rows whose labels are out of range gives the SAME bits as
    the batch without them wherever no term changed (the ignored rows add exact zeros to the fixed-point shadows)."""
    from oracle import lru_train_oracle as TO

    z, names = load(golden_dir)
    eng = make_engine(z, names).set_deterministic(True)
    ref = float(eng.loss_and_grads(z["tokens"], z["labels"]))
    g_ref = {k: as_pairs(v).copy() for k, v in eng.grad_dict().items()}
    o_loss, o_grads = TO.loss_and_grads({n: z["init/" + n] for n in names}, z["tokens"], z["labels"])
    assert abs(ref - o_loss) < 2e-5
    for n in names:
        assert rel_err(g_ref[n], o_grads[n]) < 3e-4, (n, rel_err(g_ref[n], o_grads[n]))
    lab, tok = z["labels"].copy(), z["tokens"].copy()
    lab2 = np.concatenate([lab, np.full((2, lab.shape[1]), 10 ** 6), np.full((1, lab.shape[1]), -5)])
    tok2 = np.concatenate([tok, tok[:3]])
    loss = float(eng.loss_and_grads(tok2, lab2))
    assert eng.bad_labels == 3 * lab.shape[1]
    assert np.isclose(loss, ref, rtol=1e-4, atol=1e-7)
    got = eng.grad_dict()
    for n in names:
        assert np.allclose(as_pairs(got[n]), g_ref[n], rtol=1e-4, atol=1e-7), n
    # the generic (one GEMM launch per product) path splits K with atomics into activation buffers: refused, loudly
    eng.set_fused(False)
    with pytest.raises(RuntimeError, match="deterministic mode"):
        eng.loss_and_grads(z["tokens"], z["labels"])


def test_deterministic_mode_does_not_hide_a_non_finite_gradient(golden_dir):
    """A NaN in the parameters must surface as a non-finite loss in deterministic mode as it does in the default mode: an addend that
    is not finite (or outside the fixed-point range) takes the plain fp32 atomic instead of the shadow (csrc/lr_det.h)."""
    from llamarec_amd.train import LRUTrainEngine

    z, names = load(golden_dir)
    init = {n: z["init/" + n].copy() for n in names}
    init["embedding.token.weight"][5, 3] = np.nan
    for det in (False, True):
        eng = LRUTrainEngine(init, dropout=0.0, attn_dropout=0.0)
        if det:
            eng.set_deterministic(True)
        tok = z["tokens"].copy()
        tok[0, -1] = 5                                   # the poisoned item is in the batch
        loss = float(eng.loss_and_grads(tok, z["labels"]))
        assert not np.isfinite(loss), (det, loss)
