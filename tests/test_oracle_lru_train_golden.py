"""The training-step oracle (oracle/lru_train_oracle.py) against the reference's own autograd / AdamW run
(tests/golden/lru_train_v120.npz, made by tests/gen_goldens_train.py). CPU only."""
import numpy as np
import pytest

from oracle import lru_train_oracle as TO


@pytest.fixture(scope="module")
def gold(golden_dir):
    z = np.load(f"{golden_dir}/lru_train_v120.npz", allow_pickle=False)
    names = [str(n) for n in z["param_names"]]
    return z, names


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def test_loss_and_every_gradient_match_reference_autograd(gold):
    z, names = gold
    state = {n: z["init/" + n] for n in names}
    loss, grads = TO.loss_and_grads(state, z["tokens"], z["labels"])
    assert abs(loss - float(z["step0/loss"])) < 2e-6
    assert set(grads) == set(names)
    for n in names:
        ref = z["step0/grad/" + n]
        assert grads[n].shape == ref.shape, n
        # fp32 autograd vs float64 restatement: relative to the tensor's largest entry
        assert rel_err(grads[n], ref) < 2e-4, (n, rel_err(grads[n], ref))
    # the one pad position that carries a label (dataloader/lru.py:119-131) really contributes
    assert np.abs(grads["embedding.token.weight"][0]).max() > 0


def test_two_adamw_steps_with_clipping_match_reference(gold):
    z, names = gold
    state = {n: z["init/" + n].astype(np.float64) for n in names}
    opt = TO.AdamW(state)
    for step in range(2):
        loss, grads = TO.loss_and_grads(state, z["tokens"], z["labels"])
        assert abs(loss - float(z[f"step{step}/loss"])) < 5e-6
        state, norm = opt.step(state, grads, max_grad_norm=float(z[f"step{step}/clip_limit"]))
        assert abs(norm - float(z[f"step{step}/grad_norm"])) < 1e-4 * norm
        for n in names:
            ref = z[f"step{step}/param/" + n]
            # Adam's update is lr * m / (sqrt(v) + 1e-9): where the gradient itself is rounding noise (|g| within
            # a few orders of eps) the float32 reference and this float64 restatement legitimately differ by up to
            # lr; everywhere else they agree to float32 resolution
            solid = np.minimum(np.abs(z["step0/grad/" + n]), np.abs(z[f"step{step}/grad/" + n])) > 1e-6
            err = np.abs(state[n] - ref)
            assert err[solid].max(initial=0.0) < 2e-6 + 2e-5 * np.abs(ref).max(), (step, n)
            assert err.max() <= 2.2e-3 and solid.mean() >= 0.45, (step, n)   # out_proj.bias: the imaginary half has no gradient at all
    # step 1 was clipped hard (limit 0.05 < norm), step 0 was not
    assert float(z["step1/grad_norm"]) > float(z["step1/clip_limit"]) and float(z["step0/grad_norm"]) < 5.0


def test_forty_step_loss_trajectory_matches_reference(gold):
    """Steps 2..41 of the reference run (losses only, limit 5.0): the float64 restatement follows the fp32
    reference's trajectory; Adam's sensitivity on noise-level gradients shows up as a slow drift, not a jump."""
    z, names = gold
    state = {n: z["init/" + n].astype(np.float64) for n in names}
    opt = TO.AdamW(state)
    for step in range(2):
        _, grads = TO.loss_and_grads(state, z["tokens"], z["labels"])
        state, _ = opt.step(state, grads, max_grad_norm=float(z[f"step{step}/clip_limit"]))
    ref = z["traj_loss"]
    for i in range(len(ref)):
        loss, grads = TO.loss_and_grads(state, z["tokens"], z["labels"])
        assert abs(loss - float(ref[i])) < 2e-3 * (1 + i / 10), (i, loss, float(ref[i]))
        state, _ = opt.step(state, grads, max_grad_norm=5.0)
    assert ref[-1] < 0.5 * ref[0]
