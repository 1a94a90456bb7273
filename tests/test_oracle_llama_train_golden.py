"""The LoRA-training oracle (oracle/llama_train_oracle.py) against the goldens made by running the reference's
training forward + torch autograd + AdamW (tests/gen_goldens_rank_train.py)."""
import json
import os

import numpy as np
import pytest

from llamarec_amd.synth import synth_llama_state
from oracle import llama_train_oracle as LO

CONFIGS = ["tiny_hd16", "tiny_hd128", "tiny_gqa"]


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"llama_lora_train_{name}.npz"))
    cfg = json.loads(str(z["config"]))
    sd = synth_llama_state(cfg, int(z["weight_seed"]))
    names = [str(n) for n in z["param_names"]]
    return z, cfg, sd, names


def unpack(z, step):
    lens = z[f"step{step}/lens"]
    ids, lab = z[f"step{step}/packed_ids"], z[f"step{step}/packed_labels"]
    cu = np.concatenate([[0], np.cumsum(lens)])
    return ([ids[cu[i]:cu[i + 1]].tolist() for i in range(len(lens))],
            [lab[cu[i]:cu[i + 1]].tolist() for i in range(len(lens))])


@pytest.mark.parametrize("name", CONFIGS)
def test_two_steps_match_reference(golden_dir, name):
    z, cfg, sd, names = load(golden_dir, name)
    params = {n: z["init/" + n].astype(np.float64) for n in names}
    m = {n: np.zeros_like(params[n]) for n in names}
    v = {n: np.zeros_like(params[n]) for n in names}
    sure = {}
    for step in range(2):
        seqs, labels = unpack(z, step)
        assert all(l[-3] == -100 and l[-2] != -100 for l in labels)      # the reference's label layout
        loss, grads = LO.loss_and_grads(sd, cfg, params, seqs, labels, int(z["lora_r"]), int(z["lora_alpha"]))
        assert abs(loss - float(z[f"step{step}/loss"])) < 2e-5
        for n in names:
            ref = z[f"step{step}/grad/" + n]
            assert np.abs(grads[n] - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, n
        norm = LO.clip_and_adamw(params, grads, m, v, step + 1, 2e-4, float(z[f"step{step}/clip_limit"]))
        assert abs(norm - float(z[f"step{step}/grad_norm"])) < 2e-4 * norm
        for n in names:
            ref = z[f"step{step}/param/" + n]
            # where Adam's own gradient is fp32 rounding noise the reference itself is only defined to +-lr
            g = np.abs(z[f"step{step}/grad/" + n])
            sure[n] = sure.get(n, True) & (g > 1e-2 * g.max())   # an early +-lr difference stays in the parameter
            big = sure[n]
            assert np.abs(params[n] - ref)[big].max() < 5e-6, n
            assert np.abs(params[n] - ref).max() < 4.1e-4, n
