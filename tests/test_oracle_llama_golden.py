"""Pin the stage-2 numpy oracle against the reference's patched LlamaForCausalLM outputs
(tests/golden/llama_*.npz from tests/gen_goldens_llm.py) and its verbalizer."""
import json
import os

import numpy as np
import pytest

from llamarec_amd.synth import synth_llama_state
from oracle import llama_oracle as LO


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"llama_{name}.npz"))
    cfg = json.loads(str(z["config"]))
    sd = synth_llama_state(cfg, int(z["weight_seed"]))
    T = z["input_ids"].shape[1]
    seqs = [z["input_ids"][b, T - n:] for b, n in enumerate(z["lens"])]
    return z, cfg, sd, seqs


@pytest.mark.parametrize("name", ["tiny_hd16", "tiny_hd128", "tiny_gqa"])
def test_fp32_matches_reference(golden_dir, name):
    z, cfg, sd, seqs = load(golden_dir, name)
    got = LO.last_logits(sd, cfg, seqs, "fp32")
    assert got.dtype == np.float32 and got.shape == z["logits_fp32"].shape
    assert np.abs(got - z["logits_fp32"]).max() < 2e-5          # left-padded batch
    assert np.abs(got - z["logits_fp32_unpadded"]).max() < 2e-5  # unpadded single prompts
    assert float(z["eval_loss"]) == -1.0                          # model/llm.py:128-129


@pytest.mark.parametrize("name", ["tiny_hd16", "tiny_hd128", "tiny_gqa"])
def test_bf16_mode_tracks_reference_bf16(golden_dir, name):
    z, cfg, sd, seqs = load(golden_dir, name)
    got = LO.last_logits(sd, cfg, seqs, "bf16")
    ref = z["logits_bf16"]
    # both are bf16 pipelines with different rounding placement; compare at bf16 resolution
    assert np.abs(got - ref).max() < 2.5e-2
    assert np.abs(got - z["logits_fp32"]).max() < 2.5e-2


def test_verbalizer_is_a_gather(golden_dir):
    z = np.load(os.path.join(golden_dir, "verbalizer.npz"))
    ids = z["label_words_ids"].reshape(-1)
    assert z["label_words_ids"].shape == (20, 1, 1)
    assert np.array_equal(LO.verbalize(z["logits"], ids), z["scores"])
