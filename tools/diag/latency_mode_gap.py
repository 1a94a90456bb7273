"""How far apart are the default path and latency mode (split-K GEMMs) on the 32-layer random model, relative to the scores'
scale? (test_full_depth_llama2_7b_properties bounds it at 5e-2.) Run with LLAMAREC_LIB to compare library builds."""
import numpy as np, torch, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from llamarec_amd.llm import LLAMA2_7B, LlamaRanker
model = LlamaRanker.random_init(dict(LLAMA2_7B), seed=3)
rng = np.random.default_rng(1)
lens = [460, 1, 700, 129, 300]
seqs = [np.concatenate([[1], rng.integers(3, 32000, size=n - 1)]) if n > 1 else np.array([1]) for n in lens]
label_ids = list(range(319, 339))
a = model.prefill_verbalize(seqs, label_ids)
scale = max(1.0, float(a.abs().max()))
full_last = model.set_last_layer_pruning(False).prefill_verbalize(seqs, label_ids)
model.set_last_layer_pruning(True)
lat = model.set_variants(5, 0).prefill_verbalize(seqs[:2], label_ids)
model.set_variants(0, 0)
v3 = model.set_variants(0, 3).prefill_verbalize(seqs, label_ids)
model.set_variants(0, 0)
print(os.environ.get("LLAMAREC_LIB", "product"), "scale %.3f  full-last gap %.4f  latency gap %.4f  (per prompt %s)  variant-3 gap %.4f" % (
    scale, float((full_last - a).abs().max()) / scale, float((lat - a[:2]).abs().max()) / scale,
    [round(float(x), 4) for x in (lat - a[:2]).abs().amax(1) / scale], float((v3 - a).abs().max()) / scale))
