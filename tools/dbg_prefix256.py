import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from test_gpu_llama import _prefixed_prompts, synth_llama_state
from llamarec_amd.llm import LlamaRanker
cfg = dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
           num_key_value_heads=2, max_position_embeddings=2048, rms_norm_eps=1e-5, rope_theta=10000.0)
sd = synth_llama_state(cfg, 11)
model = LlamaRanker.from_state_dict(sd, cfg)
label_ids = list(range(40, 60))
tails = [1, 5, 100, 300, 64, 27, 700, 1100, 256, 220]
for P in (1, 4, 36, 64):
    seqs = _prefixed_prompts(P, tails, 320, P)
    model.set_variants(0, 3)
    runs = {}
    for name, share in (("shared", True), ("shared2", True), ("plain", False), ("plain2", False)):
        runs[name] = model.prefill_verbalize(seqs, label_ids, share_prefix=share).cpu().numpy()
    model.set_variants(0, 2)
    v2 = model.prefill_verbalize(seqs, label_ids, share_prefix=False).cpu().numpy()
    def rows(a, b): return [i for i in range(len(seqs)) if not np.array_equal(a[i], b[i])]
    print("P=%d: shared vs shared2 differ in prompts %s; plain vs plain2 %s; shared vs plain %s; max |plain - v2| %.4f" %
          (P, rows(runs["shared"], runs["shared2"]), rows(runs["plain"], runs["plain2"]), rows(runs["shared"], runs["plain"]),
           np.abs(runs["plain"] - v2).max()), flush=True)
