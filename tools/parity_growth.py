"""How far apart are the HIP prefill, the numpy oracle in bf16 mode and the oracle in fp32 mode at Llama-2-7b WIDTH as the
depth grows? (The oracle needs ~2 s per layer for these four prompts: `parity_growth.py 16,32` runs the deep end.) Prints max-abs differences of the 20
verbalizer scores: the bf16 oracle's own distance from exact arithmetic is the yardstick for the HIP path's distance
from the bf16 oracle (tests/test_gpu_llama.py::test_full_width_parity_vs_oracle)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd.llm import LlamaRanker
from llamarec_amd.synth import bf16_round, llama_param_shapes
from oracle import llama_oracle as LO

depths = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8]
vocab = 2048
lens = [64, 300, 130, 257]
label_ids = list(range(100, 120))
for L in depths:
    cfg = dict(vocab_size=vocab, hidden_size=4096, intermediate_size=11008, num_hidden_layers=L, num_attention_heads=32,
               num_key_value_heads=32, max_position_embeddings=4096, rms_norm_eps=1e-5, rope_theta=10000.0)
    rng = np.random.default_rng(123)
    sd = {}
    for name, shape in llama_param_shapes(cfg):
        sd[name] = (np.float32(1.0) + bf16_round(rng.uniform(-0.1, 0.1, shape).astype(np.float32))) if len(shape) == 1 else \
            bf16_round(rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02))
    seqs = [np.concatenate([[1], rng.integers(3, vocab, size=n - 1)]).astype(np.int32) for n in lens]
    t0 = time.time()
    ob = LO.prefill_verbalize(sd, cfg, seqs, label_ids, "bf16")
    of = LO.prefill_verbalize(sd, cfg, seqs, label_ids, "fp32")
    t1 = time.time() - t0
    model = LlamaRanker.from_state_dict(sd, cfg)
    g = model.prefill_verbalize(seqs, label_ids).cpu().numpy()
    gf = model.set_fold_norms(True).prefill_verbalize(seqs, label_ids).cpu().numpy()
    del model
    d = lambda a, b: float(np.abs(a - b).max())
    r = lambda a, b: float(np.sqrt(((a - b) ** 2).mean()))
    print(f"layers {L}: |score| max {np.abs(of).max():.2f}  max-abs  hip-bf16o {d(g, ob):.4f}  hip-fp32o {d(g, of):.4f}  "
          f"bf16o-fp32o {d(ob, of):.4f}  folded-bf16o {d(gf, ob):.4f} folded-fp32o {d(gf, of):.4f} | rms  hip-bf16o {r(g, ob):.4f} "
          f"hip-fp32o {r(g, of):.4f} bf16o-fp32o {r(ob, of):.4f} folded-bf16o {r(gf, ob):.4f} folded-fp32o {r(gf, of):.4f}  (oracle {t1:.1f} s)", flush=True)
