#!/usr/bin/env python
"""bench.py -- users/sec through the two-stage retrieve+rerank scoring path on N MI355X.

  python bench.py --gpus N --steps K --warmup W
      N = 1: runs in this process. N > 1 without torchrun's environment: this process starts N ranks itself
      (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`)
      BEFORE it touches the GPU, relays rank 0's JSON line and exits with the children's code; it fails loudly
      when the node has fewer than N GPUs. Nothing is exec-replaced.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W          (the driver's own launch; same ranks)

Workload = BASELINE.json north_star's "synthetic sequences of the Beauty shape" (configs[2]): LRURec (V = 12 086,
L = 50, D = 64, 2 blocks) retrieve top-50 with history masking, then ONE Llama-2-7b (32 layers, bf16, random
weights) prefill over the templated prompts of the same users and the verbalizer gather over the 20 candidate
letters. Synthetic data per BASELINE.md section 3. A step = one pass of the hot path over one batch of users; the
batch is the reference's eval loop re-batched by TOKEN budget (llamarec_amd/packing.py: 32 768 prompt tokens,
about 46 Beauty users, instead of 16 prompts of whatever length -- eval order is free, dataloader/llm.py:196-202,
and a prompt's scores do not depend on its batch). Inputs (history ids, labels, prompt token ids) are resident in
HBM before the timed region. One process per GPU over RCCL; users are sharded, weights replicated; the only
collective is the final all-reduce of the int64 rank histograms (inside the timed region).

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
  roofline     -- the dominant kernel (256x256x64 bf16 MFMA GEMM): algorithmic FLOPs per launch over its measured
                  duration (HIP events recorded by the library on the launch stream)
  cpu_baseline -- the CPU oracle (a port of the reference algorithm) timed on the host cores on a bounded sample
                  of the same workload (N = 1 only)
  parity       -- GPU vs that oracle on the same sample, outside the timed region (N = 1 only)
  roofline_item_gemm -- north_star's second roofline: the item GEMM + top-K at the Synth-1M shape (260 MB table, 4 096
                  users, one retrieve_topk call), timed after a warm-up, outside `value` (N = 1 only)
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_CEILING_NORMAL_OPERANDS_TFLOPS = 1800.0   # measured, round 5: profiles/r05_mfma_power_ceiling.txt
STAGE2_TOL = 3e-2          # floor of the |GPU - oracle| bound on the O(1) verbalizer scores; at full width the bound is
                           # 2 x the bf16 oracle's own distance from its fp32 mode on the same sample (tests/test_gpu_llama.py)
# SURVEY.md section 6 / BASELINE.md section 2: the REFERENCE code's own stage-1 CPU path (model/lru.py + masking +
# top-20 of trainer/lru.py), torch 2.10 CPU, 8 cores, measured in the survey container (not on the GPU box)
REFERENCE_CODE_STAGE1_USERS_PER_S = {"ml-100k": 242.0, "beauty": 554.0, "games": 1706.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="beauty")
    ap.add_argument("--token-budget", type=int, default=None, help="prompt tokens per step (default: packing.TOKEN_BUDGET)")
    ap.add_argument("--users-per-step", type=int, default=0, help="> 0: the reference's fixed-size batches "
                    "(config.py:98: 16, ML-100k 32) instead of token-budget batches")
    ap.add_argument("--no-shared-prefix", action="store_true", help="do not compute the prompts' common template prefix once per step")
    ap.add_argument("--fold-norms", action="store_true", help="A/B only: RMSNorm folded into the QKV and gate/up GEMMs "
                    "(LlamaRanker.set_fold_norms; not the default because it leaves the reference's rounding points)")
    ap.add_argument("--layers", type=int, default=32, help="Llama layers (32 = Llama-2-7b; other values are for profiling only)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (and the parity block that reuses it)")
    ap.add_argument("--no-other-shapes", action="store_true", help="skip the short ML-100k-shape and LoRA-step side measurements")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--no-item-roofline", action="store_true", help="skip the Synth-1M item-GEMM roofline side field (N = 1)")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo only to "
                    "rehearse several ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """Parent of an N-rank run: no HIP call is made here (torch.cuda.device_count() does not initialise the GPU)."""
    import torch

    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not args.share_gpu:
        print(f"bench.py: --gpus {args.gpus} but this node exposes {n_dev} GPU(s); refusing to report a "
              f"{n_dev}-GPU number as n_gpus={args.gpus}", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stderr.write(p.stdout)
        print("bench.py: the ranks printed no result line", file=sys.stderr)
    return p.returncode if p.returncode else (0 if lines else 1)


def stage2_sample(T_sample, layers=2, vocab=2048, seed=0):
    """A 2-layer slice of Llama-2-7b at full width (d 4096, d_ff 11008, 32 x 128) with random bf16-valued weights
    and prompts of the given lengths: what the CPU oracle can finish in seconds."""
    from llamarec_amd.synth import bf16_round, llama_param_shapes

    cfg = dict(vocab_size=vocab, hidden_size=4096, intermediate_size=11008, num_hidden_layers=layers,
               num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=4096,
               rms_norm_eps=1e-5, rope_theta=10000.0)
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape in llama_param_shapes(cfg):
        sd[name] = np.ones(shape, np.float32) if len(shape) == 1 else \
            bf16_round(rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02))
    seqs = [np.concatenate([[1], rng.integers(3, vocab, size=int(t) - 1)]).astype(np.int32) for t in T_sample]
    return cfg, sd, seqs, list(range(100, 120))


def ndcg_at_10(ranked, labels):
    """mean NDCG@10 of one relevant item per row (trainer/utils.py:43-90 with one answer: 1 / log2(rank + 2))."""
    ranked, labels = np.asarray(ranked)[:, :10], np.asarray(labels).reshape(-1, 1)
    hit = ranked == labels
    return float((hit / np.log2(np.arange(2, ranked.shape[1] + 2))[None, :]).sum(1).mean())


def cpu_baseline_and_parity(workload, hist_ids, labels, T_sample, lru_sd, retriever, dev):
    """Oracle ("port") timed on the host: stage 1 over the resident users; stage 2 on 4 users x 4 of the 32
    Llama-2-7b layers, extrapolated x8 (a full 7B prefill is ~10 TFLOP per Beauty user); BASELINE.json configs[0]
    (ML-100k shape, retriever only, CPU) is timed beside it. The same oracle outputs are then compared with the HIP path
    on the same inputs (never inside the timed region): `labels` are the planted ones (plant_labels), so the metric
    dicts both sides derive -- the oracle's float64 rank metrics of its own top-50 against the GPU's int64 histogram
    -> lr_metrics_from_histogram path -- are non-zero and must be equal."""
    import torch

    from llamarec_amd import metrics as M
    from llamarec_amd.llm import LlamaRanker
    from llamarec_amd.lru import init_lru_state_dict
    from llamarec_amd.synth import WORKLOADS, synth_users
    from oracle import llama_oracle as LO
    from oracle import lru_oracle as O

    cores = len(os.sched_getaffinity(0))
    blas_threads = None
    try:  # threads the numpy BLAS actually uses for the stage-2 GEMMs (the dominant part)
        from threadpoolctl import threadpool_info

        blas = [i["num_threads"] for i in threadpool_info() if i.get("user_api") == "blas"]
        blas_threads = max(blas) if blas else None
    except Exception:
        pass
    omp_threads = O.num_threads()
    orc = O.LruOracle(lru_sd)
    t0 = time.perf_counter()
    o_top, _ = orc.retrieve_topk(hist_ids, 50, True)
    t1 = time.perf_counter() - t0
    s1_per_user = t1 / len(hist_ids)

    # configs[0]: ML-100k LRURec retriever-only top-20 eval on CPU (every user of the shape: 610 x 200 positions)
    w1 = WORKLOADS["ml-100k"]
    h1 = synth_users("ml-100k", w1["U"])[0]
    orc1 = O.LruOracle(init_lru_state_dict(w1["V"], seed=42))
    t0 = time.perf_counter()
    orc1.retrieve_topk(h1, 20, True)
    t_c1 = time.perf_counter() - t0

    S2_LAYERS = 4
    cfg, sd, seqs, label_ids = stage2_sample(T_sample, layers=S2_LAYERS)
    t0 = time.perf_counter()
    o_scores = LO.prefill_verbalize(sd, cfg, seqs, label_ids, mode="bf16")
    t2 = time.perf_counter() - t0
    o_exact = LO.prefill_verbalize(sd, cfg, seqs, label_ids, mode="fp32")   # untimed: the yardstick of the parity block
    s2_per_user = t2 / len(seqs) * (32 / cfg["num_hidden_layers"])
    base = {
        "value": 1.0 / (s1_per_user + s2_per_user), "unit": "users/s", "kind": "port",
        "cores": min(cores, blas_threads) if blas_threads else cores, "host_cores": cores,
        "omp_threads": omp_threads, "blas_threads": blas_threads,
        "sample": (f"stage 1: C oracle ({omp_threads} OpenMP threads) over {len(hist_ids)} {workload} users = {t1:.2f} s; "
                   f"stage 2: numpy oracle ({blas_threads} BLAS threads), {len(seqs)} users (T={[int(t) for t in T_sample]}) x "
                   f"{cfg['num_hidden_layers']} of 32 Llama-2-7b layers at full width = {t2:.2f} s, extrapolated x"
                   f"{32 // cfg['num_hidden_layers']}"),
        "stage1_users_per_s": 1.0 / s1_per_user, "stage2_users_per_s_extrapolated": 1.0 / s2_per_user,
        "config1_ml100k_retriever_only": {
            "users_per_s": len(h1) / t_c1, "users": int(len(h1)), "seconds": t_c1, "threads": omp_threads,
            "what": (f"BASELINE.json configs[0]: LRURec(V={w1['V']}, L={w1['L']}) encode + item scores + history mask + ordered "
                     f"top-20 for all {len(h1)} synthetic ML-100k users, C oracle on this host"),
            "reference_code_users_per_s_build_container": REFERENCE_CODE_STAGE1_USERS_PER_S["ml-100k"]},
        "reference_code_stage1_users_per_s": REFERENCE_CODE_STAGE1_USERS_PER_S.get(workload),
        "reference_code_note": "the reference's own torch CPU path (stage 1 only), 8 cores, measured in the build "
                               "container (SURVEY.md section 6); not measured on this host",
    }

    hist_dev = torch.from_numpy(hist_ids).to(dev)
    g_top, _ = retriever.retrieve_topk(hist_dev, 50, True)
    # the GPU's metric path on the planted labels: rank histogram kernel -> float64 metrics of the histogram
    ks = [1, 5, 10, 20, 50]
    g_metrics = M.metrics_from_histogram(M.rank_histogram(g_top, torch.from_numpy(labels).to(dev)), ks)
    g_top = g_top.cpu().numpy()
    # the oracle's: its own top-50 lists, the reference's formulas in float64 (oracle/lr_oracle.c rank metric sums)
    o_sums = O.rank_metric_sums(o_top, labels, sorted(ks, reverse=True))
    o_metrics = {}
    for j, k in enumerate(sorted(ks, reverse=True)):
        for c, name in enumerate(("Recall", "MRR", "NDCG")):
            o_metrics[f"{name}@{k}"] = float(o_sums[j, c] / len(labels))
    metrics_equal = all(abs(g_metrics[k] - o_metrics[k]) <= 1e-12 for k in o_metrics)

    small = LlamaRanker.from_state_dict(sd, cfg, device=dev)
    g_scores = small.prefill_verbalize(seqs, label_ids).cpu().numpy()
    del small
    err = float(np.abs(g_scores - o_scores).max())
    # Two bf16 evaluations of a 4096-wide network decorrelate after one layer (profiles/r02_parity_growth_full_width.txt):
    # the bound is twice the bf16 oracle's own distance from exact (fp32) arithmetic on this sample
    gap = float(np.abs(o_scores - o_exact).max())
    tol = max(STAGE2_TOL, 2 * gap)
    # ordering of the 20 candidate scores: every pair the oracle separates by more than 2 x tolerance must agree
    d_o = o_scores[:, :, None] - o_scores[:, None, :]
    d_g = g_scores[:, :, None] - g_scores[:, None, :]
    decided = np.abs(d_o) > 2 * tol
    agree = float((np.sign(d_o[decided]) == np.sign(d_g[decided])).mean()) if decided.any() else 1.0
    # planted answer letters for the two rankings' NDCG@10: prompt i's answer = the candidate the ORACLE ranks at 3 i
    o_rank = np.argsort(-o_scores, axis=1, kind="stable")
    g_rank = np.argsort(-g_scores, axis=1, kind="stable")
    lab2 = np.array([o_rank[i, (3 * i) % 20] for i in range(len(seqs))])
    parity = {
        "stage1_users": int(len(hist_ids)), "stage1_top50_equal": bool(np.array_equal(g_top, o_top)),
        "stage1_labels": f"planted: user j's label = the retriever's rank-(j mod {PLANT_PERIOD}) item (none for ranks >= 50)",
        "stage1_ndcg10_gpu": g_metrics["NDCG@10"], "stage1_ndcg10_oracle": o_metrics["NDCG@10"],
        "stage1_metrics_gpu": g_metrics, "stage1_metrics_oracle": o_metrics, "stage1_metrics_equal": bool(metrics_equal),
        "stage2_prompts": len(seqs), "stage2_layers": cfg["num_hidden_layers"], "stage2_width": cfg["hidden_size"],
        "stage2_max_abs_err": err, "stage2_tolerance": tol, "stage2_oracle_bf16_vs_fp32_max_abs": gap,
        "stage2_max_abs_err_vs_fp32_oracle": float(np.abs(g_scores - o_exact).max()), "stage2_rank_agree": agree,
        "stage2_pairs_decided": int(decided.sum() // 2),
        "stage2_ndcg10_gpu": ndcg_at_10(g_rank, lab2), "stage2_ndcg10_oracle": ndcg_at_10(o_rank, lab2),
    }
    parity["ok"] = bool(parity["stage1_top50_equal"] and metrics_equal and g_metrics["NDCG@10"] > 0 and err <= tol
                        and agree == 1.0)
    return base, parity


EPI_NAMES = {0: "store", 1: "residual", 2: "swiglu", 3: "rope", 4: "partial"}


def gemm_shapes(lib, d_model=4096, d_ff=11008):
    """Per-launch records of the 256x256x64 GEMM over the timed region (lr_profile_records), grouped by
    (epilogue, N, K): launches, mean duration, executed TFLOP/s -- the four projection shapes of a Llama-2-7b layer by
    name, everything else (the pruned last layer's products) by its tag."""
    n = lib.lr_profile_records(0, None, None, None, 0)
    if n <= 0:
        return {}
    ms, work, tag = np.zeros(n), np.zeros(n), np.zeros(n, np.int64)
    lib.lr_profile_records(0, ms.ctypes.data, work.ctypes.data, tag.ctypes.data, n)
    names = {(3, 3 * d_model, d_model): "qkv_rope", (1, d_model, d_model): "o_residual",
             (2, 2 * d_ff, d_model): "gate_up_swiglu", (1, d_model, d_ff): "down_residual"}
    out = {}
    for t in np.unique(tag):
        sel = tag == t
        epi, N, K = int(t >> 56), int((t >> 28) & ((1 << 28) - 1)), int(t & ((1 << 28) - 1))
        key = names.get((epi, N, K), f"{EPI_NAMES.get(epi, epi)}_N{N}_K{K}")
        out[key] = {"epilogue": EPI_NAMES.get(epi, str(epi)), "N": N, "K": K, "launches": int(sel.sum()),
                    "avg_ms": float(ms[sel].mean()), "tflops": float(work[sel].sum() / (ms[sel].sum() * 1e-3) / 1e12),
                    "mean_rows": float((work[sel] / (2.0 * N * K)).mean())}
    return out


def pmc_traffic(shapes):
    """HBM-side bytes per launch of the dominant kernel: per-epilogue bytes from the NEWEST committed rocprofv3 PMC
    summary (profiles/r*_pmc_summary.json, tools/summarize_pmc.py over separate --pmc passes of this command on a 2-layer
    slice: FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md section HBM), scaled to each shape's rows and weighted by
    THIS run's launch mix (round 2 averaged over the slice's own mix, where 8 of 18 launches were the small last-layer
    products: 2.9 GB where a Beauty step's mix averages 5.3). Per-shape bytes and their ratio to the algorithmic bytes
    (A + B + C once) ride along. (None, None, None) if no summary is present."""
    def round_no(p):
        m = re.search(r"r(\d+)[a-z]?_pmc_summary\.json$", p)
        return int(m.group(1)) if m else -1

    paths = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_summary.json")), key=round_no)
    paths = [p for p in paths if "stage1" not in os.path.basename(p)]
    if not paths or not shapes:
        return None, None, None
    path = paths[-1]
    d = json.load(open(path))
    per_epi = {}
    for k, e in d.items():
        m = re.match(r"gemm256rb_kernel<(\d+),", k)
        if m and "hbm_bytes_per_launch" in e:
            per_epi[EPI_NAMES.get(int(m.group(1)))] = e
    # the profiled slice ran the same token-budget step: bytes per launch of an epilogue class scale with the flops
    # of the shapes that share it (o / down share the residual class: split by flops)
    rows = max(v["mean_rows"] for v in shapes.values())
    full = {k: v for k, v in shapes.items() if v["mean_rows"] > 0.5 * rows}
    by_epi_flops = {}
    for k, v in full.items():
        by_epi_flops.setdefault(v["epilogue"], []).append(v["N"] * v["K"])
    num = den = 0.0
    per_shape = {}
    for k, v in full.items():
        e = per_epi.get(v["epilogue"])
        if e is None:
            continue
        mean_nk = float(np.mean(by_epi_flops[v["epilogue"]]))
        b = e["hbm_bytes_per_launch"] * (v["N"] * v["K"]) / mean_nk
        n_out = v["N"] // 2 if v["epilogue"] == "swiglu" else v["N"]
        alg = 2.0 * (v["mean_rows"] * v["K"] + v["N"] * v["K"] + v["mean_rows"] * n_out
                     + (v["mean_rows"] * n_out if v["epilogue"] == "residual" else 0))
        per_shape[k] = {"hbm_side_bytes_per_launch": b, "algorithmic_bytes_per_launch": alg, "ratio": b / alg}
        num += b * v["launches"]
        den += v["launches"]
    import hashlib

    sha = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    src = (f"profiles/{os.path.basename(path)} sha256:{sha} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command "
           f"on a 2-layer slice; per-epilogue bytes weighted by this run's launch mix of the full-row products; not collected by "
           f"this run: the hash names the committed file the figure was read from)")
    return (num / den if den else None), src, per_shape


def build_steps(workload, rank, n_steps, budget, users_per_step, dev, shared_prefix):
    """Resident inputs of `n_steps` steps for this rank: synthetic users of the rank's own block, grouped into steps."""
    import torch

    from llamarec_amd.llm import common_prefix_len
    from llamarec_amd.packing import token_budget_steps
    from llamarec_amd.synth import LLM_MAX_TEXT_LEN, TEMPLATE_PREFIX_TOKENS, WORKLOADS, synth_prompt_tokens, synth_users

    w = WORKLOADS[workload]
    if users_per_step > 0:
        n_users = n_steps * users_per_step
    else:
        probe = synth_users(workload, 256, first_user=rank * w["U"])[3]
        n_users = int(np.ceil((n_steps + 1.5) * budget / (float(probe.mean()) - (TEMPLATE_PREFIX_TOKENS if shared_prefix else 0))))
    n_users = min(max(n_users, 1), w["U"])
    hist, labels, _, T = synth_users(workload, n_users, first_user=rank * w["U"])
    if users_per_step > 0:
        groups = [np.arange(i, min(i + users_per_step, n_users)) for i in range(0, n_users, users_per_step)]
    else:
        # the template prefix is run once per step, so a step's rows are P + sum(T - P) (packing.token_budget_steps)
        P = TEMPLATE_PREFIX_TOKENS if shared_prefix and T.max() < LLM_MAX_TEXT_LEN else 0
        groups = token_budget_steps(T, budget, shared_prefix=P)
    groups = groups[:n_steps] if len(groups) >= n_steps else groups
    steps = []
    for b, g in enumerate(groups):
        pids, cu = synth_prompt_tokens(T[g], seed=1000 * (rank + 1) + b, shared_prefix=shared_prefix)
        steps.append(dict(hist=torch.from_numpy(hist[g]).to(dev), labels=torch.from_numpy(labels[g]).to(dev),
                          ids=torch.from_numpy(pids).to(dev), cu_dev=torch.from_numpy(cu).to(dev), cu=cu, users=len(g),
                          prefix=common_prefix_len(pids, cu) if shared_prefix else 0))
    used = np.concatenate(groups)
    return steps, hist[used], labels[used], T[used]


PLANT_PERIOD = 60   # label of the j-th resident user = the retriever's rank-(j mod 60) item; ranks >= 50 keep a random label


def plant_labels(steps, retriever, dev):
    """Uniform random labels over 12 086 items and random weights put < 1 expected hit into a whole run, so the
    histogram -> metric -> all-reduce path would only ever carry zeros (round 2's driver line). Outside the timed region
    the label of resident user j becomes the item the retriever itself ranks at position j mod 60 (no hit for >= 50):
    Recall/MRR/NDCG are then known functions of the planted ranks, printed beside what the GPU path measured. Synthetic
    labels are arbitrary by construction; the timed work does not depend on them.
    Returns (labels of all resident users in step order, their planted 0-based ranks with -1 = absent)."""
    import torch

    j0, all_labels, all_ranks = 0, [], []
    for s in steps:
        top, _ = retriever.retrieve_topk(s["hist"], 50, True)
        top = top.cpu().numpy()
        lab = s["labels"].cpu().numpy().copy()
        r = (j0 + np.arange(len(lab))) % PLANT_PERIOD
        hit = r < 50
        lab[hit] = top[np.nonzero(hit)[0], r[hit]]
        # a kept random label may sit in the top-50 by chance: the expectation is computed from where labels REALLY are
        pos = np.full(len(lab), -1, np.int64)
        where = np.nonzero(top == lab[:, None])
        pos[where[0]] = where[1]
        s["labels"] = torch.from_numpy(lab).to(dev)
        all_labels.append(lab)
        all_ranks.append(pos)
        j0 += len(lab)
    return np.concatenate(all_labels), np.concatenate(all_ranks)


def metrics_of_ranks(pos, ks, denom=None):
    """trainer/utils.py:43-90 for one relevant item per row, from its 0-based rank (-1 = not retrieved), in float64 on
    the host: the independent expectation for the GPU's histogram path."""
    pos = np.asarray(pos)
    n = len(pos) if denom is None else denom
    out = {}
    for k in ks:
        hit = (pos >= 0) & (pos < k)
        out["Recall@%d" % k] = float(hit.sum() / n)
        out["MRR@%d" % k] = float((1.0 / (pos[hit] + 1)).sum() / n)
        out["NDCG@%d" % k] = float((1.0 / np.log2(pos[hit] + 2)).sum() / n)
    return out


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    if env_world is not None and int(env_world) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}. Launch exactly N ranks: python -m "
              f"torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P "
              f"bench.py --gpus {args.gpus} ... (or plain `python bench.py --gpus {args.gpus}`, which does that itself)",
              file=sys.stderr)
        sys.exit(2)

    import torch

    from llamarec_amd import dist as D

    if args.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = D.init_from_env(args.dist_backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    ones = torch.ones(1, dtype=torch.int64, device=dev)
    D.all_reduce_sum_(ones)                    # over RCCL when world > 1: the driver can see N ranks took part
    ranks_seen = int(ones.item())
    if ranks_seen != world:
        raise SystemExit(f"bench.py: all-reduce saw {ranks_seen} ranks, expected {world}")

    from llamarec_amd import _lib
    from llamarec_amd.llm import LLAMA2_7B, LlamaRanker
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from llamarec_amd.packing import TOKEN_BUDGET
    from llamarec_amd.pipeline import TwoStagePipeline
    from llamarec_amd.synth import WORKLOADS

    w = WORKLOADS[args.workload]
    budget = args.token_budget or TOKEN_BUDGET
    shared = not args.no_shared_prefix
    steps, hist, labels, T = build_steps(args.workload, rank, max(args.steps, 1), budget, args.users_per_step, dev, shared)
    nb = len(steps)

    lru_sd = init_lru_state_dict(w["V"], seed=42)
    retriever = LRURec.from_state_dict(lru_sd, device=dev)
    cfg = dict(LLAMA2_7B, num_hidden_layers=args.layers)
    ranker = LlamaRanker.random_init(cfg, seed=42, device=dev)
    if args.fold_norms:
        ranker.set_fold_norms(True)
    label_ids = list(range(319, 339))  # stand-in ids of "A".."T" (taken from the tokenizer at run time in real use)
    pipe = TwoStagePipeline(retriever, ranker, label_ids, device=dev, shared_prefix=shared)
    planted_labels, planted_ranks = plant_labels(steps, retriever, dev)   # untimed; labels of hist[...] in step order
    labels = planted_labels

    def run(s):
        return pipe.step(s["hist"], s["labels"], s["ids"], s["cu_dev"], s["cu"], s["prefix"])

    for i in range(args.warmup):
        run(steps[i % nb])
    pipe.reset()
    torch.cuda.synchronize()

    lib = _lib.lib()
    if not args.no_profile:
        _lib.check(lib.lr_profile_start(args.steps * (args.layers * 8 + 24)), "lr_profile_start")
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run(steps[i % nb])
    retr, rer, total_users = pipe.finish()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    lib.lr_profile_stop()
    elapsed = D.all_reduce_max_float(elapsed, device=dev)

    users_rank = sum(steps[i % nb]["users"] for i in range(args.steps))
    # what the planted labels make the retrieve metrics: float64 on the host from the planted ranks of the timed steps'
    # users, summed over ranks (outside the timed region)
    first = np.concatenate([[0], np.cumsum([s["users"] for s in steps])])
    timed_ranks = np.concatenate([planted_ranks[first[i % nb]: first[i % nb + 1]] for i in range(args.steps)])
    exp_keys = ["Recall@10", "MRR@10", "NDCG@10", "NDCG@50"]
    exp_local = metrics_of_ranks(timed_ranks, [10, 50], denom=1)
    exp_t = torch.tensor([exp_local[k] for k in exp_keys], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(exp_t)
    expected = {k: float(v) / max(1, total_users) for k, v in zip(exp_keys, exp_t.tolist())}
    tok_rank = sum(int(steps[i % nb]["cu"][-1]) for i in range(args.steps))
    # rows the prefill executes: the shared prefix once per step instead of once per prompt
    rows_rank = sum(int(steps[i % nb]["cu"][-1]) - (steps[i % nb]["users"] - 1) * steps[i % nb]["prefix"] for i in range(args.steps))
    users = total_users                      # all-reduced count of users scored in the timed region
    assert world > 1 or users == users_rank

    def collect(kind):
        ms, work, n = C.c_double(), C.c_double(), C.c_int64()
        lib.lr_profile_collect(kind, C.byref(ms), C.byref(work), C.byref(n))
        return ms.value, work.value, n.value

    roofline = None
    extra = {}
    if not args.no_profile:
        g_ms, g_fl, g_n = collect(0)
        a_ms, a_fl, a_n = collect(2)
        e_ms, _, e_n = collect(4)
        k_ms, k_fl, k_n = collect(5)
        if g_n:
            ach = g_fl / (g_ms * 1e-3) / 1e12
            shapes = gemm_shapes(lib, cfg["hidden_size"], cfg["intermediate_size"])
            traffic, traffic_src, traffic_shapes = pmc_traffic(shapes)
            roofline = {"bound": "mfma", "kernel": "gemm256rb_kernel (bf16 256x256x64 MFMA tile, ping-pong pipeline; QKV+RoPE/O/gate-up+SwiGLU/down)",
                        "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                        "traffic": traffic, "traffic_source": traffic_src, "launches": g_n, "avg_launch_ms": g_ms / g_n,
                        "flops_per_launch": g_fl / g_n, "share_of_step_time": g_ms * 1e-3 / elapsed,
                        "per_shape": shapes, "traffic_per_shape": traffic_shapes,
                        # what the matrix pipe ALONE sustains on operands with the bit activity of real activations and weights
                        # (tools/diag/mfma_power.hip: nothing but MFMAs from registers; 2 476 TF/s on all-zero operands): the card's
                        # power-limited ceiling for any bf16 GEMM on such data. `frac` above stays priced against the dense peak.
                        "power_limited_mfma_ceiling": {"tflops": MFMA_CEILING_NORMAL_OPERANDS_TFLOPS,
                                                       "frac_of_it": ach / MFMA_CEILING_NORMAL_OPERANDS_TFLOPS,
                                                       "source": "profiles/r05_mfma_power_ceiling.txt"}}
        extra = {"attention_tflops": (a_fl / (a_ms * 1e-3) / 1e12) if a_n else None,
                 "attention_share_of_step_time": a_ms * 1e-3 / elapsed if a_n else None,
                 "stage1_ms_per_step": (e_ms + k_ms) / max(1, args.steps),
                 "item_topk_tflops_f32": (k_fl / (k_ms * 1e-3) / 1e12) if k_n else None}

    # stage-1-only throughput over this rank's resident users (one call; reported, not the metric)
    all_hist = torch.from_numpy(hist).to(dev)
    retriever.retrieve_topk(all_hist, 50, True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    retriever.retrieve_topk(all_hist, 50, True)
    torch.cuda.synchronize()
    stage1_users_per_s = len(hist) / (time.perf_counter() - t1)

    # placement: every rank on its own device, nothing allocated elsewhere (summed over the ranks)
    own, elsewhere = D.device_placement(local)
    place = torch.tensor([own, elsewhere], dtype=torch.int64, device=dev)
    D.all_reduce_sum_(place)
    place = [int(v) for v in place.tolist()]

    if rank == 0:
        # algorithmic prefill work of the timed region (T * 1.2952e10 + T^2 * 2.62e5 per user), whatever
        # the kernels executed (shared prefix, last-layer pruning)
        T_timed = np.concatenate([np.diff(steps[i % nb]["cu"]) for i in range(args.steps)]).astype(np.float64)
        alg_flops = float((T_timed * 1.2952e10 + T_timed ** 2 * 2.62e5).sum()) * (args.layers / 32.0)
        out = {
            "metric": "users/sec through retrieve+rerank", "value": users / elapsed, "unit": "users/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "peak_device_mem_gb_rank0": torch.cuda.max_memory_allocated(dev) / 1e9,
            "placement": {"ranks_on_their_own_device": place[0], "torch_bytes_on_other_devices": place[1], "share_gpu": bool(args.share_gpu)},
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"{args.workload} two-stage: LRURec(V={w['V']}, L={w['L']}, D=64, 2 blocks) top-50 with "
                                    f"history mask -> Llama-2-7b ({args.layers} layers, bf16, random init) single prefill "
                                    f"-> verbalizer over 20 candidates"),
                       "batching": (f"{args.users_per_step} users per step (reference batch)" if args.users_per_step > 0 else
                                    f"token budget {budget} per step (llamarec_amd/packing.py)"),
                       "shared_prompt_prefix_tokens": float(np.mean([steps[i % nb]["prefix"] for i in range(args.steps)])),
                       "users_per_step": users_rank / args.steps, "mean_prompt_tokens_per_step": tok_rank / args.steps,
                       "mean_rows_per_step": rows_rank / args.steps,
                       "rmsnorm_folded_into_gemm": bool(args.fold_norms), "llama_layers": args.layers, "parallelism": f"dp{world}", "collective_backend": args.dist_backend or "nccl"},
            "roofline": roofline, "stage1_only_users_per_s": stage1_users_per_s,
            "prefill_algorithmic_tflops_per_gpu": alg_flops / elapsed / 1e12,
            "metrics": {"retrieve_NDCG@10": retr["NDCG@10"], "rerank_overall_NDCG@10": rer["NDCG@10"],
                        "retrieve_Recall@10": retr["Recall@10"], "retrieve_MRR@10": retr["MRR@10"],
                        "rerank_overall_Recall@10": rer["Recall@10"], "users_counted": total_users,
                        "labels": (f"planted outside the timed region: user j's label = the retriever's rank-(j mod "
                                   f"{PLANT_PERIOD}) item, none for ranks >= 50 (bench.py plant_labels)"),
                        "expected_retrieve": expected,
                        "retrieve_matches_expected": bool(all(abs(retr[k] - v) <= 1e-9 for k, v in expected.items()))},
        }
        out.update(extra)
        if world == 1 and not args.no_other_shapes and args.layers == LLAMA2_7B["num_hidden_layers"]:
            out.update(side_measurements(args, ranker, label_ids, dev, steps, shared))
        if world == 1 and not args.no_item_roofline:
            out.update(item_gemm_roofline(dev))
        if world == 1 and not args.no_cpu_baseline:
            n1 = min(len(hist), 512)
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(args.workload, hist[:n1], labels[:n1], T[:4], lru_sd,
                                                                         retriever, dev)
        print(json.dumps(out), flush=True)
    D.barrier()


PEAK_HBM_GBPS = 8000.0     # HBM3E spec, MI355X_MICROARCH.md (6 290 GB/s measured for a float4 copy)
PEAK_F32_MFMA_TFLOPS = 157.3


def item_gemm_roofline(dev, users=4096, reps=5):
    """north_star's second roofline (the item GEMM's meaningful point is Synth-1M, whose 260 MB table does
    not sit in L2): LRURec at BASELINE.json configs[4]'s shape (V = 1e6, L = 200), `users` histories resident in HBM, ONE
    `retrieve_topk(ids, 50, exclude_history)` call per repetition, timed after a warm-up call; N = 1, never part of
    `value`. The item part (history sort, bound -> candidates -> exact rescoring, or the exact f32 pass) and the encoder
    are separated by the library's own HIP events on the launch stream (lr_profile kinds 5 and 4).
    Algorithmic work, SURVEY.md 8(d): flops = 2 * 64 * (V + 1) per user; bytes = (V + 1) * 65 * 4 (fp32 table + bias once per
    call) + B * 64 * 4 (q) + B * L * 8 (ids) + B * 50 * 8 (out). At 4 096 users the intensity is ~2 000 flop per table byte,
    i.e. MFMA-bound; the exact f32 scores would cost 0.524 TFLOP on the f32 pipe (3.3 ms at its 157 TF/s peak), the library
    instead bounds every score on the bf16 pipe twice and rescores ~300 candidates per user exactly, so `achieved` =
    algorithmic flops / item time is priced against the bf16 peak the executed instructions run on, and the f32 and HBM
    views ride along."""
    import torch

    from llamarec_amd import _lib
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from llamarec_amd.synth import WORKLOADS, synth_users

    w = WORKLOADS["synth-1m"]
    V, L = w["V"], w["L"]
    hist, _, n_hist, _ = synth_users("synth-1m", users)
    model = LRURec.from_state_dict(init_lru_state_dict(V, seed=42), device=dev)
    ids = torch.from_numpy(hist).to(dev)
    top, _ = model.retrieve_topk(ids, 50, True)          # warm-up (also sizes the workspace)
    torch.cuda.synchronize()
    lib = _lib.lib()
    _lib.check(lib.lr_profile_start(reps * 64), "lr_profile_start")
    t0 = time.perf_counter()
    for _ in range(reps):
        top, _ = model.retrieve_topk(ids, 50, True)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / reps * 1e3
    lib.lr_profile_stop()

    def collect(kind):
        ms, work, n = C.c_double(), C.c_double(), C.c_int64()
        lib.lr_profile_collect(kind, C.byref(ms), C.byref(work), C.byref(n))
        return ms.value, n.value

    enc_ms, enc_n = collect(4)
    item_ms, item_n = collect(5)
    enc_ms, item_ms = enc_ms / reps, item_ms / reps     # per call (a call may record several launches of a kind)
    top = top.cpu().numpy()
    # size-independent properties of the result (the bit-exact comparison with the oracle at this size is
    # tests/test_gpu_lru.py::test_full_size_catalog_properties / test_grouped_bound_catalogs_vs_oracle)
    distinct = bool(all(len(set(r.tolist())) == 50 for r in top[:: max(1, users // 64)]))
    no_hist = bool(all(not (set(top[u].tolist()) & set(hist[u].tolist())) for u in range(0, users, max(1, users // 64))))
    flops = 2.0 * 64 * (V + 1) * users
    nbytes = (V + 1) * 65 * 4.0 + users * 64 * 4.0 + users * L * 8.0 + users * 50 * 8.0
    ach = flops / (item_ms * 1e-3) / 1e12
    del model
    # fabric-side bytes of the item part per call, from the newest committed stage-1 PMC summary (same shape: 4 096 users)
    traffic, traffic_src = None, None
    paths = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_stage1_pmc_summary.json")))
    if paths:
        import hashlib

        pm = json.load(open(paths[-1]))
        item_kernels = [k for k in pm if k.startswith(("item_", "bound_select", "cand_rescore", "hist_sort", "topk_merge"))
                        and "hbm_bytes_per_launch" in pm[k]]
        if item_kernels:
            traffic = float(sum(pm[k]["hbm_bytes_per_launch"] for k in item_kernels))
            traffic_src = (f"profiles/{os.path.basename(paths[-1])} sha256:{hashlib.sha256(open(paths[-1], 'rb').read()).hexdigest()[:16]} "
                           f"(rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE in separate passes of tools/prof_stage1.py --only=synth-1m, "
                           f"summed over {', '.join(sorted(item_kernels))}; not collected by this run)")
    return {"roofline_item_gemm": {
        "workload": (f"synth-1m (BASELINE.json configs[4] shape): LRURec(V={V}, L={L}, D=64, 2 blocks), {users} users "
                     f"(mean history {float(n_hist.mean()):.0f}), one retrieve_topk(ids, 50, exclude_history=1) call, median-free mean of "
                     f"{reps} calls after one warm-up"),
        "kernel": "item_bound_kernel + item_cand_kernel (v_mfma_f32_32x32x16_bf16 over the packed bf16 table) -> cand_rescore_kernel (exact f32)",
        "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
        "flops_alg": flops, "bytes_alg": nbytes, "item_ms": item_ms, "encoder_ms": enc_ms, "call_ms_wall": wall_ms,
        "users_per_s": users / (wall_ms * 1e-3),
        "executed_bf16_flops": 2.0 * flops, "executed_bf16_tflops": 2.0 * ach,
        "frac_of_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
        "hbm_view": {"achieved_gbps": nbytes / (item_ms * 1e-3) / 1e9, "peak_gbps": PEAK_HBM_GBPS,
                     "frac": nbytes / (item_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                     "note": "bytes_alg once per call over the item time: the table stream is not what bounds 4 096 users"},
        "traffic": traffic, "traffic_source": traffic_src,
        "traffic_over_bytes_alg": (traffic / nbytes) if traffic else None,
        "checks": {"top50_distinct_sampled": distinct, "no_history_item_sampled": no_hist}}}


def side_measurements(args, ranker, label_ids, dev, steps, shared):
    """Reported, never the metric: (1) BASELINE.json configs[1] (ML-100k shape) through the same pipeline, 4 steps;
    (2) SURVEY.md 8(f) #4, three optimizer steps of the ranker's LoRA fine-tuning on the same weights; (3) SURVEY.md 8(f) #2,
    the retriever's training step at the reference's Beauty batch shape."""
    import torch

    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from llamarec_amd.packing import TOKEN_BUDGET
    from llamarec_amd.pipeline import TwoStagePipeline
    from llamarec_amd.synth import WORKLOADS

    out = {}
    other = "ml-100k" if args.workload != "ml-100k" else "beauty"
    wo = WORKLOADS[other]
    so, _, _, _ = build_steps(other, 0, 6, args.token_budget or TOKEN_BUDGET, args.users_per_step, dev, shared)
    ro = LRURec.from_state_dict(init_lru_state_dict(wo["V"], seed=42), device=dev)
    po = TwoStagePipeline(ro, ranker, label_ids, device=dev, shared_prefix=shared)
    for s in so[:2]:
        po.step(s["hist"], s["labels"], s["ids"], s["cu_dev"], s["cu"], s["prefix"])
    po.reset()
    torch.cuda.synchronize()
    tb = time.perf_counter()
    timed = so[2:6]
    for s in timed:
        po.step(s["hist"], s["labels"], s["ids"], s["cu_dev"], s["cu"], s["prefix"])
    po.finish()
    torch.cuda.synchronize()
    tb = time.perf_counter() - tb
    nu = sum(s["users"] for s in timed)
    out[other.replace("-", "") + "_shape"] = {"users_per_s": nu / tb, "steps": len(timed), "users_per_step": nu / len(timed),
                                              "mean_prompt_tokens_per_step": float(np.mean([s["cu"][-1] for s in timed])),
                                              "ms_per_step": tb / len(timed) * 1e3}
    # SURVEY.md 8(f) #3, the online single-user caller (demo/inference.py:46-76): one 30-item history -> top 20 (no mask) -> one
    # 460-token prompt -> prefill in LATENCY mode (split-K where the tiles alone would leave most CUs idle) -> verbalizer -> rank
    from llamarec_amd import metrics as M

    on_tokens = 460
    rng_o = np.random.default_rng(0)
    hist_o = rng_o.integers(1, wo["V"] + 1, size=(1, 30)).astype(np.int64)
    prompt_o = [np.concatenate([[1], rng_o.integers(3, 32000, size=on_tokens - 1)]).astype(np.int32)]
    lab_o = np.asarray(label_ids, dtype=np.int32)
    ranker.set_variants(5, 0)

    def online_once():
        t0 = time.perf_counter()
        idx, _ = ro.retrieve_topk(hist_o, 20, exclude_history=False)
        idx.cpu()
        t1 = time.perf_counter()
        M.rank_classes(ranker.prefill_verbalize(prompt_o, lab_o)).cpu()
        return (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3

    for _ in range(3):
        online_once()
    tr_, tp_ = zip(*[online_once() for _ in range(15)])
    ranker.set_variants(0, 0)
    nl = args.layers
    fl_o = (on_tokens * 12.95e9 + on_tokens ** 2 * 2.62e5) * nl / 32
    out["online_shape"] = {"users": 1, "prompt_tokens": on_tokens, "layers": nl, "mode": "latency (gemm variant 5)",
                           "ms_retrieve": float(np.median(tr_)), "ms_prefill": float(np.median(tp_)),
                           "ms_total": float(np.median(tr_) + np.median(tp_)), "ms_total_min": float(min(a + b for a, b in zip(tr_, tp_))),
                           "compute_floor_ms": fl_o / 2.5e15 * 1e3, "weight_stream_floor_ms": 13.5e9 * nl / 32 / 8e12 * 1e3}
    # the ranker's LoRA fine-tuning step (reference micro-batch: 16 prompts, config.py:90-97)
    from llamarec_amd.rank_train import LoraTrainEngine

    eng = LoraTrainEngine(ranker, dropout=0.05, seed=1)
    mb = []
    for s in steps[: min(3, len(steps))]:
        cu = s["cu"]
        ids = s["ids"].cpu().numpy()
        seqs = [ids[cu[i]:cu[i + 1]].copy() for i in range(min(16, len(cu) - 1))]
        for sq in seqs:
            sq[-1] = 2                                    # EOS closes a training sample
        labs = [np.where(np.arange(len(sq)) >= len(sq) - 2, sq, -100) for sq in seqs]
        mb.append((seqs, labs))
    # one workspace for the largest micro-batch of the run, allocated BEFORE the clock starts: round 2 warmed mb[0] only
    # and re-allocated the tens-of-GB workspace inside the timed loop whenever a larger micro-batch arrived (628 ms/step)
    eng.reserve([(sum(len(sq) for sq in seqs), len(seqs), 2 * len(seqs)) for seqs, _ in mb])
    eng.loss_and_grads(*mb[0])
    eng.apply(2e-4, 1.0)
    torch.cuda.synchronize()
    allocs_before = eng.ws_allocations
    tt = time.perf_counter()
    for seqs, labs in mb:
        eng.loss_and_grads(seqs, labs)
        eng.apply(2e-4, 1.0)
    torch.cuda.synchronize()
    tt = time.perf_counter() - tt
    ntok = sum(len(sq) for seqs, _ in mb for sq in seqs)
    out["lora_train_shape"] = {"micro_batch_prompts": len(mb[0][0]), "optimizer_steps": len(mb),
                               "ms_per_step": tt / len(mb) * 1e3, "tokens_per_s": ntok / tt,
                               "samples_per_s": sum(len(sq) for sq, _ in mb) / tt,
                               "tokens_per_step": ntok / len(mb),
                               "workspace_allocations_in_timed_loop": eng.ws_allocations - allocs_before,
                               "workspace_gb": eng._ws.numel() / 1e9,
                               "loss_finite": bool(np.isfinite(float(eng._out[0])))}
    del eng
    # the retriever's training step at the reference's batch shape (config.py:103-111: 64 x 50 for Beauty),
    # synthetic rows, half of them left-padded like short users; fp32, hipGraph replay as train_retriever.py runs it
    from llamarec_amd.train import LRUTrainEngine

    wb = WORKLOADS["beauty"]
    rng = np.random.default_rng(0)
    Bt, Lt = 64, wb["L"]
    seq = rng.integers(1, wb["V"] + 1, size=(Bt, Lt + 1))
    toks, labs = seq[:, :-1].copy(), seq[:, 1:].copy()
    for i, n in enumerate(rng.integers(2, Lt, size=Bt // 2)):
        toks[i, : Lt - n] = 0
        labs[i, : Lt - n - 1] = 0
    te = LRUTrainEngine(init_lru_state_dict(wb["V"], seed=1), seed=3, use_graph=True)
    tt_, tl_ = torch.from_numpy(toks).to(dev), torch.from_numpy(labs).to(dev)
    for _ in range(3):
        te.train_step(tt_, tl_)
    torch.cuda.synchronize()
    n_it = 50
    tr0 = time.perf_counter()
    for _ in range(n_it):
        loss_t = te.train_step(tt_, tl_)
    torch.cuda.synchronize()
    tr = (time.perf_counter() - tr0) / n_it
    out["retriever_train_shape"] = {"workload": "beauty", "batch": Bt, "seq_len": Lt, "num_items": wb["V"], "dtype": "f32",
                                    "optimizer_steps": n_it, "ms_per_step": tr * 1e3, "sequences_per_s": Bt / tr,
                                    "loss_finite": bool(np.isfinite(float(loss_t)))}
    del te
    # the same step in deterministic mode (lr_lru_train_set_deterministic: fixed-point shadows instead of fp32 atomics) -- its cost,
    # and that two engines started from the same state report the same bits after the same steps
    losses, trds = [], []
    for _ in range(2):
        td = LRUTrainEngine(init_lru_state_dict(wb["V"], seed=1), seed=3, use_graph=True).set_deterministic(True)
        for _ in range(3):
            td.train_step(tt_, tl_)
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        for _ in range(n_it):
            loss_d = td.train_step(tt_, tl_)
        torch.cuda.synchronize()
        trds.append((time.perf_counter() - tr0) / n_it)
        losses.append(float(loss_d))
        del td
    # (both engines' times are reported, the faster is the mode's cost: until the LoRA engine's side stream stopped being a lowest-priority
    # stream, the second engine's torch stream here was one of those the runtime then mapped onto that stream's hardware queue and ran the
    # same captured graph 2-3 x slower -- docs/EXPERIMENTS.md, tools/diag/stream_index_probe.py)
    out["retriever_train_shape"].update({"ms_per_step_deterministic": min(trds) * 1e3,
                                         "ms_per_step_deterministic_each_engine": [t * 1e3 for t in trds],
                                         "deterministic_runs_bit_identical": bool(losses[0] == losses[1])})
    return out


if __name__ == "__main__":
    main()
