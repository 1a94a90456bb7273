"""bench.py's one JSON line, end to end on the GPU at a 2-layer depth: every field the driver reads is present and
sane, the roofline and CPU-baseline objects are filled, and the line's own parity block is green."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract():
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--layers", "2", "--no-other-shapes"]
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=900)   # a child process: never exec-replace
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert d["unit"] == "users/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    ro = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s")
    assert 0.0 < ro["frac"] < 1.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9
    pc = ro["power_limited_mfma_ceiling"]          # the measured MFMA-only rate on model-like operands, beside the dense peak
    assert pc["tflops"] < ro["peak"] and abs(pc["frac_of_it"] - ro["achieved"] / pc["tflops"]) < 1e-9 and os.path.exists(
        os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), pc["source"]))
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port")
    assert d["parity"]["ok"] is True and d["metrics"]["retrieve_matches_expected"] is True
    # north_star's second roofline: the item GEMM at the Synth-1M shape rides in the same line
    ri = d["roofline_item_gemm"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "flops_alg", "bytes_alg", "item_ms", "encoder_ms"):
        assert k in ri, k
    assert ri["bound"] in ("hbm", "mfma") and ri["unit"] == "TFLOP/s" and 0.0 < ri["frac"] < 1.0
    assert abs(ri["frac"] - ri["achieved"] / ri["peak"]) < 1e-9
    assert abs(ri["flops_alg"] - 2.0 * 64 * 1000001 * 4096) < 1.0 and abs(ri["bytes_alg"] - (1000001 * 260.0 + 4096 * (256 + 1600 + 400))) < 1.0
    assert abs(ri["achieved"] - ri["flops_alg"] / (ri["item_ms"] * 1e-3) / 1e12) < 1e-6 * ri["achieved"]
    assert 0.0 < ri["item_ms"] < 50.0 and 0.0 < ri["encoder_ms"] < 50.0
    assert ri["checks"]["top50_distinct_sampled"] is True and ri["checks"]["no_history_item_sampled"] is True


def _run_bench(extra, timeout=900):
    cmd = [sys.executable, os.path.join(REPO, "bench.py")] + extra
    return subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_card_over_gloo():
    """`bench.py --gpus 2`: the parent starts its own two torchrun ranks BEFORE any HIP call (nothing is exec-replaced),
    relays rank 0's line and exits with the children's code (spawn_ranks). A 1-GPU box rehearses it with both ranks on
    cuda:0 over gloo (--share-gpu --dist-backend gloo): n_gpus, the all-reduced rank census, both ranks' users in the
    all-reduced histogram and the planted-label metrics are checked. The driver's 8-GPU run is the same code with nccl."""
    r = _run_bench(["--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--layers", "2", "--steps", "2", "--warmup", "1",
                    "--no-other-shapes"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2" and d["config"]["collective_backend"] == "gloo"
    per_rank = d["config"]["users_per_step"] * d["steps"]          # rank 0's users in the timed region
    counted = d["metrics"]["users_counted"]                         # all-reduced over both ranks
    assert counted > per_rank and abs(counted - 2 * per_rank) <= 0.2 * per_rank, (counted, per_rank)
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - counted) < 1e-6 * counted   # value = ALL ranks' users / time
    assert d["metrics"]["retrieve_matches_expected"] is True and d["metrics"]["retrieve_NDCG@10"] > 0
    assert d["roofline"] is not None and 0.0 < d["roofline"]["frac"] < 1.0
    assert d["placement"]["ranks_on_their_own_device"] == 2 and d["placement"]["torch_bytes_on_other_devices"] == 0
    assert "cpu_baseline" not in d and "roofline_item_gemm" not in d     # N = 1 only


@pytest.mark.gpu
def test_bench_four_ranks_on_one_card_over_gloo():
    """The spawn path with more ranks than the 2-rank rehearsal (make the driver's first 8-rank run
    boring): 4 ranks on cuda:0 over gloo, one layer, one step. The pool allows at most 6 processes on a card, so the 8-rank case
    itself is the driver's to run; everything rank-count-dependent -- the port/rank plumbing of spawn_ranks, N random inits,
    the census, the all-reduced histogram -- is the same code at 4. Rank 0's peak device memory rides in the line."""
    r = _run_bench(["--gpus", "4", "--share-gpu", "--dist-backend", "gloo", "--layers", "1", "--steps", "1", "--warmup", "0",
                    "--no-other-shapes"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["ranks_seen"] == 4 and d["steps"] == 1 and d["config"]["parallelism"] == "dp4"
    per_rank = d["config"]["users_per_step"] * d["steps"]
    counted = d["metrics"]["users_counted"]
    assert abs(counted - 4 * per_rank) <= 0.3 * per_rank, (counted, per_rank)
    assert d["metrics"]["retrieve_matches_expected"] is True
    assert 0.0 < d["peak_device_mem_gb_rank0"] < 24.0      # one layer + embeddings + head + the 32 768-row workspace
    assert d["placement"] == {"ranks_on_their_own_device": 4, "torch_bytes_on_other_devices": 0, "share_gpu": True}


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_gpus():
    """--gpus N on a node with fewer than N cards (and no --share-gpu): exit code 2, a message, NO result line -- a
    1-GPU number is never reported as n_gpus = N."""
    import torch

    n = torch.cuda.device_count() + 1
    r = _run_bench(["--gpus", str(n), "--layers", "2", "--steps", "1", "--warmup", "0", "--no-other-shapes"], timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "refusing" in r.stderr


@pytest.mark.gpu
def test_bench_side_fields_carry_the_training_steps():
    """The default run's side measurements (never part of `value`): the other BASELINE shape through the same pipeline, the
    ranker's LoRA step and the retriever's training step. bench.py takes them at full depth only (32 layers), so this is the
    default run with two timed steps and without the CPU and Synth-1M legs."""
    r = _run_bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-item-roofline"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    rt = d["retriever_train_shape"]
    assert rt["workload"] == "beauty" and rt["batch"] == 64 and rt["seq_len"] == 50 and rt["dtype"] == "f32"
    assert rt["loss_finite"] is True and 0.0 < rt["ms_per_step"] < 5.0
    assert abs(rt["sequences_per_s"] - 64 / (rt["ms_per_step"] * 1e-3)) < 1e-6 * rt["sequences_per_s"]
    assert rt["deterministic_runs_bit_identical"] is True and rt["ms_per_step"] * 0.8 < rt["ms_per_step_deterministic"] < 5.0
    lt = d["lora_train_shape"]
    assert lt["loss_finite"] is True and lt["workspace_allocations_in_timed_loop"] == 0 and lt["tokens_per_s"] > 0
    assert d["ml100k_shape"]["users_per_s"] > 0
    on = d["online_shape"]     # SURVEY 8(f) #3 in the driver's line
    assert on["users"] == 1 and on["prompt_tokens"] == 460 and on["layers"] == 32
    assert abs(on["ms_total"] - (on["ms_retrieve"] + on["ms_prefill"])) < 1e-6
    assert 0.0 < on["compute_floor_ms"] < on["ms_prefill"] < 50.0 and 0.0 < on["ms_retrieve"] < 10.0
