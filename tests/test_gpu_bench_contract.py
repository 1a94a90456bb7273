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
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port")
    assert d["parity"]["ok"] is True and d["metrics"]["retrieve_matches_expected"] is True
