"""The `nccl` (= RCCL) backend on the GPU box. A one-GPU box cannot run two ranks on two cards, but it can prove what a
first 8-GPU run otherwise discovers: that RCCL loads, that a process group initialises from torchrun's environment,
and that the collectives this code base issues (int64 sum of the rank histograms, float64 max, fp32 sum of a flat
gradient buffer, barrier) are supported dtypes / ops -- through the product's own call sites (TwoStagePipeline.finish,
dist.*, train.average_gradients_). Runs in a child process so a communicator problem cannot take the test session down."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_rccl_group_carries_the_histogram_all_reduce():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(REPO, "tests", "rccl_one_rank.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"rccl_one_rank"')]
    assert p.returncode == 0 and lines, (p.stdout[-2000:], p.stderr[-3000:])
    out = json.loads(lines[-1])
    assert out["rccl_one_rank"] == "ok" and out["backend"] == "nccl" and out["users"] == 40
    assert out["retrieve_NDCG@10"] > 0
