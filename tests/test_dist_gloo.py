"""CPU, world_size 2 over gloo: user sharding keeps contiguous blocks and the histogram all-reduce
reproduces the single-rank metrics exactly (integer counts) -- the N > 1 path of bench.py /
pipeline.finish() without a GPU."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _positions_hist(ranked, labels, K):
    hit = ranked == labels[:, None]
    pos = np.where(hit.any(1), hit.argmax(1), K)
    return np.bincount(pos, minlength=K + 1).astype(np.int64)


def _worker(rank, world, port, ranked, labels, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from llamarec_amd import dist as D
    from llamarec_amd import metrics as M

    r, w, _ = D.init_from_env(backend="gloo")
    lo, hi = D.shard_range(len(labels), r, w)
    K = ranked.shape[1]
    hist = torch.from_numpy(_positions_hist(ranked[lo:hi], labels[lo:hi], K))
    n = torch.tensor([hi - lo], dtype=torch.int64)
    packed = torch.cat([hist, n])
    D.all_reduce_sum_(packed)
    t = D.all_reduce_max_float(1.0 + r)
    D.barrier()
    if r == 0:
        q.put((packed.numpy().copy(), t, (lo, hi)))
    dist.destroy_process_group()


def test_two_rank_histogram_allreduce_matches_single_rank():
    rng = np.random.default_rng(0)
    U, K = 101, 50
    ranked = np.stack([rng.permutation(400)[:K] for _ in range(U)]).astype(np.int64)
    labels = np.array([ranked[u, rng.integers(0, K)] if rng.random() < 0.7 else 999 for u in range(U)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ranked, labels, q)) for r in range(2)]
    for p in procs:
        p.start()
    packed, tmax, (lo, hi) = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = _positions_hist(ranked, labels, K)
    assert np.array_equal(packed[:-1], full) and packed[-1] == U
    assert tmax == 2.0 and (lo, hi) == (0, 50)
    sys.path.insert(0, REPO)
    from llamarec_amd import metrics as M
    from oracle import lru_oracle as O

    sums = M.metric_sums_from_histogram(packed[:-1], [1, 5, 10, 20, 50])
    assert np.allclose(sums, O.rank_metric_sums(ranked.astype(np.int32), labels, [1, 5, 10, 20, 50]), atol=1e-12)


def test_shard_ranges_cover_users_contiguously():
    from llamarec_amd.dist import shard_range

    for U in (0, 1, 7, 610, 22332):
        for W in (1, 2, 3, 8):
            edges = [shard_range(U, r, W) for r in range(W)]
            assert edges[0][0] == 0 and edges[-1][1] == U
            assert all(edges[i][1] == edges[i + 1][0] for i in range(W - 1))
            assert max(h - l for l, h in edges) - min(h - l for l, h in edges) <= 1
