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


def _train_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from llamarec_amd import data as D
    from llamarec_amd import dist as DD
    from llamarec_amd.train import average_gradients_

    r, w, _ = DD.init_from_env(backend="gloo")
    # every rank draws the same permutation and takes its contiguous share of each global batch
    seqs = [[(u + i) % 50 + 1 for i in range(3 + u % 9)] for u in range(37)]
    mine = [(t.copy(), l.copy()) for t, l in D.train_batches(seqs, 4, 8, np.random.default_rng(5), r, w)]
    # "gradients": rank-dependent flat buffer -> mean over ranks after ONE all-reduce
    g = torch.arange(10, dtype=torch.float32) * (r + 1)
    average_gradients_(g)
    DD.barrier()
    q.put((r, mine, g.numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_training_shards_batches_and_averages_gradients():
    """Retriever training under data parallelism (train_retriever.py under torchrun): batch sharding and the
    one-buffer gradient average, on CPU over gloo."""
    sys.path.insert(0, REPO)
    from llamarec_amd import data as D

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (m, g)) for r, m, g in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.allclose(got[0][1], np.arange(10) * 1.5) and np.allclose(got[1][1], got[0][1])
    seqs = [[(u + i) % 50 + 1 for i in range(3 + u % 9)] for u in range(37)]
    single = list(D.train_batches(seqs, 8, 8, np.random.default_rng(5)))      # one rank, global batch 8
    assert len(got[0][0]) == len(got[1][0]) == len(single)
    for (t0, l0), (t1, l1), (ts, ls) in zip(got[0][0], got[1][0], single):
        if len(ts) == 8:
            assert np.array_equal(np.concatenate([t0, t1]), ts) and np.array_equal(np.concatenate([l0, l1]), ls)
    # LRUTrainDataset semantics: a short sequence has ONE labelled pad position
    t, l = D.lru_train_batch([[5, 6, 7]], 8)
    assert t.tolist() == [[0, 0, 0, 0, 0, 0, 5, 6]] and l.tolist() == [[0, 0, 0, 0, 0, 5, 6, 7]]
    t, l = D.lru_train_batch([list(range(1, 12))], 8)
    assert t.tolist() == [[3, 4, 5, 6, 7, 8, 9, 10]] and l.tolist() == [[4, 5, 6, 7, 8, 9, 10, 11]]
    # sliding windows from the end (dataloader/lru.py:103-110)
    ds = {"train": {1: list(range(1, 30)), 2: [1, 2, 3]}}
    assert D.lru_train_sequences(ds, 8, 1.0) == [list(range(22, 30)), list(range(14, 22)), list(range(6, 14)), [1, 2, 3]]
