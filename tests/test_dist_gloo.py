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


class _FakeLoraEngine:
    """CPU stand-in with the engine's surface (the HIP step needs a GPU): gradient = mean over the micro-batch of a
    per-prompt vector, so the trainer's sharding / accumulation / averaging arithmetic can be checked exactly."""

    def __init__(self):
        self.params = torch.zeros(4, dtype=torch.float64)
        self.grads = torch.zeros(4, dtype=torch.float64)
        self.device = torch.device("cpu")
        self.seen = []

    def loss_and_grads(self, seqs, labels, grad_scale=1.0, accumulate=False):
        v = torch.tensor([[float(s.sum()), float(len(s)), float(s[1]), 1.0] for s in seqs], dtype=torch.float64).mean(0)
        self.grads = self.grads + grad_scale * v if accumulate else grad_scale * v
        self.seen.append([int(s[1]) for s in seqs])
        return torch.tensor(float(v[0]))

    def apply(self, lr, max_grad_norm=1.0):
        self.params = self.params - lr * self.grads
        return torch.tensor(0.0)


class _FakeSamples:
    tokenizer = type("T", (), {"eos_token_id": 2})()

    def __len__(self):
        return 64

    def __getitem__(self, i):
        ids = [1, 100 + i] + [7] * (3 + i % 5) + [30, 2]
        return {"input_ids": ids, "attention_mask": [1] * len(ids), "labels": [-100] * (len(ids) - 2) + ids[-2:]}


def _lora_args(micro, batch, token_budget=0):
    """token_budget 0: the reference's micro-batches as they come (what the grouping assertions below describe)."""
    from types import SimpleNamespace

    return SimpleNamespace(lora_micro_batch_size=micro, train_batch_size=batch, lora_max_steps=5, lora_num_epochs=1,
                           warmup_steps=2, lora_lr=0.1, lora_val_iterations=100, lora_val_delay=0,
                           lora_early_stopping_patience=20, rerank_best_metric="NDCG@10", seed=3, llm_max_text_len=64,
                           lora_token_budget=token_budget)


def _lora_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from llamarec_amd import dist as DD
    from llamarec_amd.rank_train import LoraRankerTrainer

    r, w, _ = DD.init_from_env(backend="gloo")
    eng = _FakeLoraEngine()
    # per rank: 2 micro-batches of 4 per optimizer step; 2 ranks -> 16 samples per step
    tr = LoraRankerTrainer(_lora_args(4, 8), eng, _FakeSamples(), [], None, None, r, w, log=lambda *a: None)
    steps = tr.train()
    DD.barrier()
    q.put((r, steps, eng.seen, eng.params.numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_lora_training_shards_microbatches_and_averages_gradients():
    """Ranker LoRA fine-tuning under data parallelism (train_ranker.py under torchrun): 2 ranks x 2 micro-batches must
    walk the samples, and reach the parameters, of 1 rank x 4 micro-batches."""
    sys.path.insert(0, REPO)
    from llamarec_amd.rank_train import LoraRankerTrainer

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lora_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (s, seen, prm)) for r, s, seen, prm in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _FakeLoraEngine()
    tr = LoraRankerTrainer(_lora_args(4, 16), single, _FakeSamples(), [], None, None, 0, 1, log=lambda *a: None)
    assert tr.train() == got[0][0] == got[1][0] == 5
    assert np.allclose(got[0][2], got[1][2]) and np.allclose(got[0][2], single.params.numpy())
    # micro-batch j of a step on one rank = micro-batch j // 2 of rank j % 2 on two
    for step in range(5):
        for j in range(4):
            assert single.seen[4 * step + j] == got[j % 2][1][2 * step + j // 2]
    flat = [i for mb in single.seen[:4] for i in mb]
    assert len(set(flat)) == 16                                   # a step's samples are distinct


def test_token_budget_repacking_of_an_optimizer_step_keeps_the_gradient():
    """LoraRankerTrainer.repack: an optimizer step's micro-batches regrouped into token-budget passes reach the same
    parameters (every labelled token keeps its weight 1 / (accum * m)) in fewer, larger passes; a budget too small for
    two samples degenerates to one sample per pass and still agrees; unequal label counts keep the reference's grouping."""
    sys.path.insert(0, REPO)
    from llamarec_amd.rank_train import LoraRankerTrainer

    ref = _FakeLoraEngine()
    LoraRankerTrainer(_lora_args(4, 16, 0), ref, _FakeSamples(), [], None, None, 0, 1, log=lambda *a: None).train()
    assert len(ref.seen) == 5 * 4
    for budget, passes in ((16384, 5 * 1), (24, None), (9, 5 * 16)):
        eng = _FakeLoraEngine()
        tr = LoraRankerTrainer(_lora_args(4, 16, budget), eng, _FakeSamples(), [], None, None, 0, 1, log=lambda *a: None)
        assert tr.train() == 5
        assert np.allclose(eng.params.numpy(), ref.params.numpy(), rtol=0, atol=1e-12), budget
        if passes is not None:
            assert len(eng.seen) == passes, (budget, len(eng.seen))
        else:
            assert 5 * 4 < len(eng.seen) < 5 * 16
        assert sorted(i for mb in eng.seen[: len(eng.seen) // 5] for i in mb) == sorted(i for mb in ref.seen[:4] for i in mb)
    # unequal label counts between the micro-batches (train_on_inputs-like): the reference's grouping is kept
    tr = LoraRankerTrainer(_lora_args(2, 4, 16384), _FakeLoraEngine(), _FakeSamples(), [], None, None, 0, 1, log=lambda *a: None)
    a = ([np.array([1, 5, 6, 2])] * 2, [np.array([-100, -100, 6, 2])] * 2)
    b = ([np.array([1, 5, 6, 2])] * 2, [np.array([-100, 5, 6, 2])] * 2)
    out = tr.repack([a, b])
    assert len(out) == 2 and out[0][2] == out[1][2] == 0.5
    out = tr.repack([a, a])
    assert len(out) == 1 and out[0][2] == 1.0 and len(out[0][0]) == 4
