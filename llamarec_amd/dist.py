"""Data-parallel sharding of users over the GPUs of one node (one process per GPU).

Users are independent in both stages; weights are replicated; the only exchange
is one all-reduce (sum) of the int64 rank histograms at the end -- RCCL over xGMI on GPUs
(torch.distributed backend "nccl"), gloo in the CPU tests. The reference instead all-gathers
[B, 32000] fp32 logits every eval step through HF Trainer / accelerate (trainer/llm.py:122,127).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from torchrun's environment (no-op for a single process unless `force`: a 1-rank
    group still loads RCCL and runs its collectives -- tests/test_gpu_rccl.py rehearses the job's exchange that way)."""
    rank, world, local = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_users: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`: keeps the reference's positional user ids
    (user_id = running index + 1, trainer/lru.py:85,127)."""
    return (rank * n_users) // world, ((rank + 1) * n_users) // world


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized():   # a 1-rank group too: the collective is then RCCL's no-op path
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_reduce_max_float(x: float, device=None) -> float:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return x


def broadcast_shard_edges(edges, device=None):
    """Every rank scores the shard RANK 0 cut: [(lo, hi)] * world as an int64 tensor broadcast from rank 0, so that the ranks
    agree on the edges by construction (the lazy evaluation path derives them from a least-squares estimate of the prompt
    lengths, which two nodes could round differently) -- and a disagreement is impossible BEFORE the LLM evaluation runs, not
    discovered after it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return edges
    t = torch.tensor(edges, dtype=torch.int64, device=device).reshape(-1)
    dist.broadcast(t, src=0)
    t = t.cpu().reshape(-1, 2)
    return [(int(a), int(b)) for a, b in t]


def device_placement(local: int):
    """Where this rank's memory went: (1 if HIP's current device is cuda:`local` else 0, bytes the torch allocator holds on any
    OTHER visible device). Every allocation of the job names its device (the constructors take `device=`, the C library allocates
    on HIP's current device, which init_from_env / bench.main set to LOCAL_RANK), so a healthy N-rank run reports (1, 0) on every
    rank; bench.py sums both over the ranks into its line. Querying the allocator's statistics creates no context on the other
    devices. The reference leaves placement to accelerate (train_ranker.py:46-47)."""
    cur = torch.cuda.current_device()
    elsewhere = 0
    for d in range(torch.cuda.device_count()):
        if d != local:
            try:
                elsewhere += int(torch.cuda.memory_allocated(d))
            except Exception:   # a device this process may not query: nothing of ours can be there
                pass
    return int(cur == local), elsewhere
