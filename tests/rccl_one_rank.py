"""Child process of tests/test_gpu_rccl.py: ONE rank, backend "nccl" (= RCCL on ROCm). Initialises the process group the
way bench.py / train_ranker.py do under torchrun and pushes the job's real exchanges through RCCL:
  * TwoStagePipeline.finish(): all-reduce(sum) of int64 [51 + 21 + 1] rank histograms + user count;
  * bench.py's rank census (int64 ones) and its max-over-ranks float64 clock;
  * the trainers' flat fp32 gradient buffer (train.average_gradients_).
Prints one JSON line; exit code 0 only if every result is what a 1-rank sum / max must give."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    from llamarec_amd import dist as D
    from llamarec_amd.llm import LlamaRanker, pack_prompts
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from llamarec_amd.pipeline import TwoStagePipeline
    from llamarec_amd.synth import synth_llama_state
    from llamarec_amd.train import average_gradients_

    rank, world, local = D.init_from_env("nccl", force=True)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(local)
    ones = torch.ones(1, dtype=torch.int64, device=dev)
    D.all_reduce_sum_(ones)
    clock = D.all_reduce_max_float(1.25, device=dev)
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    grads = torch.arange(4096, dtype=torch.float32, device=dev)
    average_gradients_(grads)
    D.barrier()

    rng = np.random.default_rng(0)
    V, B, L = 300, 40, 12
    ids = np.zeros((B, L), np.int64)
    for b in range(B):
        n = int(rng.integers(1, L + 1))
        ids[b, L - n:] = rng.choice(V, size=n, replace=False) + 1
    retriever = LRURec.from_state_dict(init_lru_state_dict(V, seed=1), device=dev)
    cfg = dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2,
               num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-5, rope_theta=10000.0)
    ranker = LlamaRanker.from_state_dict(synth_llama_state(cfg, 5), cfg, device=dev)
    pipe = TwoStagePipeline(retriever, ranker, list(range(40, 60)), device=dev)
    top, _ = retriever.retrieve_topk(torch.from_numpy(ids).to(dev), 50, True)
    labels = top[torch.arange(B), torch.arange(B) % 50].to(torch.int64)       # planted: user j's answer sits at rank j mod 50
    seqs = [np.concatenate([[1], rng.integers(3, 320, size=int(n))]) for n in rng.integers(4, 60, size=B)]
    pids, cu = pack_prompts(seqs)
    pipe.step(torch.from_numpy(ids).to(dev), labels, torch.from_numpy(pids).to(dev), torch.from_numpy(cu).to(dev), cu)
    before = pipe.hist_retrieve.clone()
    retr, rer, n = pipe.finish()                                               # the job's only collective, over RCCL
    torch.cuda.synchronize()
    ok = (int(ones.item()) == 1 and clock == 1.25 and float(t.item()) == 1.25 and n == B
          and torch.equal(grads, torch.arange(4096, dtype=torch.float32, device=dev))
          and int(before.sum()) == B and int(before[:50].sum()) == B and retr["Recall@50"] == 1.0 and retr["NDCG@10"] > 0)
    print(json.dumps({"rccl_one_rank": "ok" if ok else "MISMATCH", "backend": dist.get_backend(), "users": n,
                      "retrieve_NDCG@10": retr["NDCG@10"], "rerank_NDCG@10": rer["NDCG@10"]}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
