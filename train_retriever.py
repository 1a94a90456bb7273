#!/usr/bin/env python
"""Entry point compatible with the reference's `python train_retriever.py` (its :76-80): `trainer.train()`,
`trainer.test()` and `trainer.generate_candidates(...)`, all on the MI355X kernels (training:
llamarec_amd/train.py over csrc/lru_train.hip; scoring: llamarec_amd/{lru,retrieve}.py). `--eval_only` skips
training and needs experiments/lru/<dataset>/models/best_acc_model.pth (as written here or by the reference).
Under torchrun every rank trains on its share of each global batch and gradients are averaged with one
all-reduce per step.

Outputs keep the reference layout: experiments/lru/<dataset>/{models/best_acc_model.pth, test_metrics.json,
retrieved.pkl}.
"""
import os
import sys

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def main(argv=None, export_root=None):
    from llamarec_amd import config as cfg
    from llamarec_amd import data as D
    from llamarec_amd.lru import LRURec, init_lru_state_dict
    from llamarec_amd.retrieve import LRUEvaluator

    args = cfg.parse(argv, model_code="lru")
    export_root = export_root or args.export_root or os.path.join(cfg.EXPERIMENT_ROOT, args.model_code, args.dataset_code)
    if args.synthetic:
        dataset = D.synthetic_dataset(num_users=300, num_items=1000, seed=args.seed)
    else:
        dataset = D.load_dataset_pkl(D.preprocessed_path(args.data_root, args.dataset_code, args.min_rating,
                                                         args.min_uc, args.min_sc))
    args.num_users, args.num_items = len(dataset["umap"]), len(dataset["smap"])
    ckpt = os.path.join(export_root, "models", "best_acc_model.pth")
    _, v_ids, v_lab = D.lru_eval_arrays(dataset, "val", args.bert_max_len)
    _, t_ids, t_lab = D.lru_eval_arrays(dataset, "test", args.bert_max_len)
    if not args.eval_only:
        from llamarec_amd import dist as DD
        from llamarec_amd.train import LRUTrainer

        if args.share_gpu:
            os.environ["LOCAL_RANK"] = "0"
        rank, world, local = DD.init_from_env(args.dist_backend)
        trainer = LRUTrainer(args, device=f"cuda:{local}", export_root=export_root, rank=rank, world=world)
        losses = trainer.train(D.lru_train_sequences(dataset, args.bert_max_len, args.sliding_window_size),
                               list(D.batches(v_ids, v_lab, args.val_batch_size)))
        if rank == 0:
            print(f"trained {trainer.iterations} iterations; epoch losses {[round(x, 4) for x in losses[:3]]} ... "
                  f"{[round(x, 4) for x in losses[-2:]]}; best {args.best_metric} {trainer.best_metric:.4f}")
        if world > 1:  # replicas must have stayed identical: same averaged gradients, same updates
            import torch

            p = trainer.engine.params.detach().clone()
            ref = p.clone()
            torch.distributed.broadcast(ref, src=0)
            print(f"rank {rank}: max |param - rank0 param| = {float((p - ref).abs().max()):.3e}")
        DD.barrier()
        if rank != 0:
            return None
    if os.path.exists(ckpt):
        model = LRURec.from_checkpoint(ckpt)
    elif args.synthetic:
        model = LRURec.from_state_dict(init_lru_state_dict(args.num_items, args.seed, args.bert_num_blocks))
    else:
        raise SystemExit(f"{ckpt} not found: train first (drop --eval_only) or use --synthetic")
    ev = LRUEvaluator(args, model, list(D.batches(v_ids, v_lab, args.val_batch_size)),
                      list(D.batches(t_ids, t_lab, args.test_batch_size)), export_root)
    print("******************** Testing Metrics ********************")
    print(ev.test())
    out = ev.generate_candidates(os.path.join(export_root, "retrieved.pkl"))
    print(out["test_metrics"])
    return out


if __name__ == "__main__":
    main()
