#!/usr/bin/env python
"""Entry point compatible with the reference's `python train_retriever.py` for the SCORING half:
`trainer.test()` and `trainer.generate_candidates(...)` (train_retriever.py:77,80 of the reference).
Training itself (:76) is outside this implementation's scope: the checkpoint
experiments/lru/<dataset>/models/best_acc_model.pth must already exist, or pass --synthetic.

Outputs keep the reference layout: experiments/lru/<dataset>/{test_metrics.json, retrieved.pkl}.
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
    if os.path.exists(ckpt):
        model = LRURec.from_checkpoint(ckpt)
    elif args.synthetic:
        model = LRURec.from_state_dict(init_lru_state_dict(args.num_items, args.seed, args.bert_num_blocks))
    else:
        raise SystemExit(f"{ckpt} not found: train the retriever with the reference first (or use --synthetic)")
    _, v_ids, v_lab = D.lru_eval_arrays(dataset, "val", args.bert_max_len)
    _, t_ids, t_lab = D.lru_eval_arrays(dataset, "test", args.bert_max_len)
    ev = LRUEvaluator(args, model, list(D.batches(v_ids, v_lab, args.val_batch_size)),
                      list(D.batches(t_ids, t_lab, args.test_batch_size)), export_root)
    print("******************** Testing Metrics ********************")
    print(ev.test())
    out = ev.generate_candidates(os.path.join(export_root, "retrieved.pkl"))
    print(out["test_metrics"])
    return out


if __name__ == "__main__":
    main()
