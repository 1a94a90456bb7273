"""Flags of the scoring path and of retriever training, with the reference's names and defaults
(config.py:151-274 and the per-dataset defaults of set_template, :12-148). Parsed only when an entry point
asks (the reference parses sys.argv at import, config.py:274, and may block on input(), :13-27 -- neither
is reproduced).
"""
from __future__ import annotations

import argparse

from .prompt import DEFAULT_INPUT_TEMPLATE, DEFAULT_SYSTEM_TEMPLATE

EXPERIMENT_ROOT = "experiments"   # config.py:5-9
STATE_DICT_KEY = "model_state_dict"
RAW_DATASET_ROOT_FOLDER = "data"


def build_parser():
    p = argparse.ArgumentParser(description="llamarec-mi355x scoring path")
    p.add_argument("--dataset_code", type=str, default=None)
    p.add_argument("--model_code", type=str, default=None)
    p.add_argument("--min_rating", type=int, default=0)
    p.add_argument("--min_uc", type=int, default=5)
    p.add_argument("--min_sc", type=int, default=5)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--val_batch_size", type=int, default=None)
    p.add_argument("--test_batch_size", type=int, default=None)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--metric_ks", nargs="+", type=int, default=None)
    p.add_argument("--rerank_metric_ks", nargs="+", type=int, default=None)
    p.add_argument("--bert_max_len", type=int, default=None)
    p.add_argument("--bert_hidden_units", type=int, default=64)
    p.add_argument("--bert_num_blocks", type=int, default=None)
    p.add_argument("--llm_base_model", type=str, default="meta-llama/Llama-2-7b-hf")
    p.add_argument("--llm_base_tokenizer", type=str, default="meta-llama/Llama-2-7b-hf")
    p.add_argument("--llm_max_title_len", type=int, default=32)
    p.add_argument("--llm_max_text_len", type=int, default=1536)
    p.add_argument("--llm_max_history", type=int, default=20)
    p.add_argument("--llm_negative_sample_size", type=int, default=19)
    p.add_argument("--llm_system_template", type=str, default=DEFAULT_SYSTEM_TEMPLATE)   # config.py:242-249
    p.add_argument("--llm_input_template", type=str, default=DEFAULT_INPUT_TEMPLATE)
    p.add_argument("--llm_retrieved_path", type=str, default=None)
    # config.py:250 (the reference's `type=bool` makes any non-empty string True; here "false" / "0" / "no" turn it off)
    p.add_argument("--llm_load_in_4bit", type=lambda v: str(v).lower() not in ("false", "0", "no", ""), default=True)
    p.add_argument("--lora_r", type=int, default=8)
    p.add_argument("--lora_alpha", type=int, default=32)
    # ranker LoRA fine-tuning (config.py:203,236-241,257-269; defaults filled in set_template like config.py:81-102)
    p.add_argument("--lora_dropout", type=float, default=0.05)
    p.add_argument("--lora_num_epochs", type=int, default=1)
    p.add_argument("--lora_val_iterations", type=int, default=None)
    p.add_argument("--lora_val_delay", type=int, default=None)
    p.add_argument("--lora_early_stopping_patience", type=int, default=None)
    p.add_argument("--lora_max_steps", type=int, default=None)
    p.add_argument("--lora_lr", type=float, default=2e-4)
    p.add_argument("--lora_micro_batch_size", type=int, default=None)
    p.add_argument("--llm_train_on_inputs", action="store_true")
    p.add_argument("--rerank_best_metric", type=str, default=None)
    p.add_argument("--lora_max_val_samples", type=int, default=None, help="validate on the first N retrieved users")
    # additions of this implementation (local assets only; nothing is downloaded)
    p.add_argument("--data_root", type=str, default=RAW_DATASET_ROOT_FOLDER)
    p.add_argument("--export_root", type=str, default=None)
    p.add_argument("--llm_adapter_path", type=str, default=None, help="local PEFT adapter directory")
    p.add_argument("--lora_token_budget", type=int, default=16384, help="tokens per forward/backward pass when an "
                   "optimizer step's micro-batches are regrouped (same summed gradient; 0 = the reference's micro-batches)")
    p.add_argument("--eval_token_budget", type=int, default=None, help="prompt tokens per evaluation prefill (default "
                   "packing.TOKEN_BUDGET; 0 = fixed batches of --test_batch_size prompts like the reference's loader)")
    p.add_argument("--synthetic", action="store_true", help="fabricate dataset / weights (nothing exists offline)")
    # retriever training (config.py:166-202,216-217; defaults filled in set_template like config.py:103-134)
    p.add_argument("--train_batch_size", type=int, default=None)
    p.add_argument("--sliding_window_size", type=float, default=1.0)
    p.add_argument("--num_epochs", type=int, default=500)
    p.add_argument("--lr", type=float, default=None)
    p.add_argument("--weight_decay", type=float, default=None)
    p.add_argument("--adam_epsilon", type=float, default=1e-9)
    p.add_argument("--max_grad_norm", type=float, default=5.0)
    p.add_argument("--enable_lr_schedule", action="store_true")
    p.add_argument("--enable_lr_warmup", action="store_true")
    p.add_argument("--warmup_steps", type=int, default=None)
    p.add_argument("--decay_step", type=int, default=None)
    p.add_argument("--gamma", type=float, default=1.0)
    p.add_argument("--val_strategy", type=str, default=None, choices=["epoch", "iteration"])
    p.add_argument("--val_iterations", type=int, default=500)
    p.add_argument("--early_stopping_patience", type=int, default=None)
    p.add_argument("--best_metric", type=str, default=None)
    p.add_argument("--bert_dropout", type=float, default=0.2)
    p.add_argument("--bert_attn_dropout", type=float, default=0.2)
    p.add_argument("--eval_only", action="store_true", help="skip training (a checkpoint must exist)")
    p.add_argument("--dist_backend", type=str, default=None, help="torch.distributed backend under torchrun (default nccl = RCCL)")
    p.add_argument("--share_gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (with --dist_backend gloo)")
    p.add_argument("--max_train_iterations", type=int, default=None, help="stop after this many optimizer steps")
    p.add_argument("--deterministic", action="store_true", help="retriever training: run-to-run identical bits (fixed-point "
                   "accumulation instead of fp32 atomics; ~0.1 ms per step on Beauty)")
    return p


def set_template(args):
    """Dataset / model dependent defaults (config.py:57-61,98,103-111,136-146)."""
    ml = args.dataset_code == "ml-100k"
    # the ranker's evaluation batches by TOKEN budget by default (packing.py); an explicit --test_batch_size still caps the
    # prompts per prefill (rerank.LLMEvaluator), a template default does not
    args.test_batch_size_explicit = getattr(args, "test_batch_size", None) is not None
    if args.bert_max_len is None:
        args.bert_max_len = 200 if ml else 50
    if args.bert_num_blocks is None:
        args.bert_num_blocks = 2
    if args.train_batch_size is None:                      # config.py:90-107
        args.train_batch_size = (32 if ml else 16) if args.model_code == "llm" else (16 if ml else 64)
    if args.lr is None:                                    # config.py:121-124
        args.lr = 1e-3
    if args.weight_decay is None:
        args.weight_decay = 1e-2
    if args.val_strategy is None:                          # config.py:76-79
        args.val_strategy = "iteration"
    if args.early_stopping_patience is None:
        args.early_stopping_patience = 20
    if args.best_metric is None:                           # config.py:140-141
        args.best_metric = "Recall@10"
    if args.metric_ks is None:
        args.metric_ks = [1, 5, 10, 20, 50]
    if args.rerank_metric_ks is None:
        args.rerank_metric_ks = [1, 5, 10]
    if args.warmup_steps is None:                          # config.py:133-134
        args.warmup_steps = 100
    if args.lora_val_iterations is None:                   # config.py:81-88
        args.lora_val_iterations = 100
    if args.lora_val_delay is None:
        args.lora_val_delay = 0
    if args.lora_early_stopping_patience is None:
        args.lora_early_stopping_patience = 20
    if args.lora_max_steps is None:
        args.lora_max_steps = -1
    if args.rerank_best_metric is None:                    # config.py:142-143
        args.rerank_best_metric = "NDCG@10"
    if args.model_code == "llm":
        if args.lora_micro_batch_size is None:             # config.py:90-97
            args.lora_micro_batch_size = 8 if args.dataset_code == "beauty" else 16
        if args.test_batch_size is None:
            args.test_batch_size = 32 if ml else 16
        if args.val_batch_size is None:
            args.val_batch_size = args.test_batch_size
    else:
        if args.test_batch_size is None:
            args.test_batch_size = 16 if ml else 64
        if args.val_batch_size is None:
            args.val_batch_size = args.test_batch_size
    return args


def parse(argv=None, model_code=None):
    args, _unknown = build_parser().parse_known_args(argv)
    if model_code:
        args.model_code = model_code
    if args.dataset_code is None:
        raise SystemExit("--dataset_code is required (ml-100k | beauty | games | ...)")
    return set_template(args)
