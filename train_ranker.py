#!/usr/bin/env python
"""Entry point compatible with the reference's
`python train_ranker.py --llm_retrieved_path experiments/lru/<dataset>`: `trainer.train()` -- LoRA
fine-tuning of q_proj / v_proj on the HIP training step (train_ranker.py:110 of the reference,
SURVEY.md 8(f) #4) -- then `trainer.test(test_retrieval)` (:111). Pass a local base model directory
(--llm_base_model); with --eval_only (and optionally a local PEFT adapter, --llm_adapter_path, merged
into the bf16 weights at load) only the scoring half runs. Nothing is downloaded. --synthetic
fabricates a tiny model + tokenizer. The tuned adapter is written in PEFT's format to
<export_root>/adapter (and the best one by --rerank_best_metric to <export_root>/best_adapter).

Outputs keep the reference layout: experiments/<model>/<dataset>/{subset,overall}_metrics.json.
Data parallel: launch with torch.distributed.run; ranks shard the retrieved users and all-reduce
one int64 histogram.
"""
import os
import pickle
import sys

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def main(argv=None, export_root=None):
    import numpy as np

    from llamarec_amd import config as cfg
    from llamarec_amd import data as D
    from llamarec_amd.llm import LlamaRanker
    from llamarec_amd import dist as DD
    from llamarec_amd.rerank import LazyEvalItems, LLMEvaluator, build_val_items
    from llamarec_amd.verb import ManualVerbalizer

    args = cfg.parse(argv, model_code="llm")
    if not args.llm_retrieved_path:
        raise SystemExit("--llm_retrieved_path experiments/lru/<dataset> is required")
    export_root = export_root or args.export_root or os.path.join(
        cfg.EXPERIMENT_ROOT, args.llm_base_model.rstrip("/").split("/")[-1], args.dataset_code)
    retrieved = pickle.load(open(os.path.join(args.llm_retrieved_path, "retrieved.pkl"), "rb"))
    if args.share_gpu:                      # rehearsal of several ranks on one card (gloo): everybody on cuda:0
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = DD.init_from_env(args.dist_backend)
    import torch

    # one replica per GPU (train_ranker.py:46-47 of the reference: device_map = {"": process_index}): the model, the
    # workspace, the flat LoRA gradient buffer and the histograms handed to the all-reduce all live on cuda:<local rank>
    device = f"cuda:{local}"
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if args.synthetic:
        from llamarec_amd.synth import FakeTokenizer, synth_llama_state

        dataset = D.synthetic_dataset(num_users=300, num_items=1000, seed=args.seed)
        tokenizer = FakeTokenizer()
        c = dict(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                 num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=2048,
                 rms_norm_eps=1e-5, rope_theta=10000.0)
        from llamarec_amd.llm import load_peft_adapter

        model = LlamaRanker.from_state_dict(synth_llama_state(c, args.seed), c, device=device,
                                            lora=load_peft_adapter(args.llm_adapter_path) if args.llm_adapter_path
                                            else None, nf4=args.llm_load_in_4bit)
    else:
        from transformers import AutoTokenizer

        dataset = D.load_dataset_pkl(D.preprocessed_path(args.data_root, args.dataset_code, args.min_rating,
                                                         args.min_uc, args.min_sc))
        tokenizer = AutoTokenizer.from_pretrained(args.llm_base_tokenizer, local_files_only=True)
        if tokenizer.pad_token is None:  # dataloader/llm.py:122-126
            tokenizer.pad_token = tokenizer.unk_token
        tokenizer.padding_side = "left"
        tokenizer.truncation_side = "left"
        tokenizer.clean_up_tokenization_spaces = True
        model = LlamaRanker.from_pretrained(args.llm_base_model, device=device, adapter_path=args.llm_adapter_path,
                                            load_in_4bit=args.llm_load_in_4bit)
    ncls = args.llm_negative_sample_size + 1
    args.num_items = len(dataset["smap"])
    verbalizer = ManualVerbalizer(tokenizer=tokenizer, prefix="", post_log_softmax=False, classes=list(range(ncls)),
                                  label_words={i: chr(ord("A") + i) for i in range(ncls)})
    if not args.eval_only:
        if args.llm_adapter_path:
            raise SystemExit("--llm_adapter_path is merged at load: combine it with --eval_only, or train from the base")
        from llamarec_amd.rank_train import LLMTrainSamples, LoraRankerTrainer, LoraTrainEngine

        engine = LoraTrainEngine(model, r=args.lora_r, alpha=args.lora_alpha, dropout=args.lora_dropout, seed=args.seed)
        samples = LLMTrainSamples(args, dataset["train"], dataset["meta"], tokenizer,
                                  rng=np.random.RandomState(args.seed + rank))
        val_items = build_val_items(dataset, retrieved, tokenizer, args)
        trainer = LoraRankerTrainer(args, engine, samples, val_items, verbalizer, export_root, rank, world,
                                    log=print if rank == 0 else (lambda *a, **k: None))
        steps = trainer.train()
        print(f"LoRA fine-tuning: {steps} optimizer steps, best {args.rerank_best_metric} = {trainer.best_metric}")
        model = engine.merge_into_base_()       # the scoring path below now serves base + tuned adapter
    # lazily built test prompts: users are sharded over the ranks first, each rank tokenises only its own shard, in a
    # producer thread ahead of its GPU loop (llamarec_amd/rerank.py, LazyEvalItems); --test_batch_size caps the prompts
    # per prefill only when given explicitly (the evaluation batches by token budget, --eval_token_budget)
    items = LazyEvalItems(dataset, retrieved, tokenizer, args, split="test")
    ev = LLMEvaluator(args, model, items, verbalizer, export_root)
    metrics = ev.test(retrieved["test_retrieval"])
    print("Ranking Performance on Subset:", metrics)
    print("Overall Performance of Our Framework:", ev.overall_metrics)
    return metrics, ev.overall_metrics


if __name__ == "__main__":
    main()
