#!/usr/bin/env python
"""Entry point compatible with the reference's
`python train_ranker.py --llm_retrieved_path experiments/lru/<dataset>` for the SCORING half:
`trainer.test(test_retrieval)` (train_ranker.py:111 of the reference). LoRA fine-tuning (:110) is
outside this implementation's scope: pass a local base model directory
(--llm_base_model) and, if trained, a local PEFT adapter (--llm_adapter_path); the adapter is merged
into bf16 weights at load. Nothing is downloaded. --synthetic fabricates a tiny model + tokenizer.

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
    from llamarec_amd.rerank import LLMEvaluator, build_test_items
    from llamarec_amd.verb import ManualVerbalizer

    args = cfg.parse(argv, model_code="llm")
    if not args.llm_retrieved_path:
        raise SystemExit("--llm_retrieved_path experiments/lru/<dataset> is required")
    export_root = export_root or args.export_root or os.path.join(
        cfg.EXPERIMENT_ROOT, args.llm_base_model.rstrip("/").split("/")[-1], args.dataset_code)
    retrieved = pickle.load(open(os.path.join(args.llm_retrieved_path, "retrieved.pkl"), "rb"))
    if args.synthetic:
        from llamarec_amd.synth import FakeTokenizer, synth_llama_state

        dataset = D.synthetic_dataset(num_users=300, num_items=1000, seed=args.seed)
        tokenizer = FakeTokenizer()
        c = dict(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                 num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=2048,
                 rms_norm_eps=1e-5, rope_theta=10000.0)
        model = LlamaRanker.from_state_dict(synth_llama_state(c, args.seed), c)
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
        model = LlamaRanker.from_pretrained(args.llm_base_model, adapter_path=args.llm_adapter_path)
    ncls = args.llm_negative_sample_size + 1
    verbalizer = ManualVerbalizer(tokenizer=tokenizer, prefix="", post_log_softmax=False, classes=list(range(ncls)),
                                  label_words={i: chr(ord("A") + i) for i in range(ncls)})
    items = build_test_items(dataset, retrieved, tokenizer, args)
    ev = LLMEvaluator(args, model, items, verbalizer, export_root, batch_size=args.test_batch_size)
    metrics = ev.test(retrieved["test_retrieval"])
    print("Ranking Performance on Subset:", metrics)
    print("Overall Performance of Our Framework:", ev.overall_metrics)
    return metrics, ev.overall_metrics


if __name__ == "__main__":
    main()
