"""A real HF fast tokenizer built offline (no Llama tokenizer files exist in this container): byte-level-fallback
BPE with Llama-style metaspace ("▁") pre-tokenisation / decoding and a BOS post-processor, trained deterministically
on the small corpus below and wrapped in transformers.PreTrainedTokenizerFast -- the class family
AutoTokenizer returns for the reference's Llama models (dataloader/llm.py:119-126). Used by the golden generator
(reference functions) and by the tests (this repo's mirrors) so both see the same tokenizer behaviour:
sub-word splits, convert_tokens_to_string, special tokens, left truncation."""
import numpy as np

_WORDS = ["Toy", "Story", "Heat", "Casino", "Se7en", "Usual", "Suspects,", "The", "Braveheart", "Apollo", "13", "Léon:",
          "Professional", "Pulp", "Fiction", "(1995)", "(1994)", "A", "very", "long", "title", "that", "certainly", "exceeds",
          "the", "limit", "of", "tokens", "Return", "King", "Star", "Wars", "Episode", "II", "Deluxe", "Edition", "Lotion"]


def corpus():
    rng = np.random.default_rng(1234)
    lines = [" ".join(rng.choice(_WORDS, size=int(rng.integers(2, 9)))) for _ in range(2000)]
    lines.append("### Instruction:\nGiven user history in chronological order, recommend an item from the candidate pool "
                 "with its index letter.\n\n### Input:\nUser history: (1) a \n (2) b; \n Candidate pool: (A) c \n (B) d\n\n"
                 "### Response:\n")
    lines.append(" ".join(chr(65 + i) for i in range(26)) + " " + " ".join(f"({chr(65 + i)})" for i in range(26)))
    lines.append(" ".join(f"({i})" for i in range(1, 40)))
    return lines


def build_llama_like_tokenizer(vocab_size=700):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    from transformers import PreTrainedTokenizerFast

    tok = Tokenizer(models.BPE(unk_token="<unk>", byte_fallback=True))
    tok.pre_tokenizer = pre_tokenizers.Metaspace(replacement="▁", prepend_scheme="first")
    tok.decoder = decoders.Sequence([decoders.Replace("▁", " "), decoders.ByteFallback(), decoders.Fuse(),
                                     decoders.Strip(content=" ", left=1)])
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=["<unk>", "<s>", "</s>"],
                                  initial_alphabet=[], show_progress=False,
                                  continuing_subword_prefix="", end_of_word_suffix="")
    tok.train_from_iterator(corpus(), trainer)
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", tok.token_to_id("<s>"))])
    t = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>")
    # dataloader/llm.py:122-126
    t.pad_token = t.unk_token
    t.padding_side = "left"
    t.truncation_side = "left"
    t.clean_up_tokenization_spaces = True
    return t
