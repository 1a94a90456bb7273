"""Deterministic stand-in for the Llama-2 sentencepiece tokenizer (absent offline); lives in
llamarec_amd.synth so that `train_ranker.py --synthetic` does not import from tests/."""
from llamarec_amd.synth import FakeTokenizer  # noqa: F401
