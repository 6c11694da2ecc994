"""Token-layout adapters between the stages: mirror of `fam/llm/adapters/{flattened_encodec,tilted_encodec}.py`.

Host-side integer bookkeeping (microseconds per utterance), same class names, constructor and `decode` contract as the
reference so `TTS` / `Model` call sites read the same.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def _equalise(hierarchies: List[List[int]]) -> List[List[int]]:
    """All codebooks must have the same number of frames; the reference truncates to the shortest with a warning
    (`flattened_encodec.py:25-31`, `tilted_encodec.py:31-37`)."""
    lengths = [len(h) for h in hierarchies]
    if len(set(lengths)) > 1:
        shortest = min(lengths)
        print("WARNING: Number of tokens at each hierarchy must be of the same length!")
        print(f"Truncating to min length of {shortest} tokens from {max(lengths)} max.")
        print(lengths)
        hierarchies = [h[:shortest] for h in hierarchies]
    return hierarchies


class BaseDataAdapter:
    """`fam/llm/adapters/base.py`."""


class FlattenedInterleavedEncodec2Codebook(BaseDataAdapter):
    """First-stage stream -> (text ids, [codebook 0, codebook 1])   (`flattened_encodec.py:8-33`).

    With E = end_of_audio_token: ids below E are codebook 0, [E, 2E) are codebook 1 (offset removed), 2E is the
    end-of-audio marker (dropped), anything above is text; the trailing text id (end-of-text) is dropped."""

    def __init__(self, end_of_audio_token: int):
        self._end_of_audio_token = end_of_audio_token

    def decode(self, tokens: Sequence[Sequence[int]]) -> Tuple[List[int], List[List[int]]]:
        assert len(tokens) == 1
        E = self._end_of_audio_token
        stream = list(tokens[0])
        text_ids = [t for t in stream if t > 2 * E]
        audio = _equalise([[t for t in stream if t < E], [t - E for t in stream if E <= t < 2 * E]])
        return text_ids[:-1], audio

    def encode(self, text_tokens, audio_tokens):
        raise NotImplementedError


class TiltedEncodec(BaseDataAdapter):
    """Second-stage (n, t) grid -> (text ids, n codebooks)   (`tilted_encodec.py:8-39`): row 0 carries text (> E) and
    codebook 0 (< E); the other rows carry one codebook each; E itself is padding."""

    def __init__(self, end_of_audio_token: int):
        self._end_of_audio_token = end_of_audio_token

    def decode(self, tokens: Sequence[Sequence[int]]) -> Tuple[List[int], List[List[int]]]:
        assert len(tokens) > 1
        E = self._end_of_audio_token
        text_ids = [t for t in tokens[0] if t > E]
        audio = _equalise([[t for t in row if t < E] for row in tokens])
        return text_ids[:-1], audio

    def encode(self, text_tokens, audio_tokens):
        raise NotImplementedError
