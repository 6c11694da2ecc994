"""Text tokeniser of the first / second stage: mirror of `fam/quantiser/text/tokenise.py:4-32`.

The reference wraps `tiktoken.Encoding(name, pat_str, mergeable_ranks, special_tokens)` built from the ranks stored in
the checkpoint (`checkpoint["meta"]["tokenizer"]`).  tiktoken is a compiled third-party package that is not part of this
image, so the same byte-pair-encoding algorithm (tiktoken 0.5.1 `_byte_pair_merge`: split with `pat_str`, then inside a
piece repeatedly merge the adjacent pair of lowest rank) is restated here in Python; it is host-side, runs in
microseconds per prompt and is not on the GPU path.
"""
from __future__ import annotations

from typing import Dict, List, Optional

try:
    import regex as _re  # tiktoken's pat_str uses \p{L} / \p{N} classes
except ImportError:  # pragma: no cover
    import re as _re


def _bpe(ranks: Dict[bytes, int], piece: bytes) -> List[int]:
    if piece in ranks:
        return [ranks[piece]]
    parts = [bytes([b]) for b in piece]
    while len(parts) > 1:
        best_rank, best_i = None, -1
        for i in range(len(parts) - 1):
            r = ranks.get(parts[i] + parts[i + 1])
            if r is not None and (best_rank is None or r < best_rank):
                best_rank, best_i = r, i
        if best_rank is None:
            break
        parts[best_i:best_i + 2] = [parts[best_i] + parts[best_i + 1]]
    return [ranks[p] for p in parts]


class TrainedBPETokeniser:
    def __init__(self, name, pat_str, mergeable_ranks, special_tokens, offset=None) -> None:
        self.name = name
        self._pat = _re.compile(pat_str)
        self._ranks: Dict[bytes, int] = dict(mergeable_ranks)
        self._special: Dict[str, int] = dict(special_tokens)
        self._decoder = {v: k for k, v in self._ranks.items()}
        self._special_decoder = {v: k.encode() for k, v in self._special.items()}
        self.offset = offset

    @property
    def _eot(self) -> int:
        return self._special["<|endoftext|>"]

    def encode(self, text: str) -> List[int]:
        # note: an end-of-text token is appended (tokenise.py:14-20)
        for s in self._special:
            if s and s in text:
                raise ValueError(f"Encountered text corresponding to disallowed special token {s!r}.")
        tokens: List[int] = []
        for piece in self._pat.findall(text):
            tokens.extend(_bpe(self._ranks, piece.encode("utf-8")))
        tokens.append(self._eot)
        if self.offset is not None:
            tokens = [t + self.offset for t in tokens]
        return tokens

    def decode(self, tokens: List[int]) -> str:
        if self.offset is not None:
            tokens = [t - self.offset for t in tokens]
        out = b"".join(self._decoder.get(t, self._special_decoder.get(t, b"")) for t in tokens)
        return out.decode("utf-8", errors="replace")

    @property
    def eot_token(self) -> int:
        return self._eot + self.offset if self.offset is not None else self._eot


def byte_level_tokenizer_info(n_merges: int = 256, offset: Optional[int] = 2049) -> dict:
    """A synthetic `checkpoint["meta"]["tokenizer"]` with the MetaVoice layout (512 ranks + EOT = 513 text ids, offset
    2049: fam/llm/preprocessing/audio_token_mode.py:38-49): 256 byte tokens + `n_merges` merges of frequent English
    bigrams / byte pairs.  Stands in for the real ranks, which live only in the (unreachable) checkpoint."""
    ranks = {bytes([b]): b for b in range(256)}
    common = ["th", "he", "in", "er", "an", "re", "on", "at", "en", "nd", "ti", "es", "or", "te", "of", "ed", "is", "it",
              "al", "ar", "st", "to", "nt", "ng", "se", "ha", "as", "ou", "io", "le", "ve", "co", "me", "de", "hi", "ri",
              "ro", "ic", "ne", "ea", "ra", "ce", "li", "ch", "ll", "be", "ma", "si", "om", "ur", " t", " a", " s", " w",
              " o", " i", " h", " b", " c", " m", " f", " p", " d", "e ", "s ", "d ", "t ", "n ", "y ", ", ", ". "]
    nxt = 256
    for s in common:
        if nxt >= 256 + n_merges:
            break
        ranks[s.encode()] = nxt
        nxt += 1
    a, b = 0, 1
    while nxt < 256 + n_merges:  # fill the remaining ranks with unlikely byte pairs so the table has exactly 512 rows
        key = bytes([128 + a, 128 + b])
        if key not in ranks:
            ranks[key] = nxt
            nxt += 1
        b += 1
        if b >= 128:
            a, b = a + 1, 0
    pat = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""
    return dict(name="synthetic_bpe_512", pat_str=pat, mergeable_ranks=ranks, special_tokens={"<|endoftext|>": 256 + n_merges},
                offset=offset)
