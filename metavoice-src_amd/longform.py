"""Long-form synthesis (BASELINE.json configs[4]): the reference synthesises at most 220 characters per call
(`fam/llm/inference.py:535-541`: "supports max. 220 characters atm. Long form synthesis coming soon...") and has no chunker;
this one splits a long text at sentence boundaries into pieces of at most `max_chars` characters, each of which is an
independent generation with the same speaker embedding.  Parity with the reference is per chunk.
"""
from __future__ import annotations

import re
from typing import List

MAX_CHARS = 220  # fam/llm/inference.py:537

# a sentence ends at . ! ? plus any closing quotes / brackets, which STAY with the sentence: only the whitespace after them is the cut
_SENTENCE_END = re.compile(r"[.!?][\"')\]]*(\s+)")
_CLAUSE_END = re.compile(r"(?<=[,;:])\s+")


def _pack(pieces: List[str], max_chars: int) -> List[str]:
    """Greedy: append pieces to the current chunk while it stays within `max_chars`."""
    chunks: List[str] = []
    cur = ""
    for p in pieces:
        if not p:
            continue
        if cur and len(cur) + 1 + len(p) <= max_chars:
            cur = cur + " " + p
        else:
            if cur:
                chunks.append(cur)
            cur = p
    if cur:
        chunks.append(cur)
    return chunks


def _split_oversized(piece: str, max_chars: int) -> List[str]:
    """A single sentence longer than `max_chars`: fall back to clause boundaries, then to whitespace, then to a hard cut."""
    if len(piece) <= max_chars:
        return [piece]
    out: List[str] = []
    for clause in _pack([c.strip() for c in _CLAUSE_END.split(piece)], max_chars):
        if len(clause) <= max_chars:
            out.append(clause)
            continue
        for words in _pack(clause.split(), max_chars):
            while len(words) > max_chars:  # one "word" longer than the limit
                out.append(words[:max_chars])
                words = words[max_chars:]
            if words:
                out.append(words)
    return out


def split_text(text: str, max_chars: int = MAX_CHARS) -> List[str]:
    """-> chunks of <= `max_chars` characters that concatenate (with single spaces) back to the whitespace-normalised text.
    Sentences are kept whole and packed greedily; nothing is dropped or truncated."""
    if max_chars < 1:
        raise ValueError("max_chars must be positive")
    text = " ".join(text.split())
    if not text:
        return []
    sentences: List[str] = []
    start = 0
    for m in _SENTENCE_END.finditer(text):
        sentences.extend(_split_oversized(text[start:m.start(1)], max_chars))
        start = m.end(1)
    sentences.extend(_split_oversized(text[start:], max_chars))
    return _pack(sentences, max_chars)
