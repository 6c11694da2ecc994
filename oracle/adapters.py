"""Oracle for the integer token-layout code between the stages.  TEST INFRASTRUCTURE.

Restates
  * `fam/llm/adapters/flattened_encodec.py:8-33`  (stage-1 stream -> 2 codebooks)
  * `fam/llm/adapters/tilted_encodec.py:8-39`     (stage-2 (8,t) grid -> 8 codebooks)
  * `fam/llm/inference.py:264-306`                (stage-2 input packing; cannot be imported here
                                                   because the module pulls in audiocraft -- restated
                                                   from source, unpinned by execution)
  * `fam/llm/utils.py:12-52`                      (normalize_text)
Pinned (adapters, normalize_text) by tests/golden/adapters.json.
"""
from __future__ import annotations

import re
from typing import List, Tuple


def flattened_interleaved_decode(tokens: List[int], end_of_audio_token: int = 1024) -> Tuple[List[int], List[List[int]]]:
    """flattened_encodec.py:8-33.  t < E -> cb0; E <= t < 2E -> cb1 (t-E); t == 2E dropped; t > 2E -> text.
    Both codebooks are truncated to the shorter one; the last text id (EOT) is dropped."""
    E = end_of_audio_token
    text_ids: List[int] = []
    cb = [[], []]
    for t in tokens:
        if t < E:
            cb[0].append(t)
        elif t < 2 * E:
            cb[1].append(t - E)
        elif t > 2 * E:
            text_ids.append(t)
    m = min(len(cb[0]), len(cb[1]))
    cb = [cb[0][:m], cb[1][:m]]
    return text_ids[:-1], cb


def tilted_decode(tokens: List[List[int]], end_of_audio_token: int = 1024) -> Tuple[List[int], List[List[int]]]:
    """tilted_encodec.py:8-39.  Row 0: > E -> text, < E -> audio; rows 1..: < E -> audio; truncate to min."""
    E = end_of_audio_token
    assert len(tokens) > 1
    text_ids = [t for t in tokens[0] if t > E]
    out = [[t for t in tokens[0] if t < E]]
    for row in tokens[1:]:
        out.append([t for t in row if t < E])
    m = min(len(r) for r in out)
    out = [r[:m] for r in out]
    return text_ids[:-1], out


def pack_second_stage_input(text_tokens: List[int], codebooks: List[List[int]], ctx_window: int,
                            pad_token: int = 1024) -> List[List[int]]:
    """inference.py:283-306.  Row 0 = text ++ cb0 ++ [pad]; row 1 = [pad]*len(text) ++ cb1 ++ [pad];
    right-pad with `pad_token` or truncate to `ctx_window`."""
    rows = [
        list(text_tokens) + list(codebooks[0]) + [pad_token],
        [pad_token] * len(text_tokens) + list(codebooks[1]) + [pad_token],
    ]
    assert len(rows[0]) == len(rows[1])
    out = []
    for r in rows:
        if len(r) < ctx_window:
            r = r + [pad_token] * (ctx_window - len(r))
        elif len(r) > ctx_window:
            r = r[:ctx_window]
        out.append(r)
    return out


_UNICODE_CONVERSION = {
    8175: "'", 8189: "'", 8190: "'", 8208: "-", 8209: "-", 8210: "-", 8211: "-", 8212: "-", 8213: "-",
    8214: "||", 8216: "'", 8217: "'", 8218: ",", 8219: "`", 8220: '"', 8221: '"', 8222: ",,", 8223: '"',
    8228: ".", 8229: "..", 8230: "...", 8242: "'", 8243: '"', 8245: "'", 8246: '"', 180: "'", 2122: "TM",
}


def normalize_text(text: str) -> str:
    """utils.py:12-52: unicode punctuation folding, reject code points >= 256, whitespace squeeze."""
    text = text.translate(_UNICODE_CONVERSION)
    bad = sorted({(c, ord(c)) for c in text if ord(c) >= 256})
    if bad:
        raise ValueError(f"Non-supported character found: {bad}")
    text = text.replace("\t", " ").replace("\n", " ").replace("\r", " ").replace("*", " ").strip()
    return re.sub(r"\s\s+", " ", text)
