"""Mirror of `fam/llm/fast_model.py` for the MI355X engine.

`ModelArgs` / `transformer_configs` / `find_multiple` keep the reference's names and meaning (:45-94).  `Transformer`
is a facade with the reference's constructor and method names (`from_name`, `setup_caches`, `setup_spk_cond_mask`,
`load_state_dict`) whose arithmetic runs in libmetavoice_hip.so -- there are no torch modules and no torch math here.
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import reduce
from math import gcd
from typing import Dict, Optional

import torch

from .engine import Stage1Engine


def find_multiple(n: int, *args: int) -> int:
    """fam/llm/fast_model.py:45-49."""
    k = reduce(lambda x, y: x * y // gcd(x, y), args + (1,))
    return n if n % k == 0 else n + k - (n % k)


@dataclass
class ModelArgs:
    """fam/llm/fast_model.py:52-84.  `dtype` is always bfloat16 here: MI355X (gfx950, major 9) takes the reference's
    bf16 branch of `get_default_dtype` (fam/llm/utils.py:77-84)."""

    block_size: int = 2048
    vocab_size: int = 32000
    n_layer: int = 32
    n_head: int = 32
    dim: int = 4096
    speaker_emb_dim: int = 256
    intermediate_size: Optional[int] = None
    n_local_heads: int = -1
    head_dim: int = 64
    norm_eps: float = 1e-5
    dtype: torch.dtype = torch.bfloat16

    def __post_init__(self):
        if self.n_local_heads == -1:
            self.n_local_heads = self.n_head
        if self.intermediate_size is None:
            hidden_dim = 4 * self.dim
            n_hidden = int(2 * hidden_dim / 3)
            self.intermediate_size = find_multiple(n_hidden, 256)
        self.head_dim = self.dim // self.n_head

    @classmethod
    def from_name(cls, name: str):
        if name in transformer_configs:
            return cls(**transformer_configs[name])
        config = [c for c in transformer_configs if c in str(name).upper() or c in str(name)]
        assert len(config) == 1, name
        return cls(**transformer_configs[config[0]])


transformer_configs = {
    "metavoice-1B": dict(n_layer=24, n_head=16, dim=2048, vocab_size=2562),
}


class Transformer:
    """Engine-backed stand-in for the reference's `Transformer` module (fam/llm/fast_model.py:119-163)."""

    def __init__(self, config: ModelArgs, max_utterances: int = 1, quantisation_mode: Optional[str] = None) -> None:
        if config.n_local_heads != config.n_head:
            raise ValueError("grouped-query attention is not part of MetaVoice-1B (n_local_heads must equal n_head)")
        self.config = config
        self.engine = Stage1Engine(config, max_utterances=max_utterances, quantisation_mode=quantisation_mode)
        self.max_batch_size = -1
        self.max_seq_length = -1

    @classmethod
    def from_name(cls, name: str, **kw):
        return cls(ModelArgs.from_name(name), **kw)

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], assign: bool = True, strict: bool = True):
        self.engine.load_state_dict(state_dict)
        return self

    def setup_spk_cond_mask(self):
        """fam/llm/fast_model.py:132-134: row 0 of each CFG pair is speaker-conditioned, row 1 is not.  The mask is
        built into the engine's embedding kernel; kept for call-site compatibility."""

    def setup_caches(self, max_batch_size: int, max_seq_length: int):
        """fam/llm/fast_model.py:136-148.  Static KV caches for 2 rows per utterance x block_size are allocated (once)
        by `mv_finalize`, together with the decode-step hipGraph."""
        if max_batch_size > 2 * self.engine.max_utterances:
            raise ValueError(f"engine was built for {self.engine.max_utterances} utterance(s) = {2 * self.engine.max_utterances} rows")
        if max_seq_length > self.config.block_size:
            raise ValueError("max_seq_length exceeds block_size")
        self.engine.finalize()
        self.max_batch_size, self.max_seq_length = max_batch_size, find_multiple(max_seq_length, 8)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self
