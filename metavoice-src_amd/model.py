"""Mirror of `fam/llm/model.py` for the second-stage (non-causal) model on the MI355X engine.

`GPTConfig` keeps the reference's fields (`model.py:27-47`); `GPT` is a facade with the reference's constructor,
`load_state_dict` and `generate` whose arithmetic runs in libmetavoice_hip.so (`mv_stage2_*`, csrc/stage2.hip).  The causal
(first-stage) use of this class in the reference is its slow path; the first stage here is `fast_model.Transformer`.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _native as N

END_OF_TEXT_TOKEN = 1537  # fam/llm/model.py:15


@dataclass
class GPTConfig:
    """`fam/llm/model.py:27-47`."""

    block_size: int = 1024
    vocab_sizes: List[int] = field(default_factory=list)
    target_vocab_sizes: Optional[List[int]] = None
    n_layer: int = 12
    n_head: int = 12
    n_embd: int = 768
    dropout: float = 0.0
    spkemb_dropout: float = 0.0
    bias: bool = True
    causal: bool = True
    spk_emb_on_text: bool = True
    norm_type: str = "layernorm"
    rmsnorm_eps: Optional[float] = None
    nonlinearity_type: str = "gelu"
    swiglu_multiple_of: Optional[int] = None
    attn_kernel_type: str = "torch_attn"
    kv_cache_enabled: bool = False

    def hidden_dim(self) -> int:
        """MLP width, `fam/llm/layers/layers.py:45-58`."""
        h = 4 * self.n_embd
        if self.nonlinearity_type == "swiglu":
            h = int(2 * h / 3)
            h = self.swiglu_multiple_of * math.ceil(h / self.swiglu_multiple_of)
        return h


class GPT:
    """Engine-backed stand-in for the reference's `GPT` (`model.py:79-156`), non-causal inference only."""

    def __init__(self, config: GPTConfig, speaker_emb_dim: Optional[int] = None, max_batch: int = 4):
        if config.causal:
            raise NotImplementedError("the causal model runs on the fast path (fast_model.Transformer), not through GPT")
        if config.norm_type not in ("rmsnorm", "layernorm") or config.nonlinearity_type not in ("swiglu", "gelu"):
            raise ValueError("unsupported norm_type / nonlinearity_type")
        self.config = config
        self.speaker_emb_dim = speaker_emb_dim
        targets = config.target_vocab_sizes if config.target_vocab_sizes is not None else config.vocab_sizes
        cfg = N.mv_stage2_config()
        cfg.n_layer, cfg.n_head, cfg.n_embd, cfg.hidden = config.n_layer, config.n_head, config.n_embd, config.hidden_dim()
        cfg.block_size, cfg.spk_dim = config.block_size, int(speaker_emb_dim or 0)
        cfg.n_in, cfg.n_out = len(config.vocab_sizes), len(targets)
        for i, v in enumerate(config.vocab_sizes):
            cfg.vocab_in[i] = v
        for i, v in enumerate(targets):
            cfg.vocab_out[i] = v
        cfg.bias = int(config.bias)
        cfg.norm_type = 0 if config.norm_type == "rmsnorm" else 1
        cfg.act = 0 if config.nonlinearity_type == "swiglu" else 1
        cfg.causal = 0
        cfg.norm_eps = float(config.rmsnorm_eps if config.rmsnorm_eps is not None else 1e-5)
        cfg.max_batch = max_batch
        self.n_out = len(targets)
        self.target_vocab_sizes = list(targets)
        self._h = C.c_void_p()
        N.check(N.lib().mv_stage2_create(C.byref(cfg), C.byref(self._h)))
        self._finalized = False

    def close(self):
        if getattr(self, "_h", None):
            N.lib().mv_stage2_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        for name, t in state_dict.items():
            if name.startswith("_orig_mod."):
                name = name[len("_orig_mod."):]
            t = t.detach()
            if t.dtype not in (torch.bfloat16, torch.float32):
                t = t.float()
            t = t.contiguous()
            rows, cols = (1, t.numel()) if t.dim() == 1 else (t.shape[0], t.shape[1])
            dt = N.MV_BF16 if t.dtype == torch.bfloat16 else N.MV_F32
            N.check(N.lib().mv_stage2_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), dt, rows, cols))
        N.check(N.lib().mv_stage2_finalize(self._h))
        self._finalized = True
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # ---- forward / sampling ------------------------------------------------------------------------------------------
    def _forward(self, idx: torch.Tensor, speaker_embs: Optional[torch.Tensor]):
        b, c, t = idx.shape
        buf = np.ascontiguousarray(idx.detach().cpu().numpy().astype(np.int32))
        spk_p = None
        if speaker_embs is not None and self.speaker_emb_dim:
            spk = speaker_embs.detach().to("cpu", torch.bfloat16).reshape(b, -1).contiguous().view(torch.int16).numpy().view(np.uint16)
            spk_p = spk.ctypes.data_as(C.POINTER(C.c_uint16))
            self._keep = spk
        N.check(N.lib().mv_stage2_forward(self._h, b, t, buf.ctypes.data_as(C.POINTER(C.c_int32)), spk_p))
        return b, t

    def logits(self, idx: torch.Tensor, speaker_embs: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        """Parity hook: one fp32 (b, t, V_i) tensor per target hierarchy (`GPT.forward`, model.py:195-314)."""
        b, t = self._forward(idx, speaker_embs)
        outs = []
        for i, v in enumerate(self.target_vocab_sizes):
            out = np.empty((b, t, v), dtype=np.float32)
            N.check(N.lib().mv_stage2_get_logits(self._h, i, out.ctypes.data_as(C.POINTER(C.c_float))))
            outs.append(torch.from_numpy(out))
        return outs

    @torch.no_grad()
    def generate(self, idx: torch.Tensor, max_new_tokens: Optional[int] = None, seq_lens=None, temperature: float = 1.0,
                 top_k: Optional[int] = None, top_p: Optional[float] = None, speaker_embs: Optional[torch.Tensor] = None,
                 batch_size: Optional[int] = None, guidance_scale=None, seed: int = 1337) -> torch.Tensor:
        """`GPT.generate`, non-causal branch (`model.py:384-408`) + `_non_causal_sample` (`mixins/non_causal.py:15-67`):
        (b, n_in, block_size) -> (b, n_out, block_size) int64."""
        if seq_lens is not None:
            raise Exception("seq_lens is not supported yet for non-causal sampling")
        if batch_size is None:
            raise Exception("batch_size must be provided for non-causal sampling")
        if guidance_scale is not None:
            raise Exception("guidance_scale is not supported for non-causal sampling")
        if top_p is not None:
            raise Exception("top_p is not supported for non-causal sampling")
        b, c, t = idx.shape
        assert t == self.config.block_size, f"input size {t} != config.block_size {self.config.block_size}"
        outs = []
        for start in range(0, b, batch_size):
            end = min(start + batch_size, b)
            self._forward(idx[start:end], speaker_embs[start:end] if speaker_embs is not None else None)
            out = np.empty((end - start, self.n_out, t), dtype=np.int32)
            N.check(N.lib().mv_stage2_sample(self._h, float(temperature), int(top_k) if top_k is not None else 0,
                                             (int(seed) + start) & 0xFFFFFFFFFFFFFFFF, out.ctypes.data_as(C.POINTER(C.c_int32))))
            outs.append(torch.from_numpy(out).long())
        return torch.cat(outs, dim=0)
