"""Oracle for the second-stage (non-causal, ~10 M) model.  TEST INFRASTRUCTURE.

Restates
  * `fam/llm/model.py:195-314`            (GPT.forward, non-causal branch: all positions, all heads)
  * `fam/llm/layers/attn.py:132-185`      (SelfAttention, `is_causal=False`)
  * `fam/llm/layers/layers.py:8-72`       (LayerNorm / RMSNorm / SwiGLU / MLP)
  * `fam/llm/layers/combined.py:7-52`     (Block)
  * `fam/llm/mixins/non_causal.py:15-67`  (_non_causal_sample: /T, top-k mask, softmax, multinomial)
Pinned against the reference `GPT` by tests/golden/stage2_*.npz.

rounding="bf16" emulates `torch.amp.autocast(bfloat16)` (`fam/llm/inference.py:80-82`): Linear inputs
and outputs are bf16, the SDPA output is bf16, GELU/SiLU outputs are bf16, while embeddings, norms and
the residual stream stay fp32 (autocast leaves parameters fp32 and promotes fp32 + bf16 -> fp32).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from .stage1 import rounder


@dataclass
class Stage2Config:
    """Subset of `GPTConfig` (fam/llm/model.py:27-47) read at inference."""

    block_size: int = 1024
    vocab_sizes: List[int] = field(default_factory=lambda: [1538, 1025])
    target_vocab_sizes: Optional[List[int]] = field(default_factory=lambda: [1025] * 6)
    n_layer: int = 6
    n_head: int = 6
    n_embd: int = 384
    bias: bool = False
    causal: bool = False
    norm_type: str = "rmsnorm"
    rmsnorm_eps: Optional[float] = 1e-5
    nonlinearity_type: str = "swiglu"
    swiglu_multiple_of: Optional[int] = 256
    speaker_emb_dim: Optional[int] = 256

    def hidden_dim(self) -> int:
        """layers.py:45-58."""
        h = 4 * self.n_embd
        if self.nonlinearity_type == "swiglu":
            h = int(2 * h / 3)
            h = self.swiglu_multiple_of * math.ceil(h / self.swiglu_multiple_of)
        return h


class Stage2Oracle:
    def __init__(self, weights: Dict[str, torch.Tensor], cfg: Stage2Config, rounding: str = "fp32"):
        self.w = {k: v.detach().to(torch.float32) for k, v in weights.items()}
        self.cfg = cfg
        self.rnd = rounder(rounding)

    def _linear(self, x, name):
        rnd = self.rnd
        y = rnd(x) @ rnd(self.w[name + ".weight"]).T
        b = self.w.get(name + ".bias")
        if b is not None:
            y = y + rnd(b)
        return rnd(y)

    def _norm(self, x, name):
        c = self.cfg
        if c.norm_type == "rmsnorm":  # layers.py:20-30
            return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + c.rmsnorm_eps) * self.w[name + ".weight"]
        return F.layer_norm(x, (c.n_embd,), self.w[name + ".weight"], self.w.get(name + ".bias"), 1e-5)  # :8-17

    def forward(self, idx: torch.Tensor, speaker_embs: Optional[torch.Tensor]) -> List[torch.Tensor]:
        """model.py:195-314.  idx (b, n_in_hierarchies, t) int; speaker_embs (b, spk) or None.
        Returns one (b, t, V_i) logits tensor per target hierarchy."""
        c, w, rnd = self.cfg, self.w, self.rnd
        b, nh, t = idx.shape
        assert nh == len(c.vocab_sizes) and t <= c.block_size
        idx = idx.long()
        x = torch.zeros(b, t, c.n_embd)
        for i in range(nh):
            x = x + w[f"transformer.wtes.{i}.weight"][idx[:, i]]
        x = x + w["transformer.wpe.weight"][torch.arange(t)][None]
        if speaker_embs is not None:
            spk = self._linear(speaker_embs.float()[:, None, :], "speaker_cond_pos")  # (b,1,c)
            x = x + spk
        hd = c.n_embd // c.n_head
        for li in range(c.n_layer):
            p = f"transformer.h.{li}."
            a_in = self._norm(x, p + "ln_1")
            qkv = self._linear(a_in, p + "attn.c_attn").view(b, t, 3, c.n_head, hd)
            q, k, v = (qkv[:, :, j].transpose(1, 2) for j in range(3))  # (b, nh, t, hd)
            s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
            if c.causal:
                s = s.masked_fill(~torch.tril(torch.ones(t, t, dtype=torch.bool)), float("-inf"))
            y = rnd(torch.softmax(s, dim=-1) @ v).transpose(1, 2).contiguous().view(b, t, c.n_embd)
            x = x + self._linear(y, p + "attn.c_proj")
            m_in = self._norm(x, p + "ln_2")
            if c.nonlinearity_type == "swiglu":
                g = self._linear(m_in, p + "mlp.swiglu.w1")
                u = self._linear(m_in, p + "mlp.swiglu.w3")
                h = rnd(rnd(F.silu(g)) * u)
            else:
                h = rnd(F.gelu(self._linear(m_in, p + "mlp.c_fc")))
            x = x + self._linear(h, p + "mlp.c_proj")
        x = self._norm(x, "transformer.ln_f")
        n_out = len(c.target_vocab_sizes) if c.target_vocab_sizes is not None else len(c.vocab_sizes)
        return [self._linear(x, f"lm_heads.{i}") for i in range(n_out)]

    def non_causal_sample(self, idx: torch.Tensor, speaker_embs: Optional[torch.Tensor], temperature: float = 1.0,
                          top_k: Optional[int] = 200, u: Optional[torch.Tensor] = None) -> torch.Tensor:
        """non_causal.py:15-67.  `u` (n_out, b, t) uniforms in (0,1) replace `torch.multinomial`'s RNG:
        the sampled id is the first index whose inclusive cumulative probability exceeds u (inverse-CDF),
        which has the same distribution.  With top_k == 1 the choice is the argmax whatever `u` is.
        Returns (b, n_out, t) int64."""
        b, _, t = idx.shape
        assert t == self.cfg.block_size
        outs = []
        for i, logits in enumerate(self.forward(idx, speaker_embs)):
            logits = self.rnd(logits / temperature).float()  # bf16 tensor / float -> bf16 under autocast
            if top_k is not None:
                v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
                logits = logits.masked_fill(logits < v[:, :, [-1]], float("-inf"))
            probs = torch.softmax(logits, dim=-1)
            if u is None:
                choice = torch.argmax(probs, dim=-1)
            else:
                cdf = torch.cumsum(probs, dim=-1)
                choice = (cdf <= u[i][..., None] * cdf[..., -1:]).sum(-1).clamp(max=probs.size(-1) - 1)
            outs.append(choice[:, None, :])
        return torch.cat(outs, dim=1)
