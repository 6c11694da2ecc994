"""Oracle for the first-stage (causal, 1.2 B) decoder and its sampler.  TEST INFRASTRUCTURE.

Functional torch-CPU restatement of
  * `fam/llm/fast_model.py:119-261`     (Transformer / Attention / SwiGLU / RMSNorm / KVCache)
  * `fam/llm/fast_inference_utils.py:61-228` (sampling, prefill, decode loop, generate)
Checked against the reference itself by `tests/golden/make_golden.py` (vectors in
`tests/golden/stage1_*.npz`).

Two arithmetic modes:
  rounding="fp32"  -- everything in fp32 (what the reference does on CPU in fp32; BASELINE
                      config 1).  Pinned to the reference to ~1e-6.
  rounding="bf16"  -- fp32 arithmetic, but every tensor the reference would materialise in
                      bf16 on its GPU path is rounded to bf16 (RNE) at that point:
                      embedding sum, norm output, every linear output, attention output,
                      residual sums, silu / gate product, logits, CFG mix, temperature divide.
                      Inside attention (scores, softmax, P.V) and inside the sampler after the
                      temperature divide the arithmetic stays fp32 -- this is the HIP engine's
                      definition (DESIGN.md "Rounding points").  The reference's own bf16 eager
                      run agrees with this mode to bf16 noise only (pinned loosely).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch

from . import rng as _rng


def find_multiple(n: int, k: int) -> int:
    """fam/llm/fast_model.py:45-49 (single-argument form)."""
    return n if n % k == 0 else n + k - (n % k)


@dataclass
class Stage1Config:
    """Mirror of `ModelArgs` (fam/llm/fast_model.py:52-84); defaults = "metavoice-1B" (:87-94)."""

    block_size: int = 2048
    vocab_size: int = 2562
    n_layer: int = 24
    n_head: int = 16
    dim: int = 2048
    speaker_emb_dim: int = 256
    intermediate_size: Optional[int] = None
    norm_eps: float = 1e-5

    def __post_init__(self):
        if self.intermediate_size is None:
            # fast_model.py:69-73
            self.intermediate_size = find_multiple(int(2 * (4 * self.dim) / 3), 256)
        assert self.dim % self.n_head == 0
        self.head_dim = self.dim // self.n_head


def _identity(t: torch.Tensor) -> torch.Tensor:
    return t


def _round_bf16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def rounder(rounding: str) -> Callable[[torch.Tensor], torch.Tensor]:
    if rounding == "fp32":
        return _identity
    if rounding == "bf16":
        return _round_bf16
    raise ValueError(f"unknown rounding mode {rounding!r}")


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float, rnd) -> torch.Tensor:
    """fam/llm/fast_model.py:250-261: fp32 normalise, cast to activation dtype, then * weight."""
    xf = x.float()
    normed = xf * torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + eps)
    return rnd(rnd(normed) * weight)


class Stage1Oracle:
    """Stateful (KV-cache) restatement of `Transformer` with `max_batch_size=2` CFG rows.

    `weights` uses the gpt-fast names produced by the key remap of
    `fam/llm/fast_inference_utils.py:250-278`, all fp32 tensors (already bf16-rounded by the
    caller when emulating a bf16 checkpoint).
    """

    def __init__(self, weights: Dict[str, torch.Tensor], cfg: Stage1Config, rounding: str = "fp32",
                 max_batch_size: int = 2, max_seq_length: Optional[int] = None):
        self.w = {k: v.detach().to(torch.float32) for k, v in weights.items()}
        self.cfg = cfg
        self.rounding = rounding
        self.rnd = rounder(rounding)
        self.max_batch_size = max_batch_size
        self.max_seq_length = max_seq_length or cfg.block_size
        self.setup_caches()

    # fam/llm/fast_model.py:136-148
    def setup_caches(self):
        c = self.cfg
        shape = (self.max_batch_size, c.n_head, self.max_seq_length, c.head_dim)
        self.k_cache = [torch.zeros(shape) for _ in range(c.n_layer)]
        self.v_cache = [torch.zeros(shape) for _ in range(c.n_layer)]

    def quantise_int8(self):
        """`WeightOnlyInt8QuantHandler` (fam/llm/fast_quantize.py:137-178): every nn.Linear -- wqkv, wo, w1, w3, w2, the LM head
        and speaker_cond_pos -- becomes (int8 codes, bf16 per-row scale); `_lin` then evaluates `WeightOnlyInt8Linear.forward`
        (:197-198): F.linear(x, q.to(bf16)) * scales, i.e. bf16(bf16(x @ q^T) * scale) on the bf16 path."""
        from . import quantize as _q

        self.int8 = {}
        for name, wt in self.w.items():
            if wt.dim() == 2 and "embeddings" not in name:
                codes, scales = _q.quantize_per_channel_int8(wt.float())
                self.int8[name] = (codes.float(), scales.to(torch.bfloat16).float())

    def _lin(self, x: torch.Tensor, name: str) -> torch.Tensor:
        q = getattr(self, "int8", {}).get(name)
        if q is not None:
            return self.rnd(self.rnd(x @ q[0].T) * q[1])
        return self.rnd(x @ self.w[name].T)

    def spk_proj(self, spk_emb: torch.Tensor) -> torch.Tensor:
        """`speaker_cond_pos(spk_emb) * spk_cond_mask` (fast_model.py:132-134,156): row 0 keeps the
        speaker projection, every odd CFG row (the unconditioned one) gets zero."""
        proj = self._lin(spk_emb.float(), "speaker_cond_pos.weight")  # (1, dim)
        return proj

    def forward(self, idx: torch.Tensor, spk_emb: torch.Tensor, input_pos: torch.Tensor) -> torch.Tensor:
        """fam/llm/fast_model.py:150-163.  idx (B,S) int, spk_emb (1,spk), input_pos (S,) -> (B,S,V).

        B must be even: rows [0, B/2) are speaker-conditioned, rows [B/2, B) unconditioned -- for the
        reference's B=2 this is exactly `spk_cond_mask` (row 0 = 1, row 1 = 0).
        """
        c, w, rnd = self.cfg, self.w, self.rnd
        B, S = idx.shape
        idx = idx.long()
        input_pos = input_pos.long()
        mask_rows = torch.zeros(B, 1, 1)
        mask_rows[: max(B // 2, 1)] = 1.0
        x = rnd(w["tok_embeddings.weight"][idx] + w["pos_embeddings.weight"][input_pos][None])
        x = rnd(x + self.spk_proj(spk_emb)[None] * mask_rows)

        L = int(input_pos.max().item()) + 1
        key_pos = torch.arange(L)
        allowed = key_pos[None, :] <= input_pos[:, None]  # (S, L) == causal_mask[input_pos][:, :L]
        scale = 1.0 / math.sqrt(c.head_dim)

        for li in range(c.n_layer):
            p = f"layers.{li}."
            # --- attention (fast_model.py:200-227)
            h_in = rms_norm(x, w[p + "attention_norm.weight"], c.norm_eps, rnd)
            qkv = self._lin(h_in, p + "attention.wqkv.weight")
            q, k, v = qkv.split([c.dim, c.dim, c.dim], dim=-1)
            q = q.view(B, S, c.n_head, c.head_dim).transpose(1, 2)
            k = k.view(B, S, c.n_head, c.head_dim).transpose(1, 2)
            v = v.view(B, S, c.n_head, c.head_dim).transpose(1, 2)
            self.k_cache[li][:B, :, input_pos] = k  # KVCache.update, fast_model.py:104-113
            self.v_cache[li][:B, :, input_pos] = v
            kk = self.k_cache[li][:B, :, :L]
            vv = self.v_cache[li][:B, :, :L]
            scores = (q @ kk.transpose(-1, -2)) * scale
            scores = scores.masked_fill(~allowed[None, None], float("-inf"))
            probs = torch.softmax(scores, dim=-1)
            y = rnd(probs @ vv)
            y = y.transpose(1, 2).contiguous().view(B, S, c.dim)
            y = self._lin(y, p + "attention.wo.weight")
            h = rnd(x + y)
            # --- feed forward (fast_model.py:230-247)
            f_in = rms_norm(h, w[p + "ffn_norm.weight"], c.norm_eps, rnd)
            g = self._lin(f_in, p + "feed_forward.swiglu.w1.weight")
            u = self._lin(f_in, p + "feed_forward.swiglu.w3.weight")
            act = rnd(rnd(torch.nn.functional.silu(g)) * u)
            d = self._lin(act, p + "feed_forward.w2.weight")
            x = rnd(h + d)

        x = rms_norm(x, w["norm.weight"], c.norm_eps, rnd)
        return self._lin(x, "output.weight")


# ----------------------------------------------------------------------------------------------
# Sampling: fam/llm/fast_inference_utils.py:61-120
# ----------------------------------------------------------------------------------------------

def top_p_keep_mask(logits: torch.Tensor, top_p: float) -> torch.Tensor:
    """fast_inference_utils.py:68-82.  Returns the boolean KEEP mask in original indexing.

    Ascending sort, softmax, cumulative sum; a token is removed iff its ascending cumulative
    probability is <= 1 - top_p; the largest logit is always kept.  Sort order among equal
    logits: ascending token id (stable) -- the reference leaves it unspecified.
    """
    sorted_logits, sorted_idx = torch.sort(logits, descending=False, stable=True)
    cum = torch.cumsum(torch.softmax(sorted_logits, dim=-1), dim=-1)
    remove_sorted = cum <= (1.0 - top_p)
    remove_sorted[-1:] = False
    remove = torch.zeros_like(remove_sorted).scatter(0, sorted_idx, remove_sorted)
    return ~remove


def logits_to_probs(logits: torch.Tensor, temperature: float, top_p: Optional[float] = None,
                    top_k: Optional[int] = None, rnd=_identity) -> torch.Tensor:
    """fast_inference_utils.py:85-104 on a (V,) vector."""
    logits = rnd(logits / max(float(temperature), 1e-5)).to(torch.float32)
    if top_k is not None:
        v, _ = torch.topk(logits, min(int(top_k), logits.size(-1)))
        pivot = v[-1]
        logits = torch.where(logits < pivot, torch.full_like(logits, float("-inf")), logits)
    if top_p is not None:
        keep = top_p_keep_mask(logits, float(top_p))
        logits = logits.masked_fill(~keep, float("-inf"))
    return torch.softmax(logits, dim=-1)


def cfg_mix(logits: torch.Tensor, guidance_scale: float, rnd=_identity) -> torch.Tensor:
    """fast_inference_utils.py:113-117: `g * cond + (1 - g) * uncond` on the last position.  `(1 - g)` is itself a
    tensor op in the model dtype there, hence the rounding of the scalar."""
    last = logits[:, -1]
    half = last.size(0) // 2
    cond, uncond = last[:half], last[half:]
    g = float(guidance_scale)
    one_minus_g = float(rnd(torch.tensor(1.0 - g)))
    return rnd(rnd(g * cond) + rnd(one_minus_g * uncond))


def sample(logits: torch.Tensor, guidance_scale: float, temperature: float, top_p: Optional[float] = None,
           top_k: Optional[int] = None, q: Optional[torch.Tensor] = None, rnd=_identity):
    """fast_inference_utils.py:107-120 + :61-65.  `q` is the Exp(1) race noise (V,), injected.

    Returns (idx_next int, probs (V,)).  argmax ties resolve to the lowest token id.
    """
    mixed = cfg_mix(logits, guidance_scale, rnd)
    probs = logits_to_probs(mixed[0], temperature, top_p=top_p, top_k=top_k, rnd=rnd)
    if q is None:
        q = torch.ones_like(probs)
    idx_next = int(torch.argmax(probs / q.to(probs.dtype)).item())
    return idx_next, probs


def generate(model: Stage1Oracle, prompt: torch.Tensor, spk_emb: torch.Tensor, *, max_new_tokens: Optional[int] = None,
             end_of_audio_token: int = 2048, guidance_scale: float = 3.0, temperature: float = 1.0,
             top_p: Optional[float] = None, top_k: Optional[int] = None, seed: int = 0,
             noise: bool = True, return_logits: bool = False, n_utt_rows: int = 2):
    """fast_inference_utils.py:182-228 (+ prefill :123-132, decode_n_tokens :148-174).

    Noise for the token sampled at sequence position p (prefill samples position T) comes from
    `oracle.rng.exp1_noise(seed, step=p, n=vocab)`, the same stream the HIP sampler uses.
    Returns the full sequence (prompt + generated) as a python list; the end-of-audio token, when
    drawn, IS included (the reference only tests it at the top of the next iteration, :161).
    """
    T = int(prompt.numel())
    block = model.cfg.block_size
    max_seq_length = block if max_new_tokens is None else min(T + max_new_tokens, block)
    max_new_tokens = max_seq_length - T
    if max_new_tokens <= 0:
        raise ValueError("Prompt is too long to generate more tokens")  # :203-204
    V = model.cfg.vocab_size
    rnd = model.rnd

    def draw(logits, pos):
        q = torch.from_numpy(_rng.exp1_noise(seed, pos, V)) if noise else None
        return sample(logits, guidance_scale, temperature, top_p=top_p, top_k=top_k, q=q, rnd=rnd)[0]

    seq: List[int] = [int(t) for t in prompt.tolist()]
    all_logits = []
    x = prompt.view(1, -1).repeat(n_utt_rows, 1)
    logits = model.forward(x, spk_emb, torch.arange(T))
    all_logits.append(logits[:, -1].clone())
    cur = draw(logits, T)
    seq.append(cur)
    pos = T
    for _ in range(max_new_tokens - 1):
        if cur == end_of_audio_token:
            break
        x = torch.full((n_utt_rows, 1), cur, dtype=torch.long)
        logits = model.forward(x, spk_emb, torch.tensor([pos]))
        all_logits.append(logits[:, -1].clone())
        pos += 1
        cur = draw(logits, pos)
        seq.append(cur)
    if return_logits:
        return seq, torch.stack(all_logits)
    return seq


# ----------------------------------------------------------------------------------------------
# Checkpoint key remap: fam/llm/fast_inference_utils.py:243-281
# ----------------------------------------------------------------------------------------------

def remap_checkpoint_keys(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """nanoGPT-style `first_stage.pt["model"]` names -> gpt-fast names (restated from :246-278)."""
    out = {}
    for k, v in state_dict.items():
        if k.startswith("_orig_mod."):
            k = k[len("_orig_mod."):]
        out[k] = v
    table = {
        "transformer.wtes.0.weight": "tok_embeddings.weight",
        "transformer.wpe.weight": "pos_embeddings.weight",
        "lm_heads.0.weight": "output.weight",
        "transformer.ln_f.weight": "norm.weight",
    }
    subs = [
        ("transformer.h.", "layers."),
        (".attn.c_attn.", ".attention.wqkv."),
        (".attn.c_proj.", ".attention.wo."),
        (".mlp.swiglu.w1.", ".feed_forward.swiglu.w1."),
        (".mlp.swiglu.w3.", ".feed_forward.swiglu.w3."),
        (".ln_1.", ".attention_norm."),
        (".ln_2.", ".ffn_norm."),
        (".mlp.c_proj.", ".feed_forward.w2."),
    ]
    final = {}
    for k, v in out.items():
        k = table.get(k, k)
        for a, b in subs:
            k = k.replace(a, b)
        final[k] = v
    return final
