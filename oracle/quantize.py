"""Oracle for the weight-only quantisers.  TEST INFRASTRUCTURE.

Restates `fam/llm/fast_quantize.py`:
  * int4, group 128, asymmetric:  :70-132  (get_group_qparams, group_quantize_tensor_from_qparams,
    pack_scales_and_zeros, group_dequantize_tensor_from_qparams)
  * int8, per output channel, symmetric:  :38-67  (dynamically_quantize_per_channel)
  * which layers get replaced:  :220-252, :273-276 (int4: nn.Linear with out_features % 8 == 0 --
    the 2562-row LM head stays bf16);  :137-160 (int8: every nn.Linear)
The reference multiplies through `aten::_weight_int4pack_mm` (a CUDA tinygemm kernel, not
available here); its documented contract is y = x @ dequant(W)^T with
dequant = (q - 8) * scale + zero in bf16 inputs / fp32 accumulate, which is what
`int4_linear` below computes.  Pinned against the reference's own quantiser functions
(tests/golden/quant_*.npz).
"""
from __future__ import annotations

import torch


def get_group_qparams(w: torch.Tensor, n_bit: int = 4, groupsize: int = 128):
    """fast_quantize.py:70-86.  Returns (scales, zeros) as bf16 tensors of shape (N, K/groupsize)."""
    assert w.dim() == 2 and w.shape[-1] % groupsize == 0
    g = w.reshape(-1, groupsize)
    max_val = g.amax(dim=1, keepdim=True)
    min_val = g.amin(dim=1, keepdim=True)
    max_int = 2 ** n_bit - 1
    scales = (max_val - min_val).clamp(min=1e-6) / max_int
    zeros = min_val + scales * (2 ** (n_bit - 1))
    return (scales.to(torch.bfloat16).reshape(w.shape[0], -1), zeros.to(torch.bfloat16).reshape(w.shape[0], -1))


def group_quantize_from_qparams(w: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, n_bit: int = 4,
                                groupsize: int = 128) -> torch.Tensor:
    """fast_quantize.py:106-124.  int32 codes in [0, 15], same shape as w.

    Note the arithmetic dtype: `w` is bf16 when called from the quant handler (:301) and the
    bf16 scales/zeros are used as stored, so sub/div/round all happen in bf16 there; pass w in
    the dtype you want to emulate.
    """
    g = w.reshape(-1, groupsize)
    s = scales.reshape(-1, 1)
    z = zeros.reshape(-1, 1)
    min_val = z - s * (2 ** (n_bit - 1))
    q = g.sub(min_val).div(s).round().clamp_(0, 2 ** n_bit - 1).to(torch.int32)
    return q.reshape_as(w)


def pack_scales_and_zeros(scales: torch.Tensor, zeros: torch.Tensor) -> torch.Tensor:
    """fast_quantize.py:89-103: (N, G) x2 -> (G, N, 2) bf16."""
    return torch.stack([scales, zeros], dim=2).transpose(0, 1).contiguous()


def group_quantize_tensor(w: torch.Tensor, n_bit: int = 4, groupsize: int = 128):
    """fast_quantize.py:127-131."""
    scales, zeros = get_group_qparams(w, n_bit, groupsize)
    q = group_quantize_from_qparams(w, scales, zeros, n_bit, groupsize)
    return q, pack_scales_and_zeros(scales, zeros)


def group_dequantize(q: torch.Tensor, scales_and_zeros: torch.Tensor, n_bit: int = 4, groupsize: int = 128,
                     out_dtype=torch.float32) -> torch.Tensor:
    """fast_quantize.py:134-147 from the packed (G, N, 2) layout: (q - 8) * scale + zero."""
    N, K = q.shape
    sz = scales_and_zeros.transpose(0, 1)  # (N, G, 2)
    s = sz[..., 0].to(out_dtype).reshape(-1, 1)
    z = sz[..., 1].to(out_dtype).reshape(-1, 1)
    g = q.reshape(-1, groupsize).to(out_dtype)
    return g.sub(2 ** (n_bit - 1)).mul(s).add(z).reshape(N, K)


def int4_linear(x: torch.Tensor, q: torch.Tensor, scales_and_zeros: torch.Tensor, groupsize: int = 128,
                round_weight_bf16: bool = True) -> torch.Tensor:
    """Contract of `linear_forward_int4` (fast_quantize.py:210-216): x (M,K) @ dequant(W (N,K))^T.

    tinygemm dequantises each weight to bf16 before the bf16 MMA, hence `round_weight_bf16`.
    """
    wd = group_dequantize(q, scales_and_zeros, groupsize=groupsize)
    if round_weight_bf16:
        wd = wd.to(torch.bfloat16).to(torch.float32)
    return x.float() @ wd.T


def quantize_per_channel_int8(x: torch.Tensor):
    """fast_quantize.py:38-67 with quant_min=-128, quant_max=127.  Returns (int8 codes, scales)."""
    eps = torch.finfo(torch.float32).eps
    min_val, max_val = torch.aminmax(x, dim=1)
    min_neg = torch.min(min_val, torch.zeros_like(min_val))
    max_pos = torch.max(max_val, torch.zeros_like(max_val))
    max_pos = torch.max(-min_neg, max_pos)
    scales = torch.clamp(max_pos / (float(127 - (-128)) / 2), min=eps).to(x.dtype)
    q = torch.clamp(torch.round(x / scales.unsqueeze(-1)), -128, 127).to(torch.int8)
    return q, scales


def int8_linear(x: torch.Tensor, q: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """fast_quantize.py:197-198: F.linear(x, W.to(x.dtype)) * scales."""
    return (x.float() @ q.float().T) * scales.float()


def int4_eligible(out_features: int, in_features: int, groupsize: int = 128, inner_k_tiles: int = 8) -> bool:
    """fast_quantize.py:219-226: out % 8 == 0 and K % groupsize == 0 and K % (inner_k_tiles*16) == 0."""
    return out_features % 8 == 0 and in_features % groupsize == 0 and in_features % (inner_k_tiles * 16) == 0
