"""Oracle for the EnCodec 24 kHz token -> waveform decode.  TEST INFRASTRUCTURE.

The reference reaches this arithmetic through third-party code that is neither vendored under /root/reference nor installed
here: `fam/llm/decoders.py:13,84-85` -> `audiocraft==1.2.0` `MultiBandDiffusion.tokens_to_wav` -> `encodec==0.1.1`
(`poetry.lock`).  This file restates the PUBLISHED algorithm (Defossez et al. 2022; SEANet decoder + residual VQ) in plain
torch functional calls on a flat dict of tensors.  Pinning: `tests/test_codec_cpu.py` checks it against the independent
implementation that IS installed, `transformers.models.encodec.EncodecModel`, on seeded random weights (exact module graph,
max |diff| ~1e-6).  Against the reference's own call (`mbd.tokens_to_wav`) parity is UNPINNED: audiocraft cannot run here, and
the diffusion refinement on top of this codec is not restated yet.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def _causal_conv1d(x, w, b, dilation=1):
    """EncodecConv1d with use_causal_conv, stride 1, pad_mode="reflect": left-pad by (k-1)*dilation."""
    pad = (w.shape[-1] - 1) * dilation
    if pad > 0:
        x = F.pad(x, (pad, 0), mode="reflect")
    return F.conv1d(x, w, b, dilation=dilation)


def _causal_conv_transpose1d(x, w, b, stride):
    """EncodecConvTranspose1d, causal, trim_right_ratio = 1: everything trimmed on the right."""
    y = F.conv_transpose1d(x, w, b, stride=stride)
    trim = w.shape[-1] - stride
    return y[..., : y.shape[-1] - trim]


def decode(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, ratios: Sequence[int] = (8, 5, 4, 2), lstm_layers: int = 2):
    """sd: folded weights with transformers' names (see metavoice-src_amd/codec.py: fold_weight_norm); tokens (n_q, T) long.
    Returns (wav (1, 1, T * prod(ratios)), latent (1, 128, T))."""
    latent = sum(F.embedding(tokens[q], sd[f"quantizer.layers.{q}.codebook.embed"]) for q in range(tokens.shape[0]))
    latent = latent.t().unsqueeze(0)  # (1, 128, T)
    x = _causal_conv1d(latent, sd["decoder.layers.0.conv.weight"], sd["decoder.layers.0.conv.bias"])
    # EncodecLSTM: y = lstm(x) + x
    seq = x[0].t()  # (T, H)
    inp = seq
    for l in range(lstm_layers):
        p = "decoder.layers.1.lstm."
        wih, whh = sd[p + f"weight_ih_l{l}"], sd[p + f"weight_hh_l{l}"]
        bias = sd[p + f"bias_ih_l{l}"] + sd[p + f"bias_hh_l{l}"]
        H = whh.shape[1]
        h = torch.zeros(H)
        c = torch.zeros(H)
        outs = []
        gin = inp @ wih.t() + bias
        for t in range(inp.shape[0]):
            g = gin[t] + whh @ h
            i, f, gg, o = torch.sigmoid(g[:H]), torch.sigmoid(g[H:2 * H]), torch.tanh(g[2 * H:3 * H]), torch.sigmoid(g[3 * H:])
            c = f * c + i * gg
            h = o * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs)
    x = (inp + seq).t().unsqueeze(0)
    idx = 2
    for r in ratios:
        x = F.elu(x)
        x = _causal_conv_transpose1d(x, sd[f"decoder.layers.{idx + 1}.conv.weight"], sd[f"decoder.layers.{idx + 1}.conv.bias"], r)
        p = f"decoder.layers.{idx + 2}."
        y = _causal_conv1d(F.elu(x), sd[p + "block.1.conv.weight"], sd[p + "block.1.conv.bias"])
        y = _causal_conv1d(F.elu(y), sd[p + "block.3.conv.weight"], sd[p + "block.3.conv.bias"])
        x = _causal_conv1d(x, sd[p + "shortcut.conv.weight"], sd[p + "shortcut.conv.bias"]) + y
        idx += 3
    x = F.elu(x)
    wav = _causal_conv1d(x, sd[f"decoder.layers.{idx + 1}.conv.weight"], sd[f"decoder.layers.{idx + 1}.conv.bias"])
    return wav, latent
