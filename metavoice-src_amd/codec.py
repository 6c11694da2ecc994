"""EnCodec 24 kHz decode on the MI355X engine (`mv_codec_*`, csrc/codec.hip): the codec half of the reference's vocoder call
(`fam/llm/decoders.py:84-85` -> audiocraft `MultiBandDiffusion.tokens_to_wav` -> EnCodec `decode` / `decode_latent`).

`EncodecDecoder.from_state_dict` accepts the state dict of a `transformers.EncodecModel` (or any dict with the same names);
weight-norm parametrisations (`...parametrizations.weight.original0/1` or the older `weight_g` / `weight_v`) are folded on the
host once.  `decode(tokens)` has the vocoder-hook signature `TTS(..., vocoder=...)` expects: (1, n_q, frames) -> (1, 1, samples).
This is EnCodec's own decoder; the multi-band diffusion refinement of the reference is not built yet.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Sequence

import numpy as np
import torch

from . import _native as N


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """-> plain `...conv.weight` tensors.  weight = g * v / ||v|| with the norm taken over every dim but 0 (torch's default
    `dim=0`, which for ConvTranspose1d is the INPUT-channel axis)."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.endswith("parametrizations.weight.original1") or k.endswith("weight_v"):
            base = k[: -len("parametrizations.weight.original1")] if k.endswith("original1") else k[: -len("weight_v")]
            g = sd[base + ("parametrizations.weight.original0" if k.endswith("original1") else "weight_g")].float()
            vf = v.float()
            norm = vf.reshape(vf.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (vf.dim() - 1)))
            out[base + "weight"] = g * vf / norm
        elif k.endswith("parametrizations.weight.original0") or k.endswith("weight_g"):
            continue
        else:
            out[k] = v
    return out


class EncodecDecoder:
    def __init__(self, *, hidden=128, num_filters=32, ratios: Sequence[int] = (8, 5, 4, 2), kernel_size=7, last_kernel_size=7,
                 residual_kernel_size=3, compress=2, lstm_layers=2, n_q=8, bins=1024, audio_channels=1):
        cfg = N.mv_codec_config()
        cfg.hidden, cfg.num_filters, cfg.n_ratios = hidden, num_filters, len(ratios)
        for i, r in enumerate(ratios):
            cfg.ratios[i] = int(r)
        cfg.kernel_size, cfg.last_kernel_size, cfg.residual_kernel_size = kernel_size, last_kernel_size, residual_kernel_size
        cfg.compress, cfg.lstm_layers, cfg.n_q, cfg.bins, cfg.audio_channels = compress, lstm_layers, n_q, bins, audio_channels
        self.hidden, self.n_q, self.sample_rate = hidden, n_q, 24_000
        self._h = C.c_void_p()
        N.check(N.lib().mv_codec_create(C.byref(cfg), C.byref(self._h)))

    @classmethod
    def from_state_dict(cls, sd: Dict[str, torch.Tensor], n_q: int = 8, **kw) -> "EncodecDecoder":
        dec = cls(n_q=n_q, **kw)
        for name, t in fold_weight_norm(sd).items():
            if name.startswith("encoder.") or "cluster_size" in name or "embed_avg" in name or "inited" in name:
                continue
            if name.startswith("quantizer.layers."):
                if int(name.split(".")[2]) >= n_q:
                    continue
            a = np.ascontiguousarray(t.detach().float().cpu().numpy())
            N.check(N.lib().mv_codec_load_weight(dec._h, name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size))
        N.check(N.lib().mv_codec_finalize(dec._h))
        return dec

    def close(self):
        if getattr(self, "_h", None):
            N.lib().mv_codec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode(self, tokens: torch.Tensor, return_latent: bool = False):
        """tokens (1, n_q', frames) with n_q' <= n_q -> wav (1, 1, frames * 320) fp32 [, latent (1, hidden, frames)]."""
        assert tokens.dim() == 3 and tokens.shape[0] == 1
        nq, frames = int(tokens.shape[1]), int(tokens.shape[2])
        tok = np.ascontiguousarray(tokens[0].detach().cpu().numpy().astype(np.int32))
        n = int(N.lib().mv_codec_samples(self._h, frames))
        wav = np.empty((1, n), dtype=np.float32)
        lat = np.empty((self.hidden, frames), dtype=np.float32) if return_latent else None
        N.check(N.lib().mv_codec_decode(self._h, tok.ctypes.data_as(C.POINTER(C.c_int32)), nq, frames,
                                        wav.ctypes.data_as(C.POINTER(C.c_float)),
                                        lat.ctypes.data_as(C.POINTER(C.c_float)) if lat is not None else None))
        w = torch.from_numpy(wav).unsqueeze(0)
        return (w, torch.from_numpy(lat).unsqueeze(0)) if return_latent else w

    def decode_device(self, tokens: torch.Tensor):
        """The same decode with the results left in the codec's device workspace: -> (wav pointer, latent pointer, samples, frames),
        raw device addresses valid until the next call on this codec -- what `mbd.MultiBandDiffusionBands.tokens_to_wav_device`
        consumes (the reference hands the codec's outputs to the diffusion bands on the GPU as well, `fam/llm/decoders.py:84-85`)."""
        assert tokens.dim() == 3 and tokens.shape[0] == 1
        nq, frames = int(tokens.shape[1]), int(tokens.shape[2])
        tok = np.ascontiguousarray(tokens[0].detach().cpu().numpy().astype(np.int32))
        d_wav, d_lat = C.c_void_p(), C.c_void_p()
        N.check(N.lib().mv_codec_decode_device(self._h, tok.ctypes.data_as(C.POINTER(C.c_int32)), nq, frames, C.byref(d_wav), C.byref(d_lat)))
        return d_wav.value, d_lat.value, int(N.lib().mv_codec_samples(self._h, frames)), frames

    __call__ = decode
