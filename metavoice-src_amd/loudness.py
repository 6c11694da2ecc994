"""Loudness normalisation of the synthesised waveform, in memory.

The reference saves through `audiocraft.data.audio.audio_write(..., strategy="loudness", loudness_compressor=True)`
(`fam/llm/decoders.py:40-47`), i.e. audiocraft 1.2.0 `normalize_loudness` (headroom 14 dB, energy floor 2e-3, tanh compressor)
on top of `torchaudio.functional.loudness` = ITU-R BS.1770-4 integrated loudness: K-weighting (a +4 dB treble shelf at 1.5 kHz,
Q 1/sqrt(2), then a 38 Hz high-pass, Q 0.5, both as sox-style biquads recomputed for the sample rate, each clamped to
[-1, 1] as torchaudio's `lfilter` does by default), 400 ms blocks with 75 % overlap, absolute gate -70 LKFS, relative gate
-10 LU.  Neither package is in this image or under /root/reference: this is a restatement of the published algorithm,
parity unpinned; it is host-side post-processing, not on the GPU path.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _biquad(x: np.ndarray, b0, b1, b2, a0, a1, a2) -> np.ndarray:
    from scipy.signal import lfilter
    y = lfilter(np.array([b0, b1, b2], dtype=np.float64) / a0, np.array([1.0, a1 / a0, a2 / a0], dtype=np.float64), x.astype(np.float64), axis=-1)
    return np.clip(y, -1.0, 1.0)  # torchaudio.functional.lfilter(clamp=True)


def _k_weight(x: np.ndarray, sample_rate: int) -> np.ndarray:
    # treble_biquad(gain=4.0, central_freq=1500.0, Q=1/sqrt(2))
    w0 = 2 * math.pi * 1500.0 / sample_rate
    A = math.exp(4.0 / 40.0 * math.log(10))
    alpha = math.sin(w0) / 2 / (1 / math.sqrt(2))
    t1, t2, t3 = 2 * math.sqrt(A) * alpha, (A - 1) * math.cos(w0), (A + 1) * math.cos(w0)
    x = _biquad(x, A * ((A + 1) + t2 + t1), -2 * A * ((A - 1) + t3), A * ((A + 1) + t2 - t1), (A + 1) - t2 + t1, 2 * ((A - 1) - t3), (A + 1) - t2 - t1)
    # highpass_biquad(cutoff_freq=38.0, Q=0.5)
    w0 = 2 * math.pi * 38.0 / sample_rate
    alpha = math.sin(w0) / 2 / 0.5
    c = math.cos(w0)
    return _biquad(x, (1 + c) / 2, -1 - c, (1 + c) / 2, 1 + alpha, -2 * c, 1 - alpha)


def integrated_loudness(wav: torch.Tensor, sample_rate: int) -> float:
    """BS.1770-4 integrated loudness (LKFS) of a (channels, samples) or (samples,) waveform; -inf for digital silence / clips
    shorter than one 400 ms block."""
    x = wav.detach().float().cpu().numpy()
    if x.ndim == 1:
        x = x[None]
    gate, step = int(round(0.4 * sample_rate)), int(round(0.4 * sample_rate * 0.25))
    if x.shape[-1] < gate:
        return float("-inf")
    y = _k_weight(x, sample_rate)
    n_blocks = (y.shape[-1] - gate) // step + 1
    sq = np.square(y)
    cs = np.concatenate([np.zeros((sq.shape[0], 1)), np.cumsum(sq, axis=-1)], axis=-1)
    starts = np.arange(n_blocks) * step
    energy = (cs[:, starts + gate] - cs[:, starts]) / gate  # (channels, blocks)
    g = np.array([1.0, 1.0, 1.0, 1.41, 1.41])[: energy.shape[0], None]
    with np.errstate(divide="ignore"):
        block_loudness = -0.691 + 10 * np.log10((g * energy).sum(axis=0))
        keep = block_loudness > -70.0
        if not keep.any():
            return float("-inf")
        gamma_rel = -0.691 + 10 * np.log10((g[:, 0] * energy[:, keep].mean(axis=-1)).sum()) - 10.0
        keep = keep & (block_loudness > gamma_rel)
        if not keep.any():
            return float("-inf")
        return float(-0.691 + 10 * np.log10((g[:, 0] * energy[:, keep].mean(axis=-1)).sum()))


def normalize_loudness(wav: torch.Tensor, sample_rate: int, loudness_headroom_db: float = 14.0, loudness_compressor: bool = True,
                       energy_floor: float = 2e-3) -> torch.Tensor:
    """audiocraft `normalize_loudness`: scale to -headroom LKFS, optional tanh compressor; near-silent input is returned as is."""
    energy = float(wav.float().pow(2).mean().sqrt())
    if energy < energy_floor:
        return wav
    lk = integrated_loudness(wav, sample_rate)
    if not math.isfinite(lk):
        return wav
    out = wav * (10.0 ** ((-loudness_headroom_db - lk) / 20.0))
    if loudness_compressor:
        out = torch.tanh(out)
    return out
