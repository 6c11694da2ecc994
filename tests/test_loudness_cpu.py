"""Loudness (ITU-R BS.1770-4) of the post-processing step (`fam/llm/decoders.py:40-47` -> audiocraft `audio_write(strategy="loudness")`):
the product's `loudness.py` and the independently written `oracle/loudness.py` against the standard's analytic cases (EBU Tech 3341
tolerance: 0.1 LU) and against each other.  audiocraft / torchaudio are not in the image: this pins the restatements to the
published standard, not to the third-party code's bits."""
import math

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import loudness as o_loud

loudness = load_pkg("loudness")


def tone(freq, dbfs, seconds, sr, channels=1):
    t = np.arange(int(seconds * sr)) / sr
    x = (10.0 ** (dbfs / 20.0)) * np.sin(2 * np.pi * freq * t)
    return np.tile(x[None], (channels, 1)).astype(np.float32)


CASES = [
    # BS.1770-4 annex: a 997 Hz sine in ONE front channel reads its dBFS level - 3.01 (0 dBFS -> -3.01 LKFS); half scale here because
    # torchaudio's biquads clamp to [-1, 1] and the shelf lifts a full-scale tone above 1
    ("997 Hz, -6.02 dBFS, mono, 48 kHz", lambda: tone(997.0, -6.0206, 5.0, 48000), 48000, -9.03),
    # EBU Tech 3341 case 1 / 2: stereo 1 kHz sine at -23 / -33 dBFS in both channels -> -23.0 / -33.0 LUFS
    ("1 kHz, -23 dBFS, stereo, 48 kHz", lambda: tone(1000.0, -23.0, 20.0, 48000, 2), 48000, -23.0),
    ("1 kHz, -33 dBFS, stereo, 48 kHz", lambda: tone(1000.0, -33.0, 20.0, 48000, 2), 48000, -33.0),
    # the engine's own rate (24 kHz mono): -20 dBFS in one channel -> -23.01
    ("997 Hz, -20 dBFS, mono, 24 kHz", lambda: tone(997.0, -20.0, 5.0, 24000), 24000, -23.01),
    # Tech 3341 case 3: -36 dBFS 10 s, -23 dBFS 60 s, -36 dBFS 10 s -> the relative gate drops the quiet parts: -23.0
    ("relative gate: -36 / -23 / -36 dBFS", lambda: np.concatenate([tone(1000.0, -36.0, 10.0, 48000, 2), tone(1000.0, -23.0, 60.0, 48000, 2),
                                                                    tone(1000.0, -36.0, 10.0, 48000, 2)], axis=1), 48000, -23.0),
    # Tech 3341 case 4 (absolute gate): -72 / -36 / -72 dBFS -> the -72 dBFS blocks fall under -70 LKFS: -36.0
    ("absolute gate: -72 / -36 / -72 dBFS", lambda: np.concatenate([tone(1000.0, -72.0, 10.0, 48000, 2), tone(1000.0, -36.0, 60.0, 48000, 2),
                                                                    tone(1000.0, -72.0, 10.0, 48000, 2)], axis=1), 48000, -36.0),
]


@pytest.mark.parametrize("name,make,sr,want", CASES, ids=[c[0] for c in CASES])
def test_integrated_loudness_meets_the_standards_analytic_cases(name, make, sr, want):
    x = make()
    got_product = loudness.integrated_loudness(torch.from_numpy(x), sr)
    assert abs(got_product - want) <= 0.1, (name, got_product, want)
    # the oracle's sample-by-sample recursion is slow in Python: it gets the first 3 s of the stationary cases and a 1:8 shortened
    # gate sequence (same structure: quiet, loud, quiet), on which it must agree with the product to 1e-3 LU
    if "gate" in name:
        n = x.shape[1]
        xo = np.concatenate([x[:, : n // 64], x[:, n // 2 - n // 16: n // 2 + n // 16], x[:, -(n // 64):]], axis=1)
        assert abs(o_loud.integrated_loudness(xo, sr) - loudness.integrated_loudness(torch.from_numpy(xo), sr)) <= 1e-3
    else:
        xo = x[:, : 3 * sr]
        got_oracle = o_loud.integrated_loudness(xo, sr)
        assert abs(got_oracle - want) <= 0.1, (name, got_oracle, want)
        assert abs(got_oracle - loudness.integrated_loudness(torch.from_numpy(xo), sr)) <= 1e-3


def test_product_and_oracle_normalisation_agree_on_a_speech_like_signal():
    sr = 24000
    rng = np.random.default_rng(0)
    t = np.arange(int(2.0 * sr)) / sr
    env = (0.5 + 0.5 * np.sin(2 * np.pi * 3.0 * t)) * (t > 0.2)
    x = (0.2 * env * (np.sin(2 * np.pi * 180.0 * t) + 0.5 * np.sin(2 * np.pi * 1230.0 * t)) + 0.01 * rng.standard_normal(t.size)).astype(np.float32)
    lp, lo = loudness.integrated_loudness(torch.from_numpy(x), sr), o_loud.integrated_loudness(x, sr)
    assert math.isfinite(lp) and abs(lp - lo) < 1e-4
    yp = loudness.normalize_loudness(torch.from_numpy(x)[None], sr).numpy()[0]
    yo = o_loud.normalize_loudness(x[None], sr)[0]
    assert np.abs(yp - yo).max() < 2e-6
    # the result sits at -14 LKFS before the compressor (audiocraft's headroom)
    pre = loudness.normalize_loudness(torch.from_numpy(x)[None], sr, loudness_compressor=False)
    assert abs(loudness.integrated_loudness(pre, sr) + 14.0) < 0.05
    # near-silence and clips shorter than one gating block are returned unchanged (energy floor 2e-3; -inf loudness)
    quiet = torch.from_numpy(1e-4 * x)[None]
    assert torch.equal(loudness.normalize_loudness(quiet, sr), quiet)
    assert np.array_equal(o_loud.normalize_loudness(1e-4 * x[None], sr), (1e-4 * x[None]).astype(np.float64))
