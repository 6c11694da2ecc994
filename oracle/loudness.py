"""TEST INFRASTRUCTURE ONLY (never imported by the product package): loudness normalisation of the saved waveform.

What the reference does: `audiocraft.data.audio.audio_write(..., strategy="loudness", loudness_compressor=True)`
(`fam/llm/decoders.py:40-47`) = audiocraft 1.2.0 `normalize_loudness` (headroom 14 dB, energy floor 2e-3, tanh compressor) on
`torchaudio.functional.loudness`, i.e. ITU-R BS.1770-4 integrated loudness.  Neither package is in this image or under
/root/reference, so this is a restatement of the published algorithms, written independently of the product's
`metavoice-src_amd/loudness.py` (plain sample-by-sample biquad recursions in float64, explicit block loops -- no scipy, no cumulative
sums) and pinned to what CAN be pinned here: the analytic cases of ITU-R BS.1770-4 / EBU Tech 3341 (tests/test_loudness_cpu.py:
a 997 Hz sine at -6.02 dBFS reads -9.03 LKFS in one channel; 1 kHz at -23 dBFS in both channels reads -23.0 LUFS; the -36 / -23 / -36
relative-gate sequence reads -23.0 LUFS; tolerance 0.1 LU as in Tech 3341).  Not pinned to torchaudio's bits: "parity unpinned"
against the third-party code itself stays true.

K-weighting as torchaudio 2.x evaluates it: `treble_biquad(x, sr, gain=4.0, central_freq=1500, Q=1/sqrt 2)` then
`highpass_biquad(x, sr, 38.0, Q=0.5)` -- the sox / RBJ cookbook sections recomputed for the sample rate -- each followed by
lfilter's default clamp to [-1, 1].
"""
import math

import numpy as np


def _section(x, b, a, clamp=True):
    """y[n] = (b0 x[n] + b1 x[n-1] + b2 x[n-2] - a1 y[n-1] - a2 y[n-2]) / a0, per channel, float64."""
    b0, b1, b2 = (v / a[0] for v in b)
    a1, a2 = a[1] / a[0], a[2] / a[0]
    y = np.zeros_like(x, dtype=np.float64)
    for c in range(x.shape[0]):
        x1 = x2 = y1 = y2 = 0.0
        xc, yc = x[c], y[c]
        for n in range(xc.shape[0]):
            v = b0 * xc[n] + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2
            x2, x1 = x1, xc[n]
            y2, y1 = y1, v
            yc[n] = v
    return np.clip(y, -1.0, 1.0) if clamp else y


def k_weight(x, sample_rate):
    # RBJ high shelf, +4 dB, 1500 Hz, Q = 1/sqrt(2)
    w0 = 2.0 * math.pi * 1500.0 / sample_rate
    A = 10.0 ** (4.0 / 40.0)
    al = math.sin(w0) / (2.0 / math.sqrt(2.0))
    cs, sq = math.cos(w0), 2.0 * math.sqrt(A) * al
    shelf_b = (A * ((A + 1) + (A - 1) * cs + sq), -2 * A * ((A - 1) + (A + 1) * cs), A * ((A + 1) + (A - 1) * cs - sq))
    shelf_a = ((A + 1) - (A - 1) * cs + sq, 2 * ((A - 1) - (A + 1) * cs), (A + 1) - (A - 1) * cs - sq)
    x = _section(np.asarray(x, dtype=np.float64), shelf_b, shelf_a)
    # RBJ high-pass, 38 Hz, Q = 0.5
    w0 = 2.0 * math.pi * 38.0 / sample_rate
    al = math.sin(w0) / (2.0 * 0.5)
    cs = math.cos(w0)
    return _section(x, ((1 + cs) / 2, -(1 + cs), (1 + cs) / 2), (1 + al, -2 * cs, 1 - al))


def integrated_loudness(wav, sample_rate):
    """BS.1770-4: 400 ms blocks, 75 % overlap, absolute gate -70 LKFS, relative gate 10 LU below the absolutely-gated mean."""
    x = np.asarray(wav, dtype=np.float64)
    if x.ndim == 1:
        x = x[None]
    block, hop = int(round(0.4 * sample_rate)), int(round(0.1 * sample_rate))
    if x.shape[-1] < block:
        return float("-inf")
    y = k_weight(x, sample_rate)
    weights = [1.0, 1.0, 1.0, 1.41, 1.41][: y.shape[0]]
    z = []   # per block: channel-weighted mean square
    start = 0
    while start + block <= y.shape[-1]:
        z.append(sum(g * float(np.mean(y[c, start:start + block] ** 2)) for c, g in enumerate(weights)))
        start += hop
    z = np.array(z)
    lk = np.full(z.shape, -np.inf)
    lk[z > 0] = -0.691 + 10.0 * np.log10(z[z > 0])
    absolute = lk > -70.0
    if not absolute.any():
        return float("-inf")
    gamma = -0.691 + 10.0 * math.log10(float(z[absolute].mean())) - 10.0
    final = absolute & (lk > gamma)
    if not final.any():
        return float("-inf")
    return -0.691 + 10.0 * math.log10(float(z[final].mean()))


def normalize_loudness(wav, sample_rate, loudness_headroom_db=14.0, loudness_compressor=True, energy_floor=2e-3):
    """audiocraft 1.2.0 `normalize_loudness`: near-silent input unchanged; gain to -headroom LKFS; optional tanh compressor."""
    x = np.asarray(wav, dtype=np.float64)
    if math.sqrt(float(np.mean(x ** 2))) < energy_floor:
        return x
    lk = integrated_loudness(x, sample_rate)
    if not math.isfinite(lk):
        return x
    out = x * 10.0 ** ((-loudness_headroom_db - lk) / 20.0)
    return np.tanh(out) if loudness_compressor else out
