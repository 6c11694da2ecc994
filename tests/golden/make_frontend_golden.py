#!/usr/bin/env python3
"""Mints tests/golden/mel_frontend.npz: the speaker encoder's mel front end (`fam/quantiser/audio/speaker_encoder/audio.py:15-21`:
`librosa.feature.melspectrogram(y, sr=16000, n_fft=400, hop_length=160, n_mels=40)`) evaluated by an implementation that is
INDEPENDENT of this repository and installed in the image: `transformers.audio_utils` -- `mel_filter_bank(norm="slaney",
mel_scale="slaney")` is librosa's default filter bank, `spectrogram(center=True, pad_mode="constant", power=2.0)` with a periodic Hann
window of 400 samples is librosa's default STFT power spectrogram (librosa >= 0.10 pads with zeros).  librosa itself is not in the
image; transformers' audio_utils is documented as its numpy equivalent and is what transformers' own feature extractors (Whisper, CLAP)
are validated against.

    python tests/golden/make_frontend_golden.py          (re-creates the file byte for byte; needs only numpy + transformers)
"""
import os

import numpy as np
from transformers.audio_utils import mel_filter_bank, spectrogram, window_function

SR, N_FFT, HOP, N_MELS = 16000, 400, 160, 40


def test_waveform(seed: int = 0, seconds: float = 1.5) -> np.ndarray:
    """Deterministic, integer-arithmetic-only recipe (no library RNG whose stream could change): a chirp 80 Hz -> 7 kHz, two tones and
    an LCG noise floor, float32."""
    n = int(SR * seconds)
    t = np.arange(n, dtype=np.float64) / SR
    chirp = np.sin(2 * np.pi * (80.0 * t + 0.5 * (7000.0 - 80.0) / seconds * t * t))
    tones = 0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.25 * np.sin(2 * np.pi * 3000.0 * t + 0.3)
    state = np.uint64(seed * 2654435761 + 12345)
    noise = np.empty(n, dtype=np.float64)
    for i in range(n):  # 64-bit LCG (Knuth's MMIX constants), top 24 bits -> [-1, 1)
        state = np.uint64((int(state) * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF)
        noise[i] = ((int(state) >> 40) / float(1 << 23)) - 1.0
    env = 0.5 - 0.5 * np.cos(2 * np.pi * np.minimum(t / seconds, 1.0))
    return (0.3 * env * chirp + 0.2 * tones * (t > 0.4) + 0.01 * noise).astype(np.float32)


def main():
    wav = test_waveform()
    fb = mel_filter_bank(num_frequency_bins=1 + N_FFT // 2, num_mel_filters=N_MELS, min_frequency=0.0, max_frequency=SR / 2.0,
                         sampling_rate=SR, norm="slaney", mel_scale="slaney")           # (201, 40)
    win = window_function(N_FFT, "hann", periodic=True)
    mel = spectrogram(wav, win, frame_length=N_FFT, hop_length=HOP, fft_length=N_FFT, power=2.0, center=True, pad_mode="constant",
                      mel_filters=fb, mel_floor=0.0)                                     # (40, frames)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mel_frontend.npz")
    np.savez_compressed(out, filter_bank=fb.T.astype(np.float32), mel=mel.T.astype(np.float32), seed=np.int64(0), seconds=np.float64(1.5),
                        source=np.array("transformers.audio_utils mel_filter_bank(norm=slaney, mel_scale=slaney) + spectrogram(center, constant pad, power 2)"))
    print("wrote", out, "filter bank", fb.T.shape, "mel", mel.T.shape)


if __name__ == "__main__":
    main()
