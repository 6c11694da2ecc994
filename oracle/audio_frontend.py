"""CPU restatement of the speaker-reference front end -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import oracle/; the product package never does: tests/test_abi_cpu.py).

What `SpeakerEncoder.embed_utterance_from_file` does to the audio before the LSTM (fam/quantiser/audio/speaker_encoder/model.py:112-117,
audio.py:10-22), through librosa 0.10.1 (third party, not in the image), restated in numpy from the published algorithms:

  librosa.feature.melspectrogram(y, sr=16000, n_fft=400, hop_length=160, n_mels=40)      (audio.py:15-21)
      = mel_filterbank(Slaney scale, Slaney area norm, fmin 0, fmax sr/2) @ |STFT|^2, periodic Hann window of n_fft samples,
        frames centred (n_fft // 2 zeros on both sides), NOT log-scaled
  librosa.effects.trim(y, top_db=20)                                                     (model.py:114)
      = frame RMS (frame 2048, hop 512, centred), dB relative to the loudest frame, keep from the first to the last frame above -20 dB
  librosa.load(path, sr=16000) resamples with soxr_hq; here, as in the product, a polyphase Kaiser resampler = scipy.signal.resample_poly

Pins: the mel front end against `transformers.audio_utils` (tests/golden/mel_frontend.npz + make_frontend_golden.py), the filter bank and the
mel scale against closed forms (tests/test_speaker_encoder_cpu.py).  librosa's trim and resampler have no comparator in the image: **parity
unpinned** for those two (the product's GPU kernels are held to THIS restatement and to scipy, tests/test_audio_front_gpu.py).
Until round 4 these functions WERE the product path (host numpy); round 5 moved the arithmetic to csrc/audio_front.hip and this file here."""
from __future__ import annotations

import numpy as np

mel_window_length = 25  # ms
mel_window_step = 10    # ms
mel_n_channels = 40
sampling_rate = 16000

def _hz_to_mel(f):
    """Slaney (auditory toolbox) scale, librosa's default (htk=False): linear below 1 kHz, logarithmic above."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax=None) -> np.ndarray:
    """(n_mels, 1 + n_fft // 2) triangular filters, each normalised to unit area in Hz (`norm="slaney"`)."""
    fmax = sr / 2.0 if fmax is None else fmax
    fft_freqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


def _frames(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    n = 1 + (len(y) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    return y[idx]


def wav_to_mel_spectrogram(wav: np.ndarray) -> np.ndarray:
    """`audio.py:10-22`: (frames, 40) float32 mel POWER spectrogram (not log)."""
    n_fft = int(sampling_rate * mel_window_length / 1000)
    hop = int(sampling_rate * mel_window_step / 1000)
    y = np.pad(np.asarray(wav, dtype=np.float32), n_fft // 2, mode="constant")
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32)  # periodic Hann
    spec = np.fft.rfft(_frames(y, n_fft, hop) * window[None, :], n=n_fft, axis=1)
    power = (spec.real ** 2 + spec.imag ** 2).astype(np.float32)
    return (power @ mel_filterbank(sampling_rate, n_fft, mel_n_channels).T).astype(np.float32)


def trim_silence(y: np.ndarray, top_db: float = 20.0, frame_length: int = 2048, hop_length: int = 512) -> np.ndarray:
    """`librosa.effects.trim(y, top_db)`: centred frame RMS in dB relative to its maximum; keep [first, last] non-silent frame."""
    if len(y) == 0:
        return y
    pad = frame_length // 2
    yp = np.pad(y.astype(np.float32), pad, mode="constant")
    if len(yp) < frame_length:
        return y
    rms = np.sqrt(np.mean(_frames(yp, frame_length, hop_length).astype(np.float64) ** 2, axis=1))
    ref = rms.max()
    if ref <= 0:
        return y[:0]
    db = 20.0 * np.log10(np.maximum(rms, 1e-5 * 1.0)) - 20.0 * np.log10(max(ref, 1e-5))
    keep = np.flatnonzero(db > -top_db)
    if keep.size == 0:
        return y[:0]
    start = int(keep[0]) * hop_length
    end = min(len(y), (int(keep[-1]) + 1) * hop_length)
    return y[start:end]


def resample(x: np.ndarray, rate: int, sr: int = sampling_rate) -> np.ndarray:
    """`scipy.signal.resample_poly` (zero-phase Kaiser-windowed polyphase FIR): the independent implementation the GPU resampler is held to."""
    from scipy.signal import resample_poly

    g = int(np.gcd(int(rate), int(sr)))
    return resample_poly(np.asarray(x, dtype=np.float32), sr // g, int(rate) // g).astype(np.float32)
