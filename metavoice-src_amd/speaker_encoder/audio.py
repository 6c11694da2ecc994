"""Mirror of `fam/quantiser/audio/speaker_encoder/audio.py` plus the audio reading / trimming `model.py:112-117` does through
librosa.  librosa (0.10.1 in the reference's requirements) is third party and not in this image, so the published algorithms of
the calls the reference makes are restated in numpy -- host pre-processing, as in the reference; **parity unpinned** (no
librosa here to mint vectors from; tests/test_speaker_encoder_cpu.py checks the maths against independent closed forms):

  librosa.feature.melspectrogram(y, sr=16000, n_fft=400, hop_length=160, n_mels=40)      (audio.py:15-21)
      = mel_filterbank(Slaney scale, Slaney area norm, fmin 0, fmax sr/2) @ |STFT|^2, periodic Hann window of n_fft samples,
        frames centred (n_fft // 2 zeros on both sides), NOT log-scaled
  librosa.load(path, sr=16000)                                                           (model.py:113)
      = decode, mix down to mono, resample (soxr_hq there; a polyphase Kaiser resampler here)
  librosa.effects.trim(y, top_db=20)                                                     (model.py:114)
      = frame RMS (frame 2048, hop 512, centred), dB relative to the loudest frame, keep from the first to the last frame above -20 dB
"""
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


def load_audio(path: str, sr: int = sampling_rate) -> np.ndarray:
    """Mono float32 waveform at `sr`.  Reads RIFF/WAVE (PCM 8/16/24/32-bit, float) through scipy; compressed formats (the
    reference's assets/*.mp3) need a decoder this image does not have -- convert them to .wav first (`ffmpeg -i in.mp3 out.wav`)."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly

    try:
        rate, data = wavfile.read(path)
    except ValueError as e:
        raise ValueError(f"{path}: only RIFF/WAVE files can be read here ({e}); convert other formats to .wav first") from e
    if data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(data.dtype, np.integer):
        x = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    else:
        x = data.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    if rate != sr:
        g = int(np.gcd(int(rate), int(sr)))
        x = resample_poly(x, sr // g, int(rate) // g).astype(np.float32)
    return x
