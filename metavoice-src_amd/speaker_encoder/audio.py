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


class UnsupportedAudioFormat(ValueError):
    """The file is not RIFF/WAVE and no decoder for it is available in this process (serving.py answers 415 for it)."""


def _decode_with_whatever_is_installed(path: str):
    """-> (rate, float32 samples [n] or [n, channels]) through the first decoder present: `soundfile` (libsndfile: FLAC / OGG / AIFF and,
    in recent builds, MP3), `audioread` (what librosa falls back to), an `ffmpeg` executable on PATH (what the reference's serving
    path uses, `serving.py:112-124`).  None of them is in the build image; on a host that has one the reference's own assets
    (`assets/*.mp3`) load unchanged."""
    import shutil
    import subprocess

    errors = []
    try:
        import soundfile  # type: ignore

        data, rate = soundfile.read(path, dtype="float32", always_2d=False)
        return int(rate), np.asarray(data, dtype=np.float32)
    except ImportError:
        errors.append("soundfile not installed")
    except Exception as e:  # libsndfile could not parse it
        errors.append(f"soundfile: {e}")
    try:
        import audioread  # type: ignore

        with audioread.audio_open(path) as f:
            rate, ch = int(f.samplerate), int(f.channels)
            pcm = np.frombuffer(b"".join(f), dtype="<i2").astype(np.float32) / 32768.0
        return rate, (pcm.reshape(-1, ch) if ch > 1 else pcm)
    except ImportError:
        errors.append("audioread not installed")
    except Exception as e:
        errors.append(f"audioread: {e}")
    ffmpeg = shutil.which("ffmpeg")
    if ffmpeg:
        p = subprocess.run([ffmpeg, "-v", "error", "-i", path, "-f", "f32le", "-ac", "1", "-ar", str(sampling_rate), "-"],
                           capture_output=True, timeout=120)
        if p.returncode == 0 and p.stdout:
            return sampling_rate, np.frombuffer(p.stdout, dtype="<f4").copy()
        errors.append(f"ffmpeg: {p.stderr.decode('utf-8', 'replace')[:200]}")
    else:
        errors.append("no ffmpeg on PATH")
    raise UnsupportedAudioFormat(f"{path}: not a RIFF/WAVE file and no other decoder is available ({'; '.join(errors)}); "
                                 "convert it to .wav first (`ffmpeg -i in.mp3 out.wav`)")


def load_audio(path: str, sr: int = sampling_rate) -> np.ndarray:
    """Mono float32 waveform at `sr` (`librosa.load(fpath, sr=16000)` in the reference, `speaker_encoder/model.py:112-114`).  RIFF/WAVE
    (PCM 8/16/24/32-bit, float) is read through scipy whatever the file is called (serving.py stores every upload as `.wav`, like the
    reference); anything else goes to the first decoder this process has (`_decode_with_whatever_is_installed`), and raises
    `UnsupportedAudioFormat` when there is none."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly

    with open(path, "rb") as f:
        magic = f.read(12)
    if magic[:4] in (b"RIFF", b"RIFX", b"RF64") and magic[8:12] == b"WAVE":
        rate, data = wavfile.read(path)
    else:
        rate, data = _decode_with_whatever_is_installed(path)
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
