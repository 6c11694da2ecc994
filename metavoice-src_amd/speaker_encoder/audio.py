"""Mirror of `fam/quantiser/audio/speaker_encoder/audio.py` plus the audio reading / trimming `model.py:112-117` does through
librosa.  librosa (0.10.1 in the reference's requirements) is third party and not in this image, so the published algorithms of
the calls the reference makes are restated.  Round 5: the arithmetic runs on the GPU (csrc/audio_front.hip through `mv_audio_*`; SURVEY.md
8f row 3) -- the host keeps what is configuration (the mel filter bank, the Hann window, the resampling filter: built once, like the
vocoder's FIR banks) and file parsing; the numpy restatement this module used to BE is the checker now (oracle/audio_frontend.py):

  librosa.feature.melspectrogram(y, sr=16000, n_fft=400, hop_length=160, n_mels=40)      (audio.py:15-21)   -> mv_audio_mel
      = mel_filterbank(Slaney scale, Slaney area norm, fmin 0, fmax sr/2) @ |STFT|^2, periodic Hann window of n_fft samples,
        frames centred (n_fft // 2 zeros on both sides), NOT log-scaled
  librosa.load(path, sr=16000)                                                           (model.py:113)     -> host decode, mv_audio_resample
      = decode, mix down to mono, resample (soxr_hq there; a polyphase Kaiser resampler here: scipy.signal.resample_poly's design)
  librosa.effects.trim(y, top_db=20)                                                     (model.py:114)     -> mv_audio_frame_rms + two comparisons
      = frame RMS (frame 2048, hop 512, centred), dB relative to the loudest frame, keep from the first to the last frame above -20 dB
**Parity unpinned** for trim and the resampler (no librosa / soxr here); the mel front end is pinned through the oracle to transformers.
"""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np

from .. import _native as N

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


_FP = C.POINTER(C.c_float)


@functools.lru_cache(maxsize=None)
def _mel_setup(sr: int, n_fft: int, n_mels: int):
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32)  # periodic Hann
    return np.ascontiguousarray(window), np.ascontiguousarray(mel_filterbank(sr, n_fft, n_mels))


def wav_to_mel_spectrogram(wav: np.ndarray) -> np.ndarray:
    """`audio.py:10-22`: (frames, 40) float32 mel POWER spectrogram (not log), on the GPU (`mel_kernel`: one workgroup per centred frame)."""
    n_fft = int(sampling_rate * mel_window_length / 1000)
    hop = int(sampling_rate * mel_window_step / 1000)
    x = np.ascontiguousarray(np.asarray(wav, dtype=np.float32).reshape(-1))
    if x.size == 0:
        raise ValueError("empty waveform")
    window, fb = _mel_setup(sampling_rate, n_fft, mel_n_channels)
    n_frames = 1 + x.size // hop
    out = np.empty((n_frames, mel_n_channels), dtype=np.float32)
    N.check(N.lib().mv_audio_mel(x.ctypes.data_as(_FP), x.size, window.ctypes.data_as(_FP), fb.ctypes.data_as(_FP), n_fft, hop, mel_n_channels,
                                 out.ctypes.data_as(_FP), n_frames))
    return out


def trim_silence(y: np.ndarray, top_db: float = 20.0, frame_length: int = 2048, hop_length: int = 512) -> np.ndarray:
    """`librosa.effects.trim(y, top_db)`: centred frame RMS (GPU, `frame_rms_kernel`) in dB relative to its maximum; keep [first, last]
    non-silent frame."""
    y = np.asarray(y)
    if len(y) == 0:
        return y
    pad = frame_length // 2
    if len(y) + 2 * pad < frame_length:
        return y
    x = np.ascontiguousarray(y.astype(np.float32))
    n_frames = 1 + (x.size + 2 * pad - frame_length) // hop_length
    rms = np.empty(n_frames, dtype=np.float64)
    N.check(N.lib().mv_audio_frame_rms(x.ctypes.data_as(_FP), x.size, frame_length, hop_length, rms.ctypes.data_as(C.POINTER(C.c_double)), n_frames))
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


@functools.lru_cache(maxsize=32)
def resample_filter(up: int, down: int):
    """The FIR `scipy.signal.resample_poly(x, up, down)` designs (window ("kaiser", 5.0), 10 zero crossings on each side at the slower rate,
    unit DC gain, gain `up`), with the zero padding that aligns its output: -> (taps with padding, samples to drop in front of the upfirdn
    output).  Host configuration, like the vocoder's FIR banks; the convolution runs on the GPU."""
    max_rate = max(up, down)
    half_len = 10 * max_rate
    n = 2 * half_len + 1
    t = np.arange(n, dtype=np.float64) - half_len
    cutoff = 1.0 / max_rate                                  # firwin(n, cutoff) with fs = 2: sinc low-pass times the window, scaled to unit DC gain
    h = cutoff * np.sinc(cutoff * t) * np.kaiser(n, 5.0)
    h = h / h.sum() * up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    return np.ascontiguousarray(np.concatenate([np.zeros(n_pre_pad), h]).astype(np.float32)), int(n_pre_remove)


def resample(x: np.ndarray, rate: int, sr: int = sampling_rate) -> np.ndarray:
    """Polyphase resampling `rate` -> `sr` on the GPU (`resample_kernel`): y[m] = sum_k h[(m + pre) down - k up] x[k], ceil(n up / down) samples."""
    g = int(np.gcd(int(rate), int(sr)))
    up, down = int(sr) // g, int(rate) // g
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1))
    if up == down or x.size == 0:
        return x
    h, pre = resample_filter(up, down)
    n_out = -(-x.size * up // down)
    y = np.empty(n_out, dtype=np.float32)
    N.check(N.lib().mv_audio_resample(x.ctypes.data_as(_FP), x.size, h.ctypes.data_as(_FP), h.size, up, down, pre, n_out, y.ctypes.data_as(_FP)))
    return y


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
        x = resample(x, int(rate), int(sr))
    return x
