"""Speaker-reference front end on the GPU (csrc/audio_front.hip through `mv_audio_*`; metavoice-src_amd/speaker_encoder/audio.py) against
its CPU restatement (oracle/audio_frontend.py: numpy, pinned to transformers.audio_utils for the mel front end) and, for the resampler,
against scipy.signal.resample_poly -- fam/quantiser/audio/speaker_encoder/model.py:112-117, audio.py:10-22 through librosa 0.10.1 (third
party, absent: librosa's trim and soxr resampler themselves are unpinned, see the oracle's header).

Tolerances: the mel power spectrogram is fp32 with a direct 400-point DFT per frame against numpy's FFT -- 2e-5 of the spectrogram's
maximum; frame RMS is double on both sides (1e-9 relative), so the trimmed ranges are identical; the resampler is a 16-bit-audio FIR
accumulated in double and rounded once: 2e-6 absolute on signals of unit scale."""
import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import audio_frontend as o_audio
from oracle import speaker_encoder as o_spk

pytestmark = pytest.mark.gpu

audio = load_pkg("speaker_encoder.audio")
synthetic = load_pkg("synthetic")


def speechlike(n, seed, sr=16000):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    env = 0.5 + 0.5 * np.sin(2 * np.pi * 3.1 * t + rng.uniform(0, 6.28))
    x = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in zip((0.4, 0.25, 0.15, 0.1), (180.0, 410.0, 1230.0, 3100.0), rng.uniform(0, 6.28, 4)))
    return (env * x + 0.01 * rng.standard_normal(n)).astype(np.float32)


@pytest.mark.parametrize("n", [160, 399, 400, 8000, 48017, 480000])
def test_mel_spectrogram_matches_the_oracle(n):
    wav = speechlike(n, n)
    got, ref = audio.wav_to_mel_spectrogram(wav), o_audio.wav_to_mel_spectrogram(wav)
    assert got.shape == ref.shape == (1 + n // 160, 40) and got.dtype == np.float32
    d = float(np.abs(got - ref).max())
    print(f"[mel n={n}] max |d| {d:.3e} of max {float(ref.max()):.3e}")
    assert d <= 2e-5 * float(ref.max())


def test_trim_is_the_oracles():
    sr = 16000
    tone = speechlike(2 * sr, 3)
    for lead, tail in ((sr, sr // 2), (0, 0), (700, 20000), (5 * sr, 3)):
        y = np.concatenate([np.zeros(lead, np.float32), tone, 1e-4 * speechlike(tail, 4) if tail else np.zeros(0, np.float32)])
        got, ref = audio.trim_silence(y, top_db=20), o_audio.trim_silence(y, top_db=20)
        assert got.shape == ref.shape and np.array_equal(got, ref), (lead, tail, got.shape, ref.shape)
    assert audio.trim_silence(np.zeros(5000, np.float32)).size == 0
    assert audio.trim_silence(np.zeros(100, np.float32)).size == o_audio.trim_silence(np.zeros(100, np.float32)).size == 0   # one padded frame, silent
    short = speechlike(100, 8)
    assert np.array_equal(audio.trim_silence(short), o_audio.trim_silence(short))


@pytest.mark.parametrize("rate,n", [(44100, 88200), (48000, 30001), (8000, 12345), (22050, 40000), (24000, 7), (32000, 64000)])
def test_resampler_matches_scipy(rate, n):
    x = speechlike(n, rate, sr=rate)
    got, ref = audio.resample(x, rate, 16000), o_audio.resample(x, rate, 16000)
    assert got.shape == ref.shape and got.dtype == np.float32
    d = float(np.abs(got - ref).max())
    print(f"[resample {rate} -> 16000, n={n}] {len(got)} samples, max |d| {d:.3e}")
    assert d <= 2e-6


def test_embedding_of_a_file_goes_through_the_gpu_front_end(tmp_path):
    """`embed_utterance_from_file` (model.py:112-117) end to end: a 22.05 kHz stereo 16-bit file with silence around the speech -> host decode
    and mix-down, GPU resample, GPU frame RMS + trim, GPU mel, HIP LSTM -- against the oracle's front end feeding the oracle's encoder."""
    from scipy.io import wavfile

    spk = load_pkg("speaker_encoder.model")
    sr = 22050
    body = speechlike(int(2.2 * sr), 9, sr=sr)
    y = np.concatenate([np.zeros(sr // 2, np.float32), body, np.zeros(sr // 3, np.float32)])
    path = str(tmp_path / "ref.wav")
    wavfile.write(path, sr, np.stack([(y * 30000).astype(np.int16), (y * 20000).astype(np.int16)], axis=1))
    sd = synthetic.speaker_encoder_state_dict(seed=5)
    enc = spk.SpeakerEncoder(state_dict=sd, verbose=False)
    got = enc.embed_utterance_from_file(path, numpy=True)
    # the oracle's chain on the same file
    rate, data = wavfile.read(path)
    x = (data.astype(np.float32) / 32768.0).mean(axis=1)
    x = o_audio.trim_silence(o_audio.resample(x, rate, 16000), top_db=20)
    starts = spk.SpeakerEncoder.partial_table(len(x), 1.3, 0.75)
    need = int(starts[-1] + 160) * 160
    if need >= len(x):
        x = np.concatenate([x, np.zeros(need - len(x), dtype=x.dtype)])
    mel = o_audio.wav_to_mel_spectrogram(x)
    frames = starts[:, None] + np.arange(160)[None, :]
    partial = o_spk.forward({k: v.float() for k, v in sd.items()}, torch.from_numpy(np.ascontiguousarray(mel[frames])))
    mean = partial.numpy().mean(axis=0)
    ref = mean / np.linalg.norm(mean, 2)
    d = float(np.abs(got - ref).max())
    print(f"[speaker embedding from a file] max |d| {d:.2e} (|e| = 1)")
    assert got.shape == (256,) and d <= 1e-4
    enc.close()


def test_front_end_device_can_be_pinned():
    """`mv_audio_*` used to run on the calling thread's current HIP device -- 0 for every new request thread, on every rank.
    `mv_audio_set_device` pins the front end process-wide (SpeakerEncoder does it with its own device); a device that does not exist is an
    error, -1 gives the old behaviour back, and a call from a fresh thread gives the same spectrogram as one from the main thread."""
    import threading

    N = load_pkg("_native")
    with pytest.raises(ValueError):
        N.check(N.lib().mv_audio_set_device(torch.cuda.device_count()))
    N.check(N.lib().mv_audio_set_device(0))
    wav = np.random.default_rng(0).standard_normal(16000).astype(np.float32) * 0.1
    want = audio.wav_to_mel_spectrogram(wav)
    got = {}
    t = threading.Thread(target=lambda: got.setdefault("mel", audio.wav_to_mel_spectrogram(wav)))
    t.start(); t.join()
    assert np.array_equal(got["mel"], want)
    N.check(N.lib().mv_audio_set_device(-1))
