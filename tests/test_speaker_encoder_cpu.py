"""Speaker encoder, host side (no GPU): the oracle network against the golden minted from the reference's `SpeakerEncoder`
class, the product's partial-utterance slicing against the reference's, and the restated librosa front end (parity unpinned:
librosa is not in this image) against independent closed forms."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_pkg
from oracle import speaker_encoder as o_spk

synthetic = load_pkg("synthetic")
audio = load_pkg("speaker_encoder.audio")
from oracle import audio_frontend as o_audio  # noqa: E402  (round 5: the numpy front end is the checker; the product runs it on the GPU)


def test_oracle_forward_matches_the_reference_class():
    g = np.load(os.path.join(GOLDEN, "speaker_encoder.npz"))
    sd = synthetic.speaker_encoder_state_dict(seed=int(g["seed"]))
    emb = o_spk.forward(sd, torch.from_numpy(g["mels"]))
    assert emb.shape == (5, 256)
    assert float((emb - torch.from_numpy(g["emb"])).abs().max()) < 1e-6
    assert torch.allclose(emb.norm(dim=1), torch.ones(5), atol=1e-6)


def test_partial_slices_match_the_reference():
    import importlib
    import re

    # the slicing is a staticmethod with no device work; read it off the class without constructing an engine
    model = importlib.import_module("metavoice-src_amd.speaker_encoder.model")
    g = np.load(os.path.join(GOLDEN, "speaker_encoder.npz"))
    for n, want in json.loads(str(g["slices"])).items():
        wav_slices, mel_slices = model.SpeakerEncoder.compute_partial_slices(int(n), rate=1.3, min_coverage=0.75)
        assert [[s.start, s.stop] for s in mel_slices] == want
        assert all(w.start == m.start * 160 and w.stop == m.stop * 160 for w, m in zip(wav_slices, mel_slices))


def test_mel_front_end_closed_forms():
    sr, n_fft = 16000, 400
    fb = audio.mel_filterbank(sr, n_fft, 40)
    assert np.array_equal(fb, o_audio.mel_filterbank(sr, n_fft, 40))   # the product builds the bank on the host (configuration), the oracle restates it
    assert fb.shape == (40, 201) and float(fb.min()) >= 0
    # Slaney area normalisation: every triangle integrates to ~1 over Hz (bin spacing sr / n_fft)
    area = fb.sum(axis=1) * (sr / n_fft)
    assert np.allclose(area, 1.0, atol=0.12)
    # centre frequencies are increasing and the scale is linear below 1 kHz (200/3 Hz per mel)
    peaks = fb.argmax(axis=1)
    assert np.all(np.diff(peaks) > 0)
    assert abs(float(audio._hz_to_mel(1000.0)) - 15.0) < 1e-9 and abs(float(audio._mel_to_hz(15.0)) - 1000.0) < 1e-6
    # a pure tone lands in the band whose triangle covers it, frames = 1 + len // hop (centred STFT)
    t = np.arange(sr, dtype=np.float32) / sr
    mel = o_audio.wav_to_mel_spectrogram(np.sin(2 * np.pi * 1000.0 * t).astype(np.float32))
    assert mel.shape == (101, 40) and mel.dtype == np.float32
    band = int(mel[50].argmax())
    assert fb[band, int(round(1000.0 / (sr / n_fft)))] > 0
    # Parseval on one frame: total power through an all-pass bank equals the windowed frame energy (checks window + rfft usage)
    x = np.random.default_rng(0).standard_normal(n_fft).astype(np.float32)
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32)
    spec = np.abs(np.fft.rfft(x * w)) ** 2
    assert np.isclose(spec[0] + 2 * spec[1:-1].sum() + spec[-1], n_fft * np.sum((x * w) ** 2), rtol=1e-4)


def test_trim_and_wav_reader(tmp_path):
    from scipy.io import wavfile

    sr = 22050
    t = np.arange(sr) / sr
    tone = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    y = np.concatenate([np.zeros(sr, np.float32), tone, np.zeros(sr // 2, np.float32)])
    trimmed = o_audio.trim_silence(y, top_db=20)
    assert abs(len(trimmed) - sr) <= 2 * 2048  # the tone survives, the silence goes (frame granularity)
    path = str(tmp_path / "a.wav")
    wavfile.write(path, sr, (y * 32767).astype(np.int16))
    x = audio.load_audio(path, sr=sr)   # the reader (no rate change here: the resampler runs on the GPU, tests/test_audio_front_gpu.py)
    assert len(x) == len(y) and x.dtype == np.float32
    assert abs(float(np.abs(x).max()) - 0.5) < 0.02
    assert o_audio.trim_silence(np.zeros(5000, np.float32)).size == 0
    # the resampling filter the product designs on the host is scipy.signal.resample_poly's own (firwin, Kaiser 5.0), with its alignment
    from scipy.signal import firwin
    for up, down in ((160, 441), (1, 3), (2, 1)):
        h, pre = audio.resample_filter(up, down)
        half = 10 * max(up, down)
        n_pre_pad = down - half % down
        assert pre == (half + n_pre_pad) // down and not h[:n_pre_pad].any()
        assert float(np.abs(h[n_pre_pad:] - (firwin(2 * half + 1, 1.0 / max(up, down), window=("kaiser", 5.0)) * up).astype(np.float32)).max()) < 1e-7


def test_load_audio_falls_back_to_an_installed_decoder_or_says_so(tmp_path, monkeypatch):
    """RIFF/WAVE is sniffed by content (serving stores every upload as `.wav`, like the reference); anything else goes to the first
    decoder the process has -- none in this image: `UnsupportedAudioFormat`, which serving turns into 415 -- e.g. `soundfile`."""
    import sys
    import types

    import numpy as np
    from scipy.io import wavfile

    audio = load_pkg("speaker_encoder.audio")
    wav = tmp_path / "upload.mp3"   # a WAVE file under another name still loads
    wavfile.write(wav, 8000, (np.sin(np.arange(4000) / 7.0) * 12000).astype(np.int16))
    x = audio.load_audio(str(wav), sr=8000)
    assert x.dtype == np.float32 and x.shape == (4000,) and 0.3 < float(np.abs(x).max()) < 0.4
    mp3 = tmp_path / "voice.wav"    # ...and something that is not WAVE does not, whatever it is called
    mp3.write_bytes(b"ID3\x03\x00" + bytes(range(256)) * 8)
    for m in ("soundfile", "audioread"):
        monkeypatch.setitem(sys.modules, m, None)   # make sure the import fails even on a host that has them
    monkeypatch.setenv("PATH", str(tmp_path))       # ...and that no ffmpeg is found
    with pytest.raises(audio.UnsupportedAudioFormat) as ei:
        audio.load_audio(str(mp3))
    assert "convert it to .wav" in str(ei.value) and isinstance(ei.value, ValueError)
    fake = types.ModuleType("soundfile")
    fake.read = lambda path, dtype="float32", always_2d=False: (np.stack([np.full(32000, 0.25, np.float32), np.full(32000, 0.75, np.float32)], axis=1), 32000)
    monkeypatch.setitem(sys.modules, "soundfile", fake)
    y = audio.load_audio(str(mp3), sr=32000)
    assert y.shape == (32000,) and abs(float(y[8000]) - 0.5) < 1e-6   # stereo -> mono (32 kHz -> 16 kHz is the GPU resampler's: tests/test_audio_front_gpu.py)


def test_get_cached_file_local_and_uri(tmp_path, monkeypatch):
    """`fam/llm/inference.py:392-416`: local paths pass through (or FileNotFoundError); an http URI is fetched ONCE into the cache
    directory under `audio_<md5(uri)><ext>`."""
    import hashlib
    import threading
    from functools import partial
    from http.server import HTTPServer, SimpleHTTPRequestHandler

    inf = load_pkg("inference")
    monkeypatch.setenv("METAVOICE_CACHE_DIR", str(tmp_path / "cache"))
    local = tmp_path / "ref.wav"
    local.write_bytes(b"RIFFxxxxWAVE" + bytes(range(200)))
    assert inf.get_cached_file(str(local)) == str(local)
    with pytest.raises(FileNotFoundError):
        inf.get_cached_file(str(tmp_path / "nope.wav"))
    with pytest.raises(FileNotFoundError):
        inf.get_cached_file("httpfoo")  # starts with "http" (the reference's test), is no URI

    hits = []

    class H(SimpleHTTPRequestHandler):
        def log_message(self, *a):
            pass

        def do_GET(self):
            hits.append(self.path)
            super().do_GET()

    srv = HTTPServer(("127.0.0.1", 0), partial(H, directory=str(tmp_path)))
    t = threading.Thread(target=srv.serve_forever, daemon=True)
    t.start()
    try:
        uri = f"http://127.0.0.1:{srv.server_address[1]}/ref.wav"
        got = inf.get_cached_file(uri)
        assert os.path.basename(got) == "audio_" + hashlib.md5(uri.encode()).hexdigest() + ".wav"
        assert os.path.dirname(got) == str(tmp_path / "cache" / "metavoice")
        assert open(got, "rb").read() == local.read_bytes()
        assert inf.get_cached_file(uri) == got and len(hits) == 1  # second call: no fetch
        with pytest.raises(Exception):
            inf.get_cached_file(uri.replace("ref.wav", "missing.wav"))
        assert not [f for f in os.listdir(os.path.dirname(got)) if f.startswith(".fetch_")]  # no half files left behind
        monkeypatch.setattr(inf, "MAX_REFERENCE_BYTES", 64)
        with pytest.raises(ValueError):
            inf.get_cached_file(uri + "?again=1")
    finally:
        srv.shutdown()


def test_get_cached_embedding_is_computed_once_per_file(tmp_path, monkeypatch):
    """`fam/llm/inference.py:419-435` (+ recomputed when the audio file is newer than the entry)."""
    import hashlib

    inf = load_pkg("inference")
    monkeypatch.setenv("METAVOICE_CACHE_DIR", str(tmp_path / "cache"))

    class Enc:
        calls = 0

        def embed_utterance_from_file(self, path, numpy):
            Enc.calls += 1
            return torch.full((256,), float(Enc.calls))

    f = tmp_path / "a.wav"
    f.write_bytes(b"x")
    with pytest.raises(FileNotFoundError):
        inf.get_cached_embedding(str(tmp_path / "b.wav"), Enc())
    e1 = inf.get_cached_embedding(str(f), Enc())
    e2 = inf.get_cached_embedding(str(f), Enc())
    assert e1.shape == (1, 256) and torch.equal(e1, e2) and Enc.calls == 1
    entry = tmp_path / "cache" / "fam" / ("embedding_" + hashlib.md5(str(f).encode()).hexdigest() + ".pt")
    assert entry.exists()  # the reference's name for the same file
    os.utime(f, (os.path.getmtime(entry) + 10, os.path.getmtime(entry) + 10))
    e3 = inf.get_cached_embedding(str(f), Enc())
    assert Enc.calls == 2 and float(e3[0, 0]) == 2.0
    entry.write_bytes(b"garbage")
    os.utime(entry, (os.path.getmtime(f) + 10, os.path.getmtime(f) + 10))
    assert float(inf.get_cached_embedding(str(f), Enc())[0, 0]) == 3.0  # unreadable entry -> recomputed, not an error


def test_uploads_under_the_temporary_directory_leave_no_cache_entry(tmp_path, monkeypatch):
    """A served upload is a fresh temporary file per request: keyed by path (the reference's scheme) every request would leave an
    `embedding_*.pt` behind for ever.  Also: a local file whose name starts with "http" is a local file."""
    inf = load_pkg("inference")
    monkeypatch.setenv("METAVOICE_CACHE_DIR", str(tmp_path / "cache"))

    class Enc:
        calls = 0

        def embed_utterance_from_file(self, path, numpy):
            Enc.calls += 1
            return torch.full((256,), float(Enc.calls))

    f = tmp_path / (inf.UPLOAD_PREFIX + "k3j2.wav")   # the name serving.py stores an upload under (pytest's tmp_path is under the temp dir)
    f.write_bytes(b"x")
    e1 = inf.get_cached_embedding(str(f), Enc())
    e2 = inf.get_cached_embedding(str(f), Enc())
    assert e1.shape == (1, 256) and Enc.calls == 2 and float(e2[0, 0]) == 2.0
    assert not (tmp_path / "cache" / "fam").exists() or not list((tmp_path / "cache" / "fam").glob("embedding_*"))
    # ... but a user's own reference that happens to live under the temp directory is cached like any other
    mine = tmp_path / "my_voice.wav"
    mine.write_bytes(b"x")
    inf.get_cached_embedding(str(mine), Enc())
    inf.get_cached_embedding(str(mine), Enc())
    assert Enc.calls == 3 and len(list((tmp_path / "cache" / "fam").glob("embedding_*"))) == 1
    g = tmp_path / "http_take3.wav"
    g.write_bytes(b"x")
    assert inf.get_cached_file(str(g)) == str(g)
    with pytest.raises(FileNotFoundError):
        inf.get_cached_file("httpx://nowhere/a.wav")


def test_mel_front_end_matches_the_independent_implementation_in_the_image():
    """`fam/quantiser/audio/speaker_encoder/audio.py:15-21` is one librosa call.  librosa is not in the image; `transformers.audio_utils`
    (its numpy equivalent: `mel_filter_bank(norm="slaney", mel_scale="slaney")`, `spectrogram(center=True, pad_mode="constant",
    power=2.0)`, periodic Hann 400 / hop 160) is, and minted `tests/golden/mel_frontend.npz` (`make_frontend_golden.py`, committed).
    The restated front end (oracle/audio_frontend.py; the product's host-built filter bank is the same array) reproduces the filter bank
    exactly and the mel power spectrogram to float32 rounding."""
    import importlib.util

    g = np.load(os.path.join(GOLDEN, "mel_frontend.npz"))
    spec = importlib.util.spec_from_file_location("make_frontend_golden", os.path.join(GOLDEN, "make_frontend_golden.py"))
    # the waveform recipe is pure integer / closed-form arithmetic: re-create it without importing transformers
    src = open(spec.origin).read()
    ns = {"np": np}
    exec(compile(src[src.index("SR, N_FFT"):src.index("def main")], spec.origin, "exec"), ns)
    wav = ns["test_waveform"](int(g["seed"]), float(g["seconds"]))
    fb = audio.mel_filterbank(16000, 400, 40)
    assert fb.shape == g["filter_bank"].shape == (40, 201)
    assert float(np.abs(fb - g["filter_bank"]).max()) <= 1e-9
    mel = o_audio.wav_to_mel_spectrogram(wav)   # the oracle is what the golden pins; the GPU kernel is held to the oracle (tests/test_audio_front_gpu.py)
    assert mel.shape == g["mel"].shape == (151, 40)
    assert float(np.abs(mel - g["mel"]).max()) <= 2e-6 * float(np.abs(g["mel"]).max())
    rel = np.abs(mel - g["mel"]) / (np.abs(g["mel"]) + 1e-6)
    assert float(np.percentile(rel, 99)) < 1e-4
