"""End-to-end `TTS.synthesise` (fam/llm/fast_inference.py:111-195) on synthetic checkpoints with the reference's file layout:
text -> normalise -> BPE -> first stage (HIP) -> adapter -> second stage (HIP) -> adapter -> vocoder (HIP EnCodec decode) -> .wav,
with NOTHING injected: `TTS(model_dir)` builds its own vocoder and speaker encoder from the model directory."""
import os
import struct

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import codec as o_codec
from oracle import loudness as o_loud

pytestmark = pytest.mark.gpu

synthetic = load_pkg("synthetic")
fast_inference = load_pkg("fast_inference")
loudness = load_pkg("loudness")
codec = load_pkg("codec")


def read_pcm16(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE"
    n = struct.unpack("<I", raw[40:44])[0]
    return np.frombuffer(raw[44:44 + n], dtype=np.int16), struct.unpack("<I", raw[24:28])[0]


@pytest.fixture(scope="module")
def tts(tmp_path_factory):
    root = tmp_path_factory.mktemp("tts")
    model_dir = synthetic.write_model_dir(str(root / "model"), seed=3)
    t = fast_inference.TTS(model_dir, seed=1, output_dir=str(root / "out"), max_utterances=2)
    return t, model_dir


def test_tts_synthesise_end_to_end_with_the_default_vocoder(tts):
    t, model_dir = tts
    assert t.smodel is not None  # speaker_encoder.pt was found and loaded into the HIP speaker encoder
    dec = t.llm_second_stage.decoder
    seen = {}
    inner = dec.vocoder

    def spy(grid):
        seen["tokens"] = grid.clone()
        return inner(grid)

    dec.vocoder = spy
    spk = synthetic.speaker_embedding(0)
    try:
        path = t.synthesise("Hello  “world” — this is a test\tof the engine.", spk, top_p=0.95, guidance_scale=3.0, temperature=1.0)
    finally:
        dec.vocoder = inner
    assert path.endswith(".wav") and os.path.getsize(path) > 44 + 2 * 9600
    grid = seen["tokens"]
    assert grid.shape[0] == 1 and grid.shape[1] == 8 and grid.shape[2] >= 30
    assert int(grid.min()) >= 0 and int(grid.max()) < 1024
    # the saved PCM == oracle EnCodec decode of the same token grid, loudness-normalised the same way, to one 16-bit step
    sd = {k: v.float() for k, v in codec.fold_weight_norm(torch.load(os.path.join(model_dir, "encodec_24khz.pt"))).items()}
    owav, _ = o_codec.decode(sd, grid[0])
    # expected PCM from the ORACLE's loudness (oracle/loudness.py: written independently of the product's, pinned to the BS.1770 /
    # EBU Tech 3341 analytic cases in tests/test_loudness_cpu.py) -- the product's post-processing is not compared with itself
    want = torch.from_numpy(o_loud.normalize_loudness(owav.reshape(1, -1).double().numpy(), 24000, loudness_headroom_db=14.0, loudness_compressor=True))
    want_pcm = (want.reshape(-1).clamp(-1, 1) * 32767.0).round().to(torch.int16).numpy()
    pcm, rate = read_pcm16(path)
    assert rate == 24000 and pcm.shape == want_pcm.shape == (grid.shape[2] * 320,)
    d = np.abs(pcm.astype(np.int32) - want_pcm.astype(np.int32))
    print(f"[tts] {grid.shape[2]} frames -> {pcm.size} samples; max |dPCM| vs oracle codec + loudness = {int(d.max())} LSB")
    assert int(d.max()) <= 2
    with pytest.raises(ValueError):
        t.synthesise("你好", spk)  # non Latin-1 text is rejected (fam/llm/utils.py:45-48)


def test_speaker_reference_from_a_wav_file(tts, tmp_path):
    """`spk_ref_path` as the reference uses it: an audio file >= 30 s, embedded by the speaker encoder (HIP LSTM)."""
    from scipy.io import wavfile

    t, _ = tts
    sr = 16000
    n = sr * 31
    tt = np.arange(n) / sr
    rng = np.random.default_rng(1)
    y = 0.3 * np.sin(2 * np.pi * 150 * tt) * (1 + 0.4 * np.sin(2 * np.pi * 2.5 * tt)) + 0.01 * rng.standard_normal(n)
    ref = str(tmp_path / "ref.wav")
    wavfile.write(ref, sr, (y * 32767).astype(np.int16))
    path = t.synthesise("A short sentence for a cloned voice.", ref)
    pcm, rate = read_pcm16(path)
    assert rate == 24000 and pcm.size >= 9600
    short = str(tmp_path / "short.wav")
    wavfile.write(short, sr, (y[: sr * 5] * 32767).astype(np.int16))
    with pytest.raises(Exception, match="too short"):
        t.synthesise("x", short)


def test_batch_and_long_form(tts):
    t, _ = tts
    spk = synthetic.speaker_embedding(1)
    texts = ["First utterance of the batch.", "The second one is a little longer than the first.", "Third."]
    blobs = t.synthesise_batch(texts, spk)
    assert len(blobs) == 3 and all(b[:4] == b"RIFF" and len(b) > 44 + 2 * 9600 for b in blobs)
    long_text = "This sentence is repeated to make a long text. " * 12
    path = t.synthesise_long(long_text, spk, max_chars=120)
    pcm, rate = read_pcm16(path)
    n_chunks = len(load_pkg("longform").split_text(long_text, 120))
    assert n_chunks >= 4 and rate == 24000 and pcm.size > n_chunks * 9600


def test_tts_without_codec_weights_fails_at_construction(tmp_path):
    model_dir = synthetic.write_model_dir(str(tmp_path / "model"), seed=5)
    os.unlink(os.path.join(model_dir, "encodec_24khz.pt"))
    with pytest.raises(FileNotFoundError, match="encodec_24khz.pt"):
        fast_inference.TTS(model_dir, output_dir=str(tmp_path / "out"))


def test_tts_with_multiband_diffusion_in_the_model_dir(tmp_path):
    """`mbd_comp_8.pt` next to the checkpoints: the default vocoder becomes EnCodec decode + multi-band diffusion + re_eq, as
    `fam/llm/decoders.py:13,84-85` has it.  `synthesise` runs end to end on the engine's own noise; with injected noise the
    vocoder's waveform is compared with the oracle chain (oracle/codec.py -> oracle/mbd.py)."""
    import math

    from oracle import mbd as o_mbd

    small = dict(n_bands=2, hidden=16, depth=3, growth=2.0, codec_dim=128, proc_bands=4)
    model_dir = synthetic.write_model_dir(str(tmp_path / "model"), seed=6, mbd=small)
    t = fast_inference.TTS(model_dir, seed=2, output_dir=str(tmp_path / "out"))
    assert t.vocoder.diffusion is not None
    path = t.synthesise("Multi band diffusion in the loop.", synthetic.speaker_embedding(2))
    pcm, rate = read_pcm16(path)
    assert rate == 24000 and pcm.size >= 9600 and int(np.abs(pcm).max()) > 100
    # parity of the vocoder call with injected noise
    g = torch.Generator().manual_seed(4)
    grid = torch.randint(0, 1024, (1, 8, 40), generator=g)
    T = 40 * 320
    noise = torch.randn(2, 21, T, generator=g)
    got = t.vocoder.tokens_to_wav(grid, noise=noise).reshape(-1)
    sd = {k: v.float() for k, v in codec.fold_weight_norm(torch.load(os.path.join(model_dir, "encodec_24khz.pt"))).items()}
    owav, olat = o_codec.decode(sd, grid[0])
    pkg = torch.load(os.path.join(model_dir, "mbd_comp_8.pt"), weights_only=False)
    bands = []
    for b in range(2):
        u = pkg[b]["cfg"]["diffusion_unet"]
        bands.append(o_mbd.BandModel(sd=pkg[b]["model_state"], processor=pkg[b]["processor_state"],
                                     cfg=o_mbd.UnetConfig(hidden=u["hidden"], depth=u["depth"], growth=u["growth"], codec_dim=u["codec_dim"])))
    ref = o_mbd.tokens_to_wav(bands, owav.view(1, 1, -1), olat.view(1, 128, -1), [noise[b, 0].view(1, 1, -1) for b in range(2)],
                              [[noise[b, 1 + i].view(1, 1, -1) for i in range(20)] for b in range(2)]).reshape(-1)
    err = float(((got - ref) ** 2).mean())
    snr = 10 * math.log10(float((ref ** 2).mean()) / max(err, 1e-30))
    print(f"[tts + mbd] vocoder waveform vs oracle chain: SNR {snr:.1f} dB, MSE {err:.3e}")
    assert snr >= 40.0
