"""Speaker encoder on the GPU (`mv_spkenc_*`, csrc/lstm.hip) against the oracle and the golden minted from the reference's own
`SpeakerEncoder.forward` (fam/quantiser/audio/speaker_encoder/model.py:54-57), plus `embed_utterance` end to end on a synthetic
waveform.  fp32 everywhere: the tolerance covers summation order and expf/tanhf implementations only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_pkg
from oracle import speaker_encoder as o_spk

pytestmark = pytest.mark.gpu

synthetic = load_pkg("synthetic")
spk = load_pkg("speaker_encoder")
audio = load_pkg("speaker_encoder.audio")


def test_forward_matches_reference_golden_and_oracle():
    g = np.load(os.path.join(GOLDEN, "speaker_encoder.npz"))
    sd = synthetic.speaker_encoder_state_dict(seed=int(g["seed"]))
    enc = spk.SpeakerEncoder(state_dict=sd, verbose=False)
    mels = torch.from_numpy(g["mels"])
    got = enc(mels)
    d_ref = float((got - torch.from_numpy(g["emb"])).abs().max())
    d_orc = float((got - o_spk.forward(sd, mels)).abs().max())
    print(f"[speaker encoder] 5 partials x 160 frames: max|d| vs reference golden {d_ref:.2e}, vs oracle {d_orc:.2e}")
    assert got.shape == (5, 256) and d_ref < 2e-5 and d_orc < 2e-5
    # a batch that needs more than one LDS tile of sequences, and an odd frame count
    gen = torch.Generator().manual_seed(1)
    big = torch.rand(70, 37, 40, generator=gen) * 2.0
    d_big = float((enc(big) - o_spk.forward(sd, big)).abs().max())
    assert d_big < 2e-5
    # repeated calls reuse the granule buffers with a new launch serial: identical bits
    assert torch.equal(enc(mels), got)
    enc.close()


def test_embed_utterance_matches_oracle_pipeline():
    sd = synthetic.speaker_encoder_state_dict(seed=7)
    enc = spk.SpeakerEncoder(state_dict=sd, verbose=False)
    rng = np.random.default_rng(0)
    t = np.arange(16000 * 4) / 16000.0
    wav = (0.3 * np.sin(2 * np.pi * 180 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * rng.standard_normal(t.size)).astype(np.float32)
    emb, partials, wav_slices = enc.embed_utterance(wav, return_partials=True)
    # the same pipeline with the oracle network on the same mel partials
    _, mel_slices = enc.compute_partial_slices(len(wav), 1.3, 0.75)
    padded = np.pad(wav, (0, max(0, wav_slices[-1].stop - len(wav))))
    mel = audio.wav_to_mel_spectrogram(padded)
    o = o_spk.forward(sd, torch.from_numpy(np.array([mel[s] for s in mel_slices])))
    raw = o.numpy().mean(axis=0)
    want = raw / np.linalg.norm(raw, 2)
    assert emb.shape == (256,) and abs(float(np.linalg.norm(emb)) - 1.0) < 1e-5
    assert float(np.abs(emb - want).max()) < 2e-5 and float((partials - o).abs().max()) < 2e-5
    enc.close()
