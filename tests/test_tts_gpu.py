"""End-to-end `TTS.synthesise` (fam/llm/fast_inference.py:111-195) on synthetic checkpoints with the reference's file layout:
text -> normalise -> BPE -> first stage (HIP) -> adapter -> second stage (HIP) -> adapter -> vocoder hook -> .wav."""
import os

import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu

synthetic = load_pkg("synthetic")
fast_inference = load_pkg("fast_inference")


def test_tts_synthesise_end_to_end(tmp_path):
    model_dir = synthetic.write_model_dir(str(tmp_path / "model"), seed=3)
    seen = {}

    def fake_vocoder(tokens):  # (1, 8, f) -> (1, 1, 320 f): stands in for MBD until the HIP vocoder lands
        seen["tokens"] = tokens.clone()
        f = tokens.shape[-1]
        t = torch.arange(320 * f) / 24000.0
        return (0.1 * torch.sin(2 * torch.pi * 220.0 * t)).view(1, 1, -1)

    tts = fast_inference.TTS(model_dir, seed=1, output_dir=str(tmp_path / "out"), vocoder=fake_vocoder)
    spk = synthetic.speaker_embedding(0)
    path = tts.synthesise("Hello  “world” — this is a test\tof the engine.", spk, top_p=0.95, guidance_scale=3.0, temperature=1.0)
    assert path.endswith(".wav") and os.path.getsize(path) > 44 + 2 * 9600
    grid = seen["tokens"]
    assert grid.shape[0] == 1 and grid.shape[1] == 8 and grid.shape[2] >= 30
    assert int(grid.min()) >= 0 and int(grid.max()) < 1024
    with pytest.raises(ValueError):
        tts.synthesise("你好", spk)  # non Latin-1 text is rejected (fam/llm/utils.py:45-48)
