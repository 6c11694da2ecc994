"""Pins `oracle/codec.py` (restatement of the published EnCodec decoder) to the independent implementation installed in this
image, transformers' EncodecModel, on seeded random weights.  No GPU."""
import pytest
import torch

from conftest import load_pkg
from oracle import codec as o_codec

transformers = pytest.importorskip("transformers")


def seeded_encodec(seed=0):
    from transformers import EncodecConfig, EncodecModel

    torch.manual_seed(seed)
    model = EncodecModel(EncodecConfig()).eval()
    with torch.no_grad():  # codebooks are zero-initialised buffers: give them values
        for layer in model.quantizer.layers:
            layer.codebook.embed.copy_(torch.randn_like(layer.codebook.embed) * 0.5)
    return model


@pytest.mark.parametrize("frames", [9, 40])
def test_oracle_matches_transformers_encodec(frames):
    model = seeded_encodec(1)
    fold = load_pkg("codec").fold_weight_norm  # host-side helper of the product package (pure torch, no GPU)
    sd = {k: v.float() for k, v in fold(model.state_dict()).items()}
    g = torch.Generator().manual_seed(frames)
    tokens = torch.randint(0, 1024, (8, frames), generator=g)
    with torch.no_grad():
        ref = model._decode_frame(tokens.unsqueeze(0))  # (1, 1, frames * 320): quantizer.decode + decoder
        wav, latent = o_codec.decode(sd, tokens)
        ref_latent = model.quantizer.decode(tokens.unsqueeze(1))
    assert wav.shape == ref.shape == (1, 1, frames * 320)
    assert float((latent - ref_latent).abs().max()) < 1e-5
    assert float((wav - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
