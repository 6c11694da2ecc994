"""HIP EnCodec decode (mv_codec_*, csrc/codec.hip) against the oracle (pinned to transformers' EncodecModel by
tests/test_codec_cpu.py) and against transformers itself when importable, on seeded random weights."""
import pytest
import torch

from conftest import load_pkg
from oracle import codec as o_codec
from test_codec_cpu import seeded_encodec

pytestmark = pytest.mark.gpu

codec = load_pkg("codec")


@pytest.mark.parametrize("frames,n_q", [(8, 8), (75, 8), (150, 2), (33, 5)])
def test_decode_matches_oracle_and_transformers(frames, n_q):
    model = seeded_encodec(2)
    sd_raw = model.state_dict()
    dec = codec.EncodecDecoder.from_state_dict(sd_raw, n_q=8)
    sd = {k: v.float() for k, v in codec.fold_weight_norm(sd_raw).items()}
    g = torch.Generator().manual_seed(frames + n_q)
    tokens = torch.randint(0, 1024, (n_q, frames), generator=g)
    wav, latent = dec.decode(tokens.unsqueeze(0), return_latent=True)
    with torch.no_grad():
        owav, olat = o_codec.decode(sd, tokens)
        twav = model._decode_frame(tokens.unsqueeze(0))
    assert wav.shape == (1, 1, frames * 320)
    scale = float(owav.abs().max())
    d_lat = float((latent - olat).abs().max())
    d_o = float((wav - owav).abs().max())
    d_t = float((wav - twav).abs().max())
    mse = float(((wav - owav) ** 2).mean())
    print(f"[codec frames={frames} n_q={n_q}] max|dwav| vs oracle {d_o:.2e}, vs transformers {d_t:.2e} (|wav|max {scale:.3f}), MSE {mse:.2e}, latent {d_lat:.1e}")
    # fp32 everywhere; differences are summation order only.  Tolerance: 1e-4 of full scale, MSE < 1e-9 * scale^2
    assert d_lat < 1e-5 and d_o < 1e-4 * max(scale, 1.0) and d_t < 1e-4 * max(scale, 1.0)
    assert mse < 1e-9 * max(scale, 1.0) ** 2
    dec.close()


def test_rejects_bad_tokens():
    model = seeded_encodec(3)
    dec = codec.EncodecDecoder.from_state_dict(model.state_dict())
    with pytest.raises(ValueError):
        dec.decode(torch.full((1, 8, 20), 1024))
    with pytest.raises(ValueError):
        dec.decode(torch.zeros(1, 8, 3, dtype=torch.long))  # too short for the reflect padding
    dec.close()
