"""HIP EnCodec decode (mv_codec_*, csrc/codec.hip) against the oracle (pinned to transformers' EncodecModel by
tests/test_codec_cpu.py) and against transformers itself when importable, on seeded random weights."""
import pytest
import torch

from conftest import load_pkg
from oracle import codec as o_codec
from test_codec_cpu import seeded_encodec

pytestmark = pytest.mark.gpu

codec = load_pkg("codec")


@pytest.mark.parametrize("frames,n_q", [(8, 8), (75, 8), (150, 2), (33, 5), (960, 8)])   # (960, 8): the utterance bench.py times
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


@pytest.mark.parametrize("hook", ["MV_LSTM_FAULT", "MV_LSTM_FORCE_STEPWISE"])
def test_lstm_falls_back_to_one_launch_per_step(hook):
    """when the persistent LSTM kernels cannot run (their workgroups not co-resident: a wait times out and raises the abort
    word) the codec and the speaker encoder used to fail the request.  Now the call is redone with one launch per time step, and the
    process stays in that mode.  `MV_LSTM_FAULT` raises the abort word behind the first persistent launch as a timed-out wait would;
    `MV_LSTM_FORCE_STEPWISE` starts in the per-step mode.  Run in a child process: the mode is sticky."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    code = r'''
import sys, torch
sys.path.insert(0, "tests")
from conftest import load_pkg
from oracle import codec as o_codec
from oracle import speaker_encoder as o_spk
from test_codec_cpu import seeded_encodec
codec, synthetic = load_pkg("codec"), load_pkg("synthetic")
model = seeded_encodec(2)
dec = codec.EncodecDecoder.from_state_dict(model.state_dict(), n_q=8)
sd = {k: v.float() for k, v in codec.fold_weight_norm(model.state_dict()).items()}
tokens = torch.randint(0, 1024, (8, 40), generator=torch.Generator().manual_seed(1))
for rep in range(2):   # the first call takes the fault (or starts stepwise), the second one runs in the sticky mode
    wav = dec.decode(tokens.unsqueeze(0))
    with torch.no_grad():
        owav, _ = o_codec.decode(sd, tokens)
    d = float((wav - owav).abs().max())
    assert d < 1e-4 * max(float(owav.abs().max()), 1.0), d
spk = load_pkg("speaker_encoder.model")
ssd = synthetic.speaker_encoder_state_dict(seed=4)
enc = spk.SpeakerEncoder(state_dict=ssd, verbose=False)
mels = torch.rand(3, 160, 40, generator=torch.Generator().manual_seed(12)) * 3.0
d2 = float((enc(mels) - o_spk.forward(ssd, mels)).abs().max())
assert d2 < 1e-5, d2
print("FALLBACK_OK", d, d2)
'''
    env = dict(os.environ)
    env[hook] = "1"
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "FALLBACK_OK" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]
    if hook == "MV_LSTM_FAULT":
        assert "continue with one launch per time step" in p.stderr


def test_lstm_rows_in_registers_equal_rows_in_lds():
    """The persistent LSTM kernels keep their gate rows in registers at EnCodec's size (H = 512, `lstm_seq2_kernel<8>`) and the speaker
    encoder's (H = 256, `lstm_seq_kernel<4>`); with `MV_LSTM_ROWS_IN_LDS` the general kernels (rows in LDS, any H) run instead.  Same
    products in the same order: the decoded waveform and the speaker embedding must be the same bits.  Child processes: the switch is
    read once per process."""
    import hashlib
    import os
    import subprocess
    import sys

    from conftest import ROOT

    code = r'''
import sys, hashlib, torch
sys.path.insert(0, "tests")
from conftest import load_pkg
from test_codec_cpu import seeded_encodec
codec = load_pkg("codec")
dec = codec.EncodecDecoder.from_state_dict(seeded_encodec(2).state_dict(), n_q=8)
tokens = torch.randint(0, 1024, (1, 8, 75), generator=torch.Generator().manual_seed(3))
wav = dec.decode(tokens)
spk, synthetic = load_pkg("speaker_encoder.model"), load_pkg("synthetic")
enc = spk.SpeakerEncoder(state_dict=synthetic.speaker_encoder_state_dict(seed=4), verbose=False)   # H = 256: lstm_seq_kernel<4>
emb = enc(torch.rand(5, 160, 40, generator=torch.Generator().manual_seed(12)) * 3.0)
h = hashlib.sha1(wav.detach().float().cpu().contiguous().numpy().tobytes())
h.update(emb.detach().float().cpu().contiguous().numpy().tobytes())
print("DIGEST", h.hexdigest(), float(wav.abs().max()), float(emb.abs().max()))
'''
    digests = []
    for rows_in_lds in (False, True):
        env = dict(os.environ)
        env.pop("MV_LSTM_ROWS_IN_LDS", None)
        if rows_in_lds:
            env["MV_LSTM_ROWS_IN_LDS"] = "1"
        p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "DIGEST" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]
        digests.append(p.stdout.split("DIGEST")[1].split()[0])
    assert digests[0] == digests[1]
