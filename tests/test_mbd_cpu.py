"""Multi-band diffusion, host side (no GPU): self-consistency of the oracle's restated julius / audiocraft pieces and agreement of
the product's host-side configuration math (filter banks, sampler scalars) with the oracle's independent restatement."""
import math

import numpy as np
import torch

from conftest import load_pkg
from oracle import mbd as o_mbd

mbd = load_pkg("mbd")
synthetic = load_pkg("synthetic")


def test_split_bands_is_a_partition_of_the_signal():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 1, 6000, generator=g)
    bands = o_mbd.split_bands(x, 8, 24000)
    assert bands.shape == (8, 1, 1, 6000)
    assert float((bands.sum(0) - x).abs().max()) < 1e-5
    bank = o_mbd.lowpass_bank([c / 24000 for c in o_mbd.mel_cutoffs(8, 24000)])
    assert np.allclose(bank.sum(1).numpy(), 1.0, atol=1e-6) and bank.shape[1] % 2 == 1
    # a low tone stays in the lowest band, a high one in the highest
    t = torch.arange(6000) / 24000.0
    lo = o_mbd.split_bands(torch.sin(2 * math.pi * 100 * t).view(1, 1, -1), 8, 24000)[:, 0, 0, 1000:5000].pow(2).mean(1)
    hi = o_mbd.split_bands(torch.sin(2 * math.pi * 11000 * t).view(1, 1, -1), 8, 24000)[:, 0, 0, 1000:5000].pow(2).mean(1)
    assert int(lo.argmax()) == 0 and int(hi.argmax()) == 7


def test_product_filterbank_and_schedule_match_the_oracle():
    for nb in (4, 8, 32):
        a = mbd.lowpass_bank(nb, 24000)
        b = o_mbd.lowpass_bank([c / 24000 for c in o_mbd.mel_cutoffs(nb, 24000)]).numpy()
        assert a.shape == b.shape and float(np.abs(a - b).max()) < 1e-7
    steps, est, isa, sig = mbd.subsampled_schedule({}, o_mbd.DEFAULT_STEP_LIST)
    ref = o_mbd.subsampled_coefficients(o_mbd.ScheduleConfig())
    assert steps == [c["step"] for c in ref] and len(steps) == 20 and steps[0] == 999 and steps[-1] == 49
    assert max(abs(a - c["est"]) for a, c in zip(est, ref)) < 1e-7 and max(abs(a - c["sigma"]) for a, c in zip(sig, ref)) < 1e-7
    assert sig[-1] == 0.0 and all(s > 0 for s in sig[:-1])  # no noise is added after the last evaluation


def test_re_eq_and_processor_identities():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 1, 5000, generator=g)
    assert float((o_mbd.re_eq(x, x, n_bands=8) - x).abs().max()) < 1e-4   # equalising a signal against itself changes nothing
    assert float((o_mbd.re_eq(3.0 * x, x, n_bands=8) - x).abs().max()) < 1e-4  # a gain is undone band by band
    proc = {"counts": torch.tensor([4.0]), "sum_x": torch.zeros(4), "sum_x2": 4 * torch.ones(4), "sum_target_x2": 4 * torch.ones(4)}
    assert float((o_mbd.processor_return_sample(x, proc) - x).abs().max()) < 1e-5  # std == target_std, mean 0: identity


def test_synthetic_package_has_audiocraft_layout_and_the_oracle_runs_it():
    pkg = synthetic.mbd_package(seed=0, n_bands=2, hidden=8, depth=2, growth=2.0, codec_dim=16, proc_bands=4)
    assert pkg["n_bands"] == 2 and set(pkg[0]) == {"model_state", "processor_state", "cfg"}
    sd = pkg[1]["model_state"]
    assert sd["encoders.1.conv.weight"].shape == (16, 8, 8) and sd["decoders.0.convtr.weight"].shape == (16, 8, 8)
    cfg = o_mbd.UnetConfig(hidden=8, depth=2, growth=2.0, codec_dim=16)
    y = o_mbd.unet_forward(sd, cfg, torch.randn(1, 1, 333), 10, torch.randn(1, 16, 3))
    assert y.shape == (1, 1, 333) and torch.isfinite(y).all()
