"""Parity at the sizes `bench.py` TIMES: until round 5 the vocoder, codec and second-stage kernels inside the headline's
timed region were compared with the oracle on smaller shapes only.  Here, on the bench's own configurations:

  (i)   the four band UNets of `synthetic.mbd_package()` (hidden 48, depth 4, growth 4: 48 -> 192 -> 768 -> 3072 channels, 166 M parameters
        per band) on 307 200 samples = 960 frames, ONE evaluation of all bands per launch exactly as `mv_mbd_generate` launches them
        (`mv_mbd_unet_forward_sum`), against oracle/mbd.py -- asserting by `mv_mbd_kernel_counts` that `mbd_gemm_t256_kernel`,
        `mbd_gemm_glds3_kernel<64,192>` and `mbd_gemm_thin_kernel` ran;
  (ii)  `tokens_to_wav` at those widths, 4 bands x 20 evaluations + processor + 32-band re_eq on a short utterance with injected noise;
  (iii) EnCodec decode of (960 frames, 8 codebooks) -- in tests/test_codec_gpu.py's own parametrisation;
  (iv)  second-stage logits at `synthetic.stage2_model_args(block_size=2048)` (6 layers x 384) on ~1 090 tokens.

Tolerances are the ones of the small-shape tests (tests/test_mbd_gpu.py, tests/test_stage2_gpu.py): three-term bf16 products >= 60 dB per
UNet evaluation, >= 45 dB on a sampled waveform; second-stage logits within 0.05 sigma of the bf16-rounding oracle.
The oracle for (i)-(ii) restates audiocraft 1.2.0 from its published module graph: parity unpinned (oracle/mbd.py header)."""
import math
import time

import pytest
import torch

from conftest import load_pkg
from oracle import mbd as o_mbd
from oracle import stage2 as o_s2

pytestmark = pytest.mark.gpu

synthetic = load_pkg("synthetic")
mbd = load_pkg("mbd")
mv = load_pkg()

BENCH_SAMPLES = 960 * 320   # bench.py: 1920 first-stage tokens = 960 frames = 12.8 s at 24 kHz


def snr_db(got, ref):
    err = float(((got - ref) ** 2).mean())
    return 10.0 * math.log10(float((ref ** 2).mean()) / max(err, 1e-30))


def oracle_bands(pkg):
    out = []
    for b in range(pkg["n_bands"]):
        u = pkg[b]["cfg"]["diffusion_unet"]
        cfg = o_mbd.UnetConfig(hidden=u["hidden"], depth=u["depth"], growth=u["growth"], kernel=u["kernel"], stride=u["stride"],
                               norm_groups=u["norm_groups"], res_blocks=u["res_blocks"], codec_dim=u["codec_dim"],
                               emb_all_layers=u["emb_all_layers"], bilstm=u["bilstm"], activation=u["activation"])
        out.append(o_mbd.BandModel(sd=pkg[b]["model_state"], cfg=cfg, processor=pkg[b]["processor_state"]))
    return out


@pytest.fixture(scope="module")
def bench_vocoder():
    """The package and engine `bench.py` builds (`synthetic.mbd_package(seed=rank)`, precision 3, the default 20-evaluation schedule)."""
    pkg = synthetic.mbd_package(seed=0)
    u = pkg[0]["cfg"]["diffusion_unet"]
    assert (u["hidden"], u["depth"], u["growth"], u["codec_dim"], pkg["n_bands"]) == (48, 4, 4.0, 128, 4)
    eng = mbd.MultiBandDiffusionBands.from_package(pkg, seed=1234, precision=3)
    yield pkg, eng
    eng.close()


def test_unet_evaluation_of_all_bands_at_the_benched_shape(bench_vocoder):
    """(i) 4 bands x one UNet evaluation on 307 200 samples, launched as the generator launches them."""
    pkg, eng = bench_vocoder
    T, frames = BENCH_SAMPLES, 960
    g = torch.Generator().manual_seed(17)
    x = torch.randn(4, T, generator=g)
    lat = torch.randn(1, 128, frames, generator=g)
    before = eng.kernel_counts()
    got = eng.unet_forward_sum(x, 449, lat)
    ran = {k: v - before[k] for k, v in eng.kernel_counts().items()}
    t0 = time.perf_counter()
    ref = torch.zeros(T)
    with torch.no_grad():
        for b, band in enumerate(oracle_bands(pkg)):
            ref += o_mbd.unet_forward(band.sd, band.cfg, x[b].view(1, 1, -1), 449, lat).reshape(-1)
    s = snr_db(got, ref)
    print(f"[benched mbd unet: 4 bands x 166 M parameters, T = {T}] SNR {s:.1f} dB (|ref| rms {float(ref.pow(2).mean().sqrt()):.3f}); "
          f"GEMM launches {ran}; oracle {time.perf_counter() - t0:.1f} s")
    # the kernels the bench's stage time is made of (profiles/: t256 88 ms, 64 x 192 tiles ~11 ms, thin 4.6 ms of 128 ms)
    assert ran["mbd_gemm_t256_kernel"] > 0 and ran["mbd_gemm_glds3_kernel<64,192>"] > 0 and ran["mbd_gemm_thin_kernel"] > 0
    # ... and nothing else: 768- and 3072-channel layers on 256 x 256 tiles, 192-channel layers on 64 x 192, 48-channel layers thin
    assert sum(ran.values()) == ran["mbd_gemm_t256_kernel"] + ran["mbd_gemm_glds3_kernel<64,192>"] + ran["mbd_gemm_thin_kernel"]
    assert s >= 60.0


def test_tokens_to_wav_at_the_benched_widths_with_injected_noise(bench_vocoder):
    """(ii) the whole vocoder call after the codec at the bench's widths: 4 bands x 20 evaluations, MultiBandProcessor (8 bands), sum,
    re_eq (32 bands) -- on 40 frames with the same Gaussian draws on both sides."""
    pkg, eng = bench_vocoder
    frames = 40
    T = frames * 320
    g = torch.Generator().manual_seed(23)
    lat = torch.randn(1, 128, frames, generator=g)
    wav_ref = 0.3 * torch.randn(1, 1, T, generator=g).cumsum(-1) / 20.0   # a coloured reference signal for re_eq
    noise = torch.randn(4, 21, T, generator=g)
    t0 = time.perf_counter()
    outs = []
    with torch.no_grad():
        for b, band in enumerate(oracle_bands(pkg)):
            xb = o_mbd.generate_subsampled(band.sd, band.cfg, band.schedule, noise[b, 0].view(1, 1, -1), lat,
                                           [noise[b, 1 + i].view(1, 1, -1) for i in range(20)])
            outs.append(o_mbd.processor_return_sample(xb, band.processor))
        ref = o_mbd.re_eq(outs[0] + outs[1] + outs[2] + outs[3], wav_ref, n_bands=32).reshape(-1)
    t_or = time.perf_counter() - t0
    got = eng.tokens_to_wav(lat, wav_ref, noise=noise).reshape(-1)
    s = snr_db(got, ref)
    mse = float(((got - ref) ** 2).mean())
    print(f"[benched mbd tokens_to_wav: 4 bands x 20 evaluations, T = {T}] SNR {s:.1f} dB, MSE {mse:.3e} (rms {float(ref.pow(2).mean().sqrt()):.3f}); oracle {t_or:.1f} s")
    assert torch.isfinite(got).all()
    assert s >= 45.0


def test_second_stage_logits_at_the_benched_size():
    """(iv) `bench.py`'s second stage: `synthetic.stage2_model_args(block_size=2048)` = 6 layers x 384, on 128 text ids + 960 frames + the
    end marker = 1 089 positions (packed like `inference.pack_second_stage_input`), against Stage2Oracle(rounding="bf16")."""
    inference = load_pkg("inference")
    args = synthetic.stage2_model_args(block_size=2048)
    sd = synthetic.stage2_state_dict(args, speaker_emb_dim=256, seed=100)
    model = mv.GPT(mv.GPTConfig(**args), speaker_emb_dim=256, max_batch=1)
    model.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    text_ids = torch.randint(1025, 1025 + 512, (128,), generator=g).tolist()
    cbs = [torch.randint(0, 1024, (960,), generator=g).tolist() for _ in range(2)]
    grid = inference.pack_second_stage_input(text_ids, cbs, args["block_size"])
    idx = torch.tensor([grid], dtype=torch.long)   # (1, 2, 2048): 1 089 live positions, padded with 1024 like the reference does
    n_live = 128 + 960 + 1
    spk = synthetic.speaker_embedding(2)
    got = torch.stack(model.logits(idx, spk))      # (6, 1, 2048, V)
    cfg = o_s2.Stage2Config(**{k: v for k, v in args.items() if k in o_s2.Stage2Config.__dataclass_fields__})
    with torch.no_grad():
        ref = torch.stack(o_s2.Stage2Oracle(sd, cfg, rounding="bf16").forward(idx, spk))
    scale = float(ref.std())
    d_all = float((got - ref).abs().max())
    d_live = float((got[:, :, :n_live] - ref[:, :, :n_live]).abs().max())
    agree = float((got[:, :, :n_live].argmax(-1) == ref[:, :, :n_live].argmax(-1)).float().mean())
    print(f"[benched stage 2: 6 x 384, {n_live} live of 2048 positions] max|dlogit| {d_all:.4f} (live positions {d_live:.4f}), logit std {scale:.3f}, "
          f"argmax agreement on the live positions {100 * agree:.2f} %")
    assert d_all <= 0.05 * scale
    assert agree >= 0.97
    model.close()
