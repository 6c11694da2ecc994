"""Multi-band diffusion vocoder on the GPU (`mv_mbd_*`, csrc/mbd.hip) against the oracle (oracle/mbd.py: audiocraft 1.2.0's
DiffusionUnet / NoiseSchedule.generate_subsampled / MultiBandProcessor / re_eq restated in torch fp32; parity unpinned, see its
header), on synthetic packages with audiocraft's layout and IDENTICAL injected noise tensors.

Tolerances (fp32 reference, `autocast(dtype=float32)` at fam/llm/decoders.py:84): the GEMMs run on bf16 MFMA with fp32
accumulation; `precision=3` splits both operands hi + lo (~16 mantissa bits): SNR >= 60 dB on a single UNet evaluation and
>= 45 dB on a whole sampled waveform; `precision=1` (plain bf16 operands) >= 30 dB on one evaluation."""
import math

import pytest
import torch

from conftest import load_pkg
from oracle import mbd as o_mbd

pytestmark = pytest.mark.gpu

synthetic = load_pkg("synthetic")
mbd = load_pkg("mbd")


def snr_db(got, ref):
    err = float(((got - ref) ** 2).mean())
    return 10.0 * math.log10(float((ref ** 2).mean()) / max(err, 1e-30))


def oracle_bands(pkg, **ucfg):
    out = []
    for b in range(pkg["n_bands"]):
        u = pkg[b]["cfg"]["diffusion_unet"]
        cfg = o_mbd.UnetConfig(hidden=u["hidden"], depth=u["depth"], growth=u["growth"], kernel=u["kernel"], stride=u["stride"],
                               norm_groups=u["norm_groups"], res_blocks=u["res_blocks"], codec_dim=u["codec_dim"],
                               emb_all_layers=u["emb_all_layers"], bilstm=u["bilstm"], activation=u["activation"])
        out.append(o_mbd.BandModel(sd=pkg[b]["model_state"], cfg=cfg, processor=pkg[b]["processor_state"]))
    return out


@pytest.mark.parametrize("kw,T,frames", [
    (dict(hidden=16, depth=3, growth=2.0, codec_dim=32), 2048, 7),
    (dict(hidden=48, depth=2, growth=4.0, codec_dim=None), 4099, 0),          # 48 -> 192 channels: partial GEMM tiles, ragged length
    (dict(hidden=8, depth=4, growth=2.0, codec_dim=16, emb_all_layers=False, activation="gelu"), 5000, 16),
    (dict(hidden=16, depth=2, growth=2.0, codec_dim=16, res_blocks=2), 1600, 5),  # second ResBlock: dilation 2
    (dict(hidden=64, depth=2, growth=4.0, codec_dim=None), 205003, 0),         # >= 400 GEMM tiles with K % 64 == 0: the direct-to-LDS GEMM
    (dict(hidden=32, depth=3, growth=2.0, codec_dim=16, norm_groups=8), 6001, 7),   # 8 groups (4 / 8 / 16 channels wide): the GM = 8 instances of every kernel that sums
    (dict(hidden=48, depth=2, growth=4.0, codec_dim=None, norm_groups=2), 30001, 0),  # 2 groups of 24 / 96 channels on the thin kernel and the 64 x 192 tiles
])
def test_unet_forward_matches_oracle(kw, T, frames):
    pkg = synthetic.mbd_package(seed=3, n_bands=1, proc_bands=1, **kw)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, generator=g)
    lat = torch.randn(1, kw["codec_dim"], frames, generator=g) if kw.get("codec_dim") else None
    band = oracle_bands(pkg)[0]
    ref = o_mbd.unet_forward(band.sd, band.cfg, x.view(1, 1, -1), 731, lat).reshape(-1)
    for precision, floor in ((3, 60.0), (1, 30.0)):
        eng = mbd.MultiBandDiffusionBands.from_package(pkg, precision=precision, eq_bands=1)
        got = eng.unet_forward(0, x, 731, lat)
        s = snr_db(got, ref)
        print(f"[mbd unet {kw} T={T}] precision {precision}: SNR {s:.1f} dB (|ref| rms {float(ref.pow(2).mean().sqrt()):.3f})")
        assert s >= floor
        eng.close()


@pytest.mark.parametrize("kw,T", [
    (dict(hidden=64, depth=2, growth=4.0, codec_dim=None), 20011),    # 256 channels: one column tile, ragged last row tile
    (dict(hidden=96, depth=2, growth=4.0, codec_dim=None), 9000),     # 384 channels: a ragged second column tile (rows clamped)
    (dict(hidden=32, depth=3, growth=4.0, codec_dim=24), 16000),      # 512 channels at the bottom: two full column tiles, K = 1536 / 1024
])
def test_unet_forward_on_the_256x256_tile_gemm(kw, T, monkeypatch):
    """`mbd_gemm_t256_kernel` (256 x 256 tiles, double-buffered direct-to-LDS stages, three-term products): the kernel the 768- and
    3072-channel layers of the bench run on.  Its block-count threshold is lowered so that small shapes with ragged row / column
    tiles take it; the result must match the oracle like every other GEMM path -- and the 128 x 128 path on the same input."""
    pkg = synthetic.mbd_package(seed=11, n_bands=2, proc_bands=1, **kw)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, generator=g)
    lat = torch.randn(1, kw["codec_dim"], 11, generator=g) if kw.get("codec_dim") else None
    band = oracle_bands(pkg)[1]
    ref = o_mbd.unet_forward(band.sd, band.cfg, x.view(1, 1, -1), 412, lat).reshape(-1)
    eng = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, eq_bands=1)
    small = eng.unet_forward(1, x, 412, lat)               # default thresholds: 128 x 128 / 64 x 128 tiles at these sizes
    monkeypatch.setenv("MV_MBD_T256_MIN_BLOCKS", "1")
    big = eng.unet_forward(1, x, 412, lat)
    monkeypatch.delenv("MV_MBD_T256_MIN_BLOCKS")
    s_big, s_small, s_pair = snr_db(big, ref), snr_db(small, ref), snr_db(big, small)
    print(f"[mbd t256 {kw} T={T}] SNR vs oracle: 256x256 tiles {s_big:.1f} dB, 128x128 tiles {s_small:.1f} dB; between the two {s_pair:.1f} dB")
    assert s_big >= 60.0 and s_small >= 60.0 and s_pair >= 80.0
    assert not torch.equal(big, small) or s_pair == float("inf")
    eng.close()


@pytest.mark.parametrize("kw,T", [
    (dict(hidden=64, depth=2, growth=4.0, codec_dim=None), 20011),    # M = 5003 / 5004 rows: 21 tiles of 240 (the last one ragged) against 20 of 256
    (dict(hidden=32, depth=3, growth=4.0, codec_dim=24), 15360),      # M = 960 = 4 x 240 exactly, two column tiles; M = 961 for the up-convolution
])
def test_240_row_tiles_of_the_256x256_gemm(kw, T, monkeypatch):
    """Round 6: `mbd_gemm_t256_kernel` with 240-row tiles (the lower waves skip their eighth 16-row block): what launch_gemm picks when it
    saves a sixteenth of a one-round launch (M = 1200 / 4800 rows of the bench).  Forced here (`MV_MBD_T256_ROWS`) on shapes with ragged and
    exact tile counts, against the oracle and against 256-row tiles -- the same K order per output element, so only the GroupNorm statistics
    (summed per tile) may differ in their last bits."""
    pkg = synthetic.mbd_package(seed=17, n_bands=2, proc_bands=1, **kw)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, generator=g)
    lat = torch.randn(1, kw["codec_dim"], 11, generator=g) if kw.get("codec_dim") else None
    band = oracle_bands(pkg)[1]
    ref = o_mbd.unet_forward(band.sd, band.cfg, x.view(1, 1, -1), 333, lat).reshape(-1)
    eng = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, eq_bands=1)
    monkeypatch.setenv("MV_MBD_T256_MIN_BLOCKS", "1")
    monkeypatch.setenv("MV_MBD_T256_ROWS", "240")
    t240 = eng.unet_forward(1, x, 333, lat)
    again = eng.unet_forward(1, x, 333, lat)
    monkeypatch.setenv("MV_MBD_T256_ROWS", "256")
    t256 = eng.unet_forward(1, x, 333, lat)
    s_240, s_256, s_pair = snr_db(t240, ref), snr_db(t256, ref), snr_db(t240, t256)
    print(f"[mbd t256 rows {kw} T={T}] SNR vs oracle: 240-row tiles {s_240:.1f} dB, 256-row tiles {s_256:.1f} dB; between the two {s_pair:.1f} dB")
    assert torch.equal(t240, again)
    assert s_240 >= 60.0 and s_256 >= 60.0 and s_pair >= 100.0
    eng.close()


@pytest.mark.parametrize("kw,T", [
    (dict(hidden=48, depth=2, growth=4.0, codec_dim=None), 40003),   # the bench's two outer levels: 48 channels (thin kernel) and 192 (64 x 192 tiles)
    (dict(hidden=16, depth=3, growth=2.0, codec_dim=32), 9001),      # 16 / 32 / 64 channels: every instance of the thin kernel, K = 48 / 96 (two K-steps and a tail)
])
def test_thin_and_192_column_gemm_tiles(kw, T, monkeypatch):
    """Round 4: `mbd_gemm_thin_kernel` (N <= 64: the whole W in registers, A fragments straight from global memory) and
    `mbd_gemm_glds3_kernel<64, 192>` (the 192-channel layers without padding columns) against the oracle and against the 128-column
    tiles they replace (MV_MBD_NO_THIN / MV_MBD_NO_N192)."""
    pkg = synthetic.mbd_package(seed=13, n_bands=2, proc_bands=1, **kw)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, generator=g)
    lat = torch.randn(1, kw["codec_dim"], 9, generator=g) if kw.get("codec_dim") else None
    band = oracle_bands(pkg)[1]
    ref = o_mbd.unet_forward(band.sd, band.cfg, x.view(1, 1, -1), 250, lat).reshape(-1)
    eng = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, eq_bands=1)
    monkeypatch.setenv("MV_MBD_N192_MIN_BLOCKS", "1")
    new = eng.unet_forward(1, x, 250, lat)
    # ten 16-row tiles per wave instead of four (what launch_gemm picks at the bench's 76 800 rows): the same tiles, the same bits -- only the
    # GroupNorm records are cut differently
    monkeypatch.setenv("MV_MBD_THIN_MT", "10")
    mt10 = eng.unet_forward(1, x, 250, lat)
    monkeypatch.delenv("MV_MBD_THIN_MT")
    assert snr_db(mt10, new) >= 100.0
    monkeypatch.setenv("MV_MBD_NO_THIN", "1")
    monkeypatch.setenv("MV_MBD_NO_N192", "1")
    old = eng.unet_forward(1, x, 250, lat)
    s_new, s_old, s_pair = snr_db(new, ref), snr_db(old, ref), snr_db(new, old)
    print(f"[mbd thin / 64x192 {kw} T={T}] SNR vs oracle: new tiles {s_new:.1f} dB, 128-column tiles {s_old:.1f} dB; between the two {s_pair:.1f} dB")
    assert s_new >= 60.0 and s_old >= 60.0 and s_pair >= 80.0
    assert not torch.equal(new, old) or s_pair == float("inf")   # (identical would mean the switches did nothing)
    eng.close()


@pytest.mark.parametrize("kw,T,frames,t256", [
    (dict(hidden=16, depth=3, growth=2.0, codec_dim=32), 7001, 9, False),     # ragged lengths: the up-convolution's overhang past skip.T is dropped
    (dict(hidden=64, depth=2, growth=4.0, codec_dim=None), 20011, 0, True),   # the 256 x 256-tile kernel writes z + skip (groups repeat with period Cout)
    (dict(hidden=48, depth=2, growth=4.0, codec_dim=None), 40003, 0, False),  # 64 x 192 tiles
    (dict(hidden=16, depth=2, growth=2.0, codec_dim=24, bilstm=True), 3000, 6, False),   # behind a BiLSTM: the first level's sum comes from the statistics pass
])
def test_skip_addition_in_the_up_convolution_epilogue(kw, T, frames, t256, monkeypatch):
    """Round 6: the decoder's `z[:, :, :skip.T] + skip` is written by the epilogue of the transposed convolution that produces z (the s output
    phases of a step are one GEMM row: `MbdGemmArgs::valid_lo / valid_hi` drop the `pad` leading steps and the overhang, `residual` is the
    skip) together with the GroupNorm partials of the sum; the innermost skip is read in place when there is no BiLSTM.  With
    MV_MBD_NO_ADD_FUSE the sum is a pass of its own (`add_kernel`, then the two-pass statistics): the same values, statistics summed in another order."""
    pkg = synthetic.mbd_package(seed=21, n_bands=2, proc_bands=1, **kw)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, generator=g)
    lat = torch.randn(1, kw["codec_dim"], frames, generator=g) if kw.get("codec_dim") else None
    band = oracle_bands(pkg)[0]
    ref = o_mbd.unet_forward(band.sd, band.cfg, x.view(1, 1, -1), 600, lat).reshape(-1)
    eng = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, eq_bands=1)
    if t256:
        monkeypatch.setenv("MV_MBD_T256_MIN_BLOCKS", "1")
    monkeypatch.setenv("MV_MBD_N192_MIN_BLOCKS", "1")
    n0 = eng.fused_skip_additions()
    fused = eng.unet_forward(0, x, 600, lat)
    n1 = eng.fused_skip_additions()
    again = eng.unet_forward(0, x, 600, lat)
    monkeypatch.setenv("MV_MBD_NO_ADD_FUSE", "1")
    n2 = eng.fused_skip_additions()
    separate = eng.unet_forward(0, x, 600, lat)
    s_f, s_s, s_pair = snr_db(fused, ref), snr_db(separate, ref), snr_db(fused, separate)
    print(f"[mbd skip add {kw} T={T}] SNR vs oracle: in the epilogue {s_f:.1f} dB, separate pass {s_s:.1f} dB; between the two {s_pair:.1f} dB "
          f"(> 280 = identical bits: the statistics are rounded to fp32 from double sums)")
    assert n1 - n0 == kw["depth"] - 1 and eng.fused_skip_additions() == n2   # every level but the outermost; none with the switch
    assert torch.equal(fused, again)
    assert s_f >= 60.0 and s_s >= 60.0 and s_pair >= 100.0
    eng.close()


def test_fir_bank_on_the_fp32_matrix_cores_equals_the_valu_kernel(monkeypatch):
    """`fir_bank_mfma_kernel` (the julius low-pass bank as a Toeplitz product on v_mfma_f32_32x32x2_f32: exact fp32 products, fp32
    accumulation in tap order) against `fir_bank_kernel` (fmaf chain in tap order): the whole `tokens_to_wav` -- processor band split,
    band statistics, re_eq -- must agree to fp32 rounding."""
    kw = dict(hidden=16, depth=3, growth=2.0, codec_dim=32)
    pkg = synthetic.mbd_package(seed=7, n_bands=2, proc_bands=4, **kw)
    step_list = [999, 666, 333, 0]
    T, frames = 6400, 20
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(1, 32, frames, generator=g)
    wav_ref = 0.3 * torch.randn(1, 1, T, generator=g).cumsum(-1) / 20.0
    noise = torch.randn(2, len(step_list), T, generator=g)
    eng = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, step_list=step_list, eq_bands=8)
    a = eng.tokens_to_wav(lat, wav_ref, noise=noise).reshape(-1)
    monkeypatch.setenv("MV_MBD_FIR_VALU", "1")
    b = eng.tokens_to_wav(lat, wav_ref, noise=noise).reshape(-1)
    s = snr_db(a, b)
    print(f"[mbd fir] matrix-core bank vs VALU bank: SNR {s:.1f} dB (inf = identical bits)")
    assert s >= 110.0
    eng.close()


@pytest.mark.parametrize("kw,T,t256", [
    (dict(hidden=64, depth=2, growth=4.0, codec_dim=None), 205003, False),   # 256 channels on the three-term 128 x 128 kernel (>= 400 tiles)
    (dict(hidden=96, depth=2, growth=4.0, codec_dim=None), 9000, True),      # 384 channels, 256 x 256 tiles: a group (96 channels) inside a tile,
                                                                             # a ragged second column tile
    (dict(hidden=32, depth=3, growth=4.0, codec_dim=24), 16000, True),       # 512 channels: a tile spans two groups of 128
])
def test_group_norm_statistics_from_the_gemm_epilogue(kw, T, t256, monkeypatch):
    """The GEMMs whose output is only read through a GroupNorm leave the norm's partial sums from their epilogue (`MbdGemmArgs::gn_part`)
    instead of `gn_partial_kernel` reading the output again.  Against the two-pass statistics (`MV_MBD_NO_GN_FUSE`) the result may differ
    in the last bits of the statistics only."""
    pkg = synthetic.mbd_package(seed=5, n_bands=2, proc_bands=1, **kw)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, generator=g)
    lat = torch.randn(1, kw["codec_dim"], 11, generator=g) if kw.get("codec_dim") else None
    band = oracle_bands(pkg)[0]
    ref = o_mbd.unet_forward(band.sd, band.cfg, x.view(1, 1, -1), 90, lat).reshape(-1)
    eng = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, eq_bands=1)
    if t256:
        monkeypatch.setenv("MV_MBD_T256_MIN_BLOCKS", "1")
    fused = eng.unet_forward(0, x, 90, lat)
    again = eng.unet_forward(0, x, 90, lat)
    monkeypatch.setenv("MV_MBD_NO_GN_FUSE", "1")
    two_pass = eng.unet_forward(0, x, 90, lat)
    s_f, s_t, s_pair = snr_db(fused, ref), snr_db(two_pass, ref), snr_db(fused, two_pass)
    print(f"[mbd gn fuse {kw} T={T}] SNR vs oracle: fused {s_f:.1f} dB, two-pass {s_t:.1f} dB; between the two {s_pair:.1f} dB")
    assert torch.equal(fused, again)          # fixed-order partials: the same bits on every run
    assert s_f >= 60.0 and s_t >= 60.0 and s_pair >= 100.0
    assert not torch.equal(fused, two_pass) or s_pair == float("inf")   # (identical would mean the switch did nothing)
    eng.close()


def test_unet_with_bilstm_bottleneck():
    """BiLSTM bottleneck (codec conditioning reaches the output only through it): the HIP persistent LSTM kernel in both directions
    against the oracle's explicit-gate restatement and against torch.nn.LSTM on the same weights."""
    kw = dict(hidden=16, depth=2, growth=2.0, codec_dim=24, bilstm=True)
    pkg = synthetic.mbd_package(seed=5, n_bands=1, proc_bands=1, **kw)
    g = torch.Generator().manual_seed(9)
    x, lat = torch.randn(3200, generator=g), torch.randn(1, 24, 10, generator=g)
    sd = {k: v.float() for k, v in pkg[0]["model_state"].items()}
    cfg = oracle_bands(pkg)[0].cfg
    ref = o_mbd.unet_forward(sd, cfg, x.view(1, 1, -1), 500, lat).reshape(-1)
    # the oracle's BLSTM against torch's own
    z = torch.randn(1, 32, 50, generator=g)
    lstm = torch.nn.LSTM(bidirectional=True, num_layers=2, hidden_size=32, input_size=32)
    lstm.load_state_dict({k[len("bilstm.lstm."):]: v for k, v in sd.items() if k.startswith("bilstm.lstm.")})
    with torch.no_grad():
        want = (lstm(z.permute(2, 0, 1))[0] @ sd["bilstm.linear.weight"].T + sd["bilstm.linear.bias"]).permute(1, 2, 0)
    assert float((o_mbd.blstm(sd, z) - want).abs().max()) < 1e-5
    eng = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, eq_bands=1)
    got = eng.unet_forward(0, x, 500, lat)
    s = snr_db(got, ref)
    print(f"[mbd unet + BiLSTM] SNR {s:.1f} dB")
    assert s >= 60.0
    # the conditioning matters in this configuration
    other = eng.unet_forward(0, x, 500, lat + 1.0)
    assert float((other - got).abs().max()) > 1e-4
    eng.close()


def test_tokens_to_wav_with_injected_noise_matches_oracle():
    """The whole vocoder call after the codec: 2 band models x 3 sampler evaluations, MultiBandProcessor (4 bands), sum, re_eq (8
    bands) against a reference waveform -- same Gaussian draws on both sides."""
    kw = dict(hidden=16, depth=3, growth=2.0, codec_dim=32)
    pkg = synthetic.mbd_package(seed=7, n_bands=2, proc_bands=4, **kw)
    step_list = [999, 666, 333, 0]
    T, frames = 6400, 20
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(1, 32, frames, generator=g)
    wav_ref = 0.3 * torch.randn(1, 1, T, generator=g).cumsum(-1) / 20.0  # a coloured reference signal
    noise = torch.randn(2, len(step_list), T, generator=g)
    bands = oracle_bands(pkg)
    outs = []
    for b, band in enumerate(bands):
        x = o_mbd.generate_subsampled(band.sd, band.cfg, band.schedule, noise[b, 0].view(1, 1, -1), lat,
                                      [noise[b, 1 + i].view(1, 1, -1) for i in range(len(step_list) - 1)], step_list)
        outs.append(o_mbd.processor_return_sample(x, band.processor))
    ref = o_mbd.re_eq(outs[0] + outs[1], wav_ref, n_bands=8).reshape(-1)
    eng = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, step_list=step_list, eq_bands=8)
    got = eng.tokens_to_wav(lat, wav_ref, noise=noise).reshape(-1)
    s = snr_db(got, ref)
    mse = float(((got - ref) ** 2).mean())
    print(f"[mbd tokens_to_wav] SNR {s:.1f} dB, MSE {mse:.3e} (rms {float(ref.pow(2).mean().sqrt()):.3f})")
    assert s >= 45.0
    # without injected noise: the engine's own Philox stream; two calls differ (fresh draws), both finite and band-limited like the reference
    a, b = eng.tokens_to_wav(lat, wav_ref), eng.tokens_to_wav(lat, wav_ref)
    assert torch.isfinite(a).all() and not torch.equal(a, b)
    eng.close()


def test_device_hand_off_from_the_codec_equals_the_host_path():
    """`mv_codec_decode_device` -> `mv_mbd_generate_device` (the codec's waveform and latent read in place from device memory, what
    `vocoder.MultiBandDiffusion.tokens_to_wav` does by default) against the host-buffer path: same kernels, same Philox stream
    (same seed, same call count) => identical samples."""
    codec_mod, vocoder = load_pkg("codec"), load_pkg("vocoder")
    codec = codec_mod.EncodecDecoder.from_state_dict(synthetic.encodec_state_dict(seed=3))
    kw = dict(hidden=16, depth=3, growth=2.0, codec_dim=128)
    pkg = synthetic.mbd_package(seed=9, n_bands=2, proc_bands=4, **kw)
    tokens = torch.randint(0, 1024, (1, 8, 40), generator=torch.Generator().manual_seed(5))
    dev = vocoder.MultiBandDiffusion(codec, mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, seed=77, step_list=[999, 500, 0], eq_bands=8))
    a = dev.tokens_to_wav(tokens)
    host = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, seed=77, step_list=[999, 500, 0], eq_bands=8)
    wav_encodec, latent = codec.decode(tokens, return_latent=True)
    b = host.tokens_to_wav(latent, wav_encodec)
    assert a.shape == b.shape == (1, 1, 40 * 320) and torch.isfinite(a).all()
    assert torch.equal(a, b), f"max |d| {float((a - b).abs().max()):.3e}"
    # the workspace is kept between calls: a second utterance of another length on the same engines
    tokens2 = torch.randint(0, 1024, (1, 8, 64), generator=torch.Generator().manual_seed(6))
    a2 = dev.tokens_to_wav(tokens2)
    w2, l2 = codec.decode(tokens2, return_latent=True)
    assert torch.equal(a2, host.tokens_to_wav(l2, w2))
    host.close()
    dev.close()


def test_concurrent_lanes_give_the_samples_of_one_call_per_utterance():
    """`tokens_to_wav_many`: the utterances of a batch on several handles at once (`mv_mbd_clone`: shared weights, own stream and workspace,
    one host thread each).  Every utterance gets the seed of its place in the call order, so the waveforms are the SAME samples as with
    one `tokens_to_wav` per utterance -- utterances of different lengths, more utterances than lanes."""
    kw = dict(hidden=16, depth=3, growth=2.0, codec_dim=32)
    pkg = synthetic.mbd_package(seed=31, n_bands=2, proc_bands=4, **kw)
    g = torch.Generator().manual_seed(4)
    items = []
    for frames in (20, 33, 25, 40, 20):
        T = frames * 320
        items.append((torch.randn(1, 32, frames, generator=g), 0.3 * torch.randn(1, 1, T, generator=g).cumsum(-1) / 20.0))
    one = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, seed=99, step_list=[999, 666, 333, 0], eq_bands=8)
    seq = [one.tokens_to_wav(lat, wav) for lat, wav in items]
    many = mbd.MultiBandDiffusionBands.from_package(pkg, precision=3, seed=99, step_list=[999, 666, 333, 0], eq_bands=8)
    par = many.tokens_to_wav_many(items, lanes=3)
    assert len(par) == len(seq) and len(many._lanes) == 2
    for a, b in zip(seq, par):
        assert a.shape == b.shape and torch.isfinite(b).all()
        assert torch.equal(a, b), f"max |d| {float((a - b).abs().max()):.3e}"
    # and again on the same lanes (workspaces kept, call counter advanced): still the one-at-a-time samples
    seq2 = [one.tokens_to_wav(lat, wav) for lat, wav in items[:3]]
    par2 = many.tokens_to_wav_many(items[:3], lanes=3)
    assert all(torch.equal(a, b) for a, b in zip(seq2, par2)) and not torch.equal(seq2[0], seq[0])
    one.close()
    many.close()
