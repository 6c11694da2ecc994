"""CPU oracle of the multi-band diffusion vocoder (TEST INFRASTRUCTURE -- only tests/, smoke() and bench.py's cpu leg may import this).

The reference reaches this arithmetic through a third-party dependency that is NOT under /root/reference and not installed here:
`fam/llm/decoders.py:13` builds `audiocraft.models.MultiBandDiffusion.get_mbd_24khz(bw=6)` (audiocraft==1.2.0, poetry.lock:131-132;
julius==0.2.7 for the band splitting) and `:84-85` calls `mbd.tokens_to_wav(tokens)`.  This file restates the published
algorithm of that call chain (Roman et al. 2023, "From Discrete Tokens to High-Fidelity Audio Using Multi-Band Diffusion";
module graph of audiocraft 1.2.0 `models/unet.py`, `modules/diffusion_schedule.py`, `models/multibanddiffusion.py`, julius
`lowpass.py` / `bands.py`), in plain torch fp32 on CPU:

    tokens_to_wav(tokens):  wav_encodec = codec.decode(tokens); cond = codec.decode_latent(tokens)        (oracle/codec.py)
                            wav = sum over the 4 bands of DiffusionProcess.generate(cond, initial_noise)
                            return re_eq(wav, ref=wav_encodec, n_bands=32)
    DiffusionProcess.generate = NoiseSchedule.generate_subsampled(model, initial, step_list = 999, 949, ..., 49, 0):
        20 evaluations of the band's DiffusionUnet, DDPM posterior mean + sigma * fresh Gaussian noise (none at the last
        step), clamp to +-clip, then MultiBandProcessor.return_sample (per-band affine rescale, 8 mel bands)
    DiffusionUnet: depth x [Conv1d(k, stride, pad (k - stride) / 2, no bias) -> GroupNorm -> act -> ResBlock(s)] with the step
        embedding added after every encoder layer, the codec latent (1x1 conv, nearest-resized) added at the bottleneck, then the
        mirrored decoder [ResBlock(s) -> GroupNorm -> act -> ConvTranspose1d] with skip connections added before each layer.
    re_eq: julius.SplitBands (windowed-sinc low-pass bank on mel-spaced cut-offs), per-band std matched to the reference signal.

PARITY UNPINNED: the reference holds no vector for this boundary and audiocraft / julius cannot be executed here; sizes of the
released checkpoint (`mbd_comp_8.pt`) are read from its own `cfg` entries at load time, the defaults below are the values of
audiocraft's `config/model/score/basic.yaml` as recalled (hidden 48, depth 4, growth 4, kernel 8, stride 4, 1 ResBlock, 4 norm
groups, step embeddings on all layers, codec_dim 128).  Stochastic: waveform parity needs the SAME noise tensors, so every
function here takes the Gaussian draws as arguments.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


# ---- julius 0.2.7: LowPassFilters / SplitBands -------------------------------------------------------------------------

def hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_to_hz(m):
    return 700.0 * (10.0 ** (m / 2595.0) - 1.0)


def mel_cutoffs(n_bands: int, sample_rate: int) -> List[float]:
    """julius.bands: `mel_frequencies(n_bands + 1, 0, sample_rate / 2)[1:-1]` -- n_bands - 1 cut-offs in Hz."""
    lo, hi = hz_to_mel(0.0), hz_to_mel(sample_rate / 2.0)
    mels = [lo + (hi - lo) * i / n_bands for i in range(n_bands + 1)]
    return [mel_to_hz(m) for m in mels][1:-1]


def lowpass_bank(cutoffs_norm: Sequence[float], zeros: float = 8.0) -> torch.Tensor:
    """julius.LowPassFilters: (n_filters, 2 * half + 1) Hann-windowed sinc filters for cut-offs given as a fraction of the
    sample rate, each normalised to unit DC gain; half = int(zeros / min(cutoff) / 2)."""
    half = int(zeros / min(c for c in cutoffs_norm if c > 0) / 2)
    window = torch.hann_window(2 * half + 1, periodic=False, dtype=torch.float64)
    t = torch.arange(-half, half + 1, dtype=torch.float64)
    out = []
    for c in cutoffs_norm:
        x = 2 * c * math.pi * t
        sinc = torch.where(x == 0, torch.ones_like(x), torch.sin(x) / torch.where(x == 0, torch.ones_like(x), x))
        f = 2 * c * window * sinc
        out.append(f / f.sum())
    return torch.stack(out).float()


def split_bands(x: torch.Tensor, n_bands: int, sample_rate: int) -> torch.Tensor:
    """julius.SplitBands(sample_rate, n_bands)(x): (n_bands, *x.shape); the bands add up to x.  Replicate padding, as
    LowPassFilters(pad=True) does."""
    if n_bands == 1:
        return x[None]
    bank = lowpass_bank([c / sample_rate for c in mel_cutoffs(n_bands, sample_rate)])
    half = (bank.shape[1] - 1) // 2
    flat = x.reshape(-1, 1, x.shape[-1]).float()
    padded = F.pad(flat, (half, half), mode="replicate")
    # direct convolution in fp64 accumulate-free form: conv1d fp32 (julius switches to FFT convolution above 32 taps half-width;
    # both evaluate the same sum)
    lows = F.conv1d(padded, bank[:, None])  # (rows, n_bands - 1, T)
    lows = lows.permute(1, 0, 2).reshape(n_bands - 1, *x.shape)
    bands = [lows[0]]
    for i in range(1, n_bands - 1):
        bands.append(lows[i] - lows[i - 1])
    bands.append(x - lows[-1])
    return torch.stack(bands)


# ---- audiocraft 1.2.0 models/unet.py ---------------------------------------------------------------------------------------

@dataclass
class UnetConfig:
    chin: int = 1
    hidden: int = 48
    depth: int = 4
    growth: float = 4.0
    max_channels: int = 10_000
    num_steps: int = 1000
    emb_all_layers: bool = True
    codec_dim: Optional[int] = 128
    kernel: int = 8
    stride: int = 4
    norm_groups: int = 4
    res_blocks: int = 1
    activation: str = "relu"   # nn.ReLU is DiffusionUnet's default activation
    bilstm: bool = False
    transformer: bool = False
    cross_attention: bool = False

    def channels(self) -> List[int]:
        out, h = [], self.hidden
        for _ in range(self.depth):
            out.append(h)
            h = min(int(h * self.growth), self.max_channels)
        return out


def _act(x, kind):
    if kind == "relu":
        return torch.relu(x)
    if kind == "gelu":
        return F.gelu(x)
    if kind in ("silu", "swish"):
        return F.silu(x)
    raise ValueError(f"activation {kind}")


def _res_block(sd, p, x, cfg: UnetConfig, dilation: int):
    pad = dilation * (3 - 1) // 2
    h = F.group_norm(x, cfg.norm_groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-5)
    h = F.conv1d(_act(h, cfg.activation), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=pad, dilation=dilation)
    h = F.group_norm(h, cfg.norm_groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-5)
    h = F.conv1d(_act(h, cfg.activation), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=pad, dilation=dilation)
    return x + h


def _lstm_direction(x: torch.Tensor, w_ih, w_hh, bias, reverse: bool) -> torch.Tensor:
    """One direction of one nn.LSTM layer on (T, I) -> (T, H), explicit gates (order i, f, g, o)."""
    T, H = x.shape[0], w_hh.shape[1]
    h, c = torch.zeros(H), torch.zeros(H)
    gin = x @ w_ih.T + bias
    out = torch.zeros(T, H)
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        g = gin[t] + w_hh @ h
        i, f, gg, o = g[:H], g[H:2 * H], g[2 * H:3 * H], g[3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[t] = h
    return out


def blstm(sd: Dict[str, torch.Tensor], z: torch.Tensor) -> torch.Tensor:
    """audiocraft `BLSTM(dim)`: nn.LSTM(dim, dim, num_layers=2, bidirectional=True) over time, then Linear(2 dim -> dim); z (1, C, T)."""
    x = z[0].T  # (T, C)
    for l in range(2):
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            outs.append(_lstm_direction(x, sd[f"bilstm.lstm.weight_ih_l{l}{suffix}"], sd[f"bilstm.lstm.weight_hh_l{l}{suffix}"],
                                        sd[f"bilstm.lstm.bias_ih_l{l}{suffix}"] + sd[f"bilstm.lstm.bias_hh_l{l}{suffix}"], rev))
        x = torch.cat(outs, dim=1)
    y = x @ sd["bilstm.linear.weight"].T + sd["bilstm.linear.bias"]
    return y.T[None]


def unet_forward(sd: Dict[str, torch.Tensor], cfg: UnetConfig, x: torch.Tensor, step: int, condition: Optional[torch.Tensor]) -> torch.Tensor:
    """DiffusionUnet.forward(x (B, chin, T), step, condition (B, codec_dim, frames)).sample."""
    if cfg.transformer or cfg.cross_attention:
        raise NotImplementedError("transformer bottlenecks are not part of the 24 kHz multi-band models")
    sd = {k: v.float() for k, v in sd.items()}
    k, s = cfg.kernel, cfg.stride
    pad = (k - s) // 2
    skips = []
    z = x.float()
    for i in range(cfg.depth):
        T = z.shape[-1]
        z = F.pad(z, (0, (s - T % s) % s))
        p = f"encoders.{i}."
        z = F.conv1d(z, sd[p + "conv.weight"], None, stride=s, padding=pad)
        z = _act(F.group_norm(z, cfg.norm_groups, sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-5), cfg.activation)
        for r in range(cfg.res_blocks):
            z = _res_block(sd, p + f"res_blocks.{r}.", z, cfg, 2 ** r)
        if i == 0:
            z = z + sd["embedding.weight"][step].view(1, -1, 1)
        elif cfg.emb_all_layers:
            z = z + sd[f"embeddings.{i - 1}.weight"][step].view(1, -1, 1)
        skips.append(z)
    if cfg.codec_dim is not None:
        c = F.conv1d(condition.float(), sd["conv_codec.weight"], sd["conv_codec.bias"])
        assert c.shape[-1] <= 2 * z.shape[-1]
        z = z + F.interpolate(c, z.shape[-1])  # nearest
    if cfg.bilstm:
        assert z.shape[0] == 1
        z = blstm(sd, z)
    else:
        z = torch.zeros_like(z)  # `if self.bilstm is None: z = torch.zeros_like(z)` (unet.py, no-LSTM / no-transformer bottleneck)
    for i in range(cfg.depth):
        skip = skips.pop(-1)
        z = z[:, :, : skip.shape[2]] + skip
        p = f"decoders.{i}."
        for r in range(cfg.res_blocks):
            z = _res_block(sd, p + f"res_blocks.{r}.", z, cfg, 2 ** r)
        z = _act(F.group_norm(z, cfg.norm_groups, sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-5), cfg.activation)
        z = F.conv_transpose1d(z, sd[p + "convtr.weight"], None, stride=s, padding=pad)
    return z[:, :, : x.shape[2]]


# ---- audiocraft 1.2.0 modules/diffusion_schedule.py ------------------------------------------------------------------------

@dataclass
class ScheduleConfig:
    beta_t0: float = 1e-4
    beta_t1: float = 0.02
    num_steps: int = 1000
    beta_exp: float = 1.0
    clip: float = 5.0
    rescale: float = 1.0
    noise_scale: float = 1.0

    def betas(self) -> torch.Tensor:
        return torch.linspace(self.beta_t0 ** (1 / self.beta_exp), self.beta_t1 ** (1 / self.beta_exp), self.num_steps,
                              dtype=torch.float32) ** self.beta_exp


DEFAULT_STEP_LIST = list(range(1000))[::-50] + [0]  # 999, 949, ..., 49, 0: 20 model evaluations


def subsampled_coefficients(sch: ScheduleConfig, step_list: Sequence[int] = DEFAULT_STEP_LIST):
    """The scalars of `generate_subsampled`: per evaluated step (c_est, c_cur, sigma) with
    previous = c_cur * current - c_est * estimate (+ sigma * noise), evaluated in the reference's order of operations."""
    betas = sch.betas()
    abar_all = (1 - betas).cumprod(dim=0)
    sub = abar_all[list(reversed(step_list))]
    alphas = torch.cat([sub[:1], sub[1:] / sub[:-1]])
    betas_sub = 1 - alphas
    alpha_bar = (1 - betas[: sch.num_steps]).prod()  # get_alpha_bar(num_steps - 1)
    out = []
    for idx, step in enumerate(step_list[:-1]):
        alpha = 1 - betas_sub[-1 - idx]
        coef = dict(step=int(step), alpha=float(alpha), inv_sqrt_alpha=float(1.0 / alpha.sqrt()),
                    est=float((1 - alpha) / (1 - alpha_bar).sqrt()))
        if step == step_list[-2]:
            sigma2, prev_abar = 0.0, torch.tensor(1.0)
        else:
            prev_abar = (1 - betas[: step_list[idx + 1] + 1]).prod()
            sigma2 = float((1 - prev_abar) / (1 - alpha_bar) * (1 - alpha))
        coef["sigma"] = sigma2 ** 0.5
        out.append(coef)
        alpha_bar = prev_abar
    return out


def generate_subsampled(sd, cfg: UnetConfig, sch: ScheduleConfig, initial: torch.Tensor, condition: torch.Tensor,
                        noises: Sequence[torch.Tensor], step_list: Sequence[int] = DEFAULT_STEP_LIST) -> torch.Tensor:
    """NoiseSchedule.generate_subsampled without the sample processor.  `noises[i]` is the Gaussian draw of evaluation i
    (the last one is never used: sigma = 0 there)."""
    current = initial * sch.noise_scale
    for i, c in enumerate(subsampled_coefficients(sch, step_list)):
        estimate = unet_forward(sd, cfg, current, c["step"], condition) * sch.noise_scale
        previous = (current - c["est"] * estimate) * c["inv_sqrt_alpha"]
        if c["sigma"] > 0:
            previous = previous + c["sigma"] * noises[i] * sch.noise_scale
        if sch.clip:
            previous = previous.clamp(-sch.clip, sch.clip)
        current = previous
        if c["step"] == 0:
            current = current * sch.rescale
    return current


# ---- audiocraft 1.2.0 modules/diffusion_schedule.py: MultiBandProcessor ---------------------------------------------------------

def processor_return_sample(x: torch.Tensor, proc: Dict[str, torch.Tensor], sample_rate: int = 24_000, power_std: float = 1.0) -> torch.Tensor:
    """MultiBandProcessor.return_sample: split into the processor's mel bands, band * (std / target_std) ** power_std + mean, sum.
    `proc` = the processor's buffers {counts, sum_x, sum_x2, sum_target_x2}."""
    n_bands = proc["sum_x"].numel()
    counts = proc["counts"].float()
    mean = proc["sum_x"].float() / counts
    std = (proc["sum_x2"].float() / counts - mean ** 2).clamp(min=0).sqrt()
    target_std = proc["sum_target_x2"].float() / counts
    bands = split_bands(x, n_bands, sample_rate)
    rescale = (std / target_std) ** power_std
    return (bands * rescale.view(-1, 1, 1, 1) + mean.view(-1, 1, 1, 1)).sum(dim=0)


def re_eq(wav: torch.Tensor, ref: torch.Tensor, n_bands: int = 32, sample_rate: int = 24_000, strictness: float = 1.0) -> torch.Tensor:
    """MultiBandDiffusion.re_eq: every mel band of `wav` rescaled to the standard deviation of the same band of `ref`."""
    bands, bands_ref = split_bands(wav, n_bands, sample_rate), split_bands(ref, n_bands, sample_rate)
    out = torch.zeros_like(ref)
    for i in range(n_bands):
        out = out + bands[i] * (bands_ref[i].std() / bands[i].std()) ** strictness
    return out


# ---- audiocraft 1.2.0 models/multibanddiffusion.py -------------------------------------------------------------------------------

@dataclass
class BandModel:
    sd: Dict[str, torch.Tensor]
    cfg: UnetConfig = field(default_factory=UnetConfig)
    schedule: ScheduleConfig = field(default_factory=ScheduleConfig)
    processor: Optional[Dict[str, torch.Tensor]] = None
    power_std: float = 1.0


def generate(bands: Sequence[BandModel], condition: torch.Tensor, size, initial_noises: Sequence[torch.Tensor],
             step_noises: Sequence[Sequence[torch.Tensor]], sample_rate: int = 24_000) -> torch.Tensor:
    """MultiBandDiffusion.generate: sum over the band models of DiffusionProcess.generate(condition, initial_noise)."""
    out = torch.zeros(size)
    for b, init, noises in zip(bands, initial_noises, step_noises):
        x = generate_subsampled(b.sd, b.cfg, b.schedule, init, condition, noises)
        if b.processor is not None:
            x = processor_return_sample(x, b.processor, sample_rate, b.power_std)
        out = out + x
    return out


def tokens_to_wav(bands: Sequence[BandModel], wav_encodec: torch.Tensor, latent: torch.Tensor, initial_noises, step_noises,
                  n_eq_bands: int = 32, sample_rate: int = 24_000) -> torch.Tensor:
    """MultiBandDiffusion.tokens_to_wav given the codec's decode / decode_latent outputs (oracle/codec.py)."""
    wav = generate(bands, latent, wav_encodec.shape, initial_noises, step_noises, sample_rate)
    return re_eq(wav, wav_encodec, n_eq_bands, sample_rate)
