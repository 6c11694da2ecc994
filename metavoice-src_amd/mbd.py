"""Host mirror of the parts of audiocraft==1.2.0 the reference's vocoder call runs after the codec decode
(`fam/llm/decoders.py:13,84-85` -> `MultiBandDiffusion.tokens_to_wav`): loading the multi-band diffusion package, the scalar noise
schedule of `NoiseSchedule.generate_subsampled`, the `MultiBandProcessor` statistics and the julius low-pass banks -- configuration
work done once at load time, like the reference does it on the host -- and the calls into libmetavoice_hip.so (`mv_mbd_*`,
csrc/mbd.hip) where every per-sample operation runs.

Package layout (audiocraft `load_diffusion_models`): {"sample_rate": 24000, "n_bands": 4, 0..3: {"model_state": DiffusionUnet state
dict, "processor_state": MultiBandProcessor buffers, "cfg": {"diffusion_unet": {...}, "schedule": {...}, "processor": {...}}}}.
`cfg` may be an OmegaConf object (real checkpoint; needs omegaconf to unpickle) or plain dicts (synthetic.mbd_package).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _native as N

_ACT = {"relu": 0, "gelu": 1, "silu": 2, "swish": 2}


def _get(cfg, key, default=None):
    try:
        v = cfg[key] if key in cfg else default
    except TypeError:
        v = getattr(cfg, key, default)
    return default if v is None else v


def mel_cutoffs(n_bands: int, sample_rate: int) -> List[float]:
    """julius.bands.SplitBands: `mel_frequencies(n_bands + 1, 0, sr / 2)[1:-1]` (HTK mel scale)."""
    hi = 2595.0 * math.log10(1.0 + (sample_rate / 2.0) / 700.0)
    return [700.0 * (10.0 ** ((hi * i / n_bands) / 2595.0) - 1.0) for i in range(1, n_bands)]


def lowpass_bank(n_bands: int, sample_rate: int, zeros: float = 8.0) -> np.ndarray:
    """julius.LowPassFilters for SplitBands(n_bands): (n_bands - 1, 2 half + 1) Hann-windowed sinc filters, unit DC gain."""
    cut = [c / sample_rate for c in mel_cutoffs(n_bands, sample_rate)]
    half = int(zeros / min(cut) / 2)
    t = np.arange(-half, half + 1, dtype=np.float64)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(2 * half + 1) / (2 * half))  # torch.hann_window(periodic=False)
    bank = []
    for c in cut:
        x = 2 * c * np.pi * t
        sinc = np.where(x == 0, 1.0, np.sin(x) / np.where(x == 0, 1.0, x))
        f = 2 * c * window * sinc
        bank.append(f / f.sum())
    return np.ascontiguousarray(np.stack(bank).astype(np.float32))


def subsampled_schedule(sch, step_list: Sequence[int]):
    """The scalars `NoiseSchedule.generate_subsampled` derives from its betas, in torch fp32 like the reference
    (audiocraft modules/diffusion_schedule.py): per evaluation (step, est coefficient, 1/sqrt(alpha), sigma)."""
    num_steps = int(_get(sch, "num_steps", 1000))
    beta_exp = float(_get(sch, "beta_exp", 1.0))
    betas = torch.linspace(float(_get(sch, "beta_t0", 1e-4)) ** (1 / beta_exp), float(_get(sch, "beta_t1", 0.02)) ** (1 / beta_exp), num_steps,
                           dtype=torch.float32) ** beta_exp
    sub = (1 - betas).cumprod(dim=0)[list(reversed(step_list))]
    betas_sub = 1 - torch.cat([sub[:1], sub[1:] / sub[:-1]])
    alpha_bar = (1 - betas).prod()
    steps, est, isa, sig = [], [], [], []
    for idx, step in enumerate(step_list[:-1]):
        alpha = 1 - betas_sub[-1 - idx]
        steps.append(int(step))
        est.append(float((1 - alpha) / (1 - alpha_bar).sqrt()))
        isa.append(float(1.0 / alpha.sqrt()))
        if step == step_list[-2]:
            sigma2, prev = 0.0, torch.tensor(1.0)
        else:
            prev = (1 - betas[: step_list[idx + 1] + 1]).prod()
            sigma2 = float((1 - prev) / (1 - alpha_bar) * (1 - alpha))
        sig.append(sigma2 ** 0.5)
        alpha_bar = prev
    return steps, est, isa, sig


class MultiBandDiffusionBands:
    """The diffusion half of `MultiBandDiffusion`: `tokens_to_wav(latent, wav_encodec)` = re_eq(sum of the band processes)."""

    def __init__(self, handle, n_models: int, n_eval: int, sample_rate: int, seed: int):
        self._h, self.n_models, self.n_eval, self.sample_rate = handle, n_models, n_eval, sample_rate
        self._seed, self._calls = seed, 0

    @classmethod
    def from_package(cls, pkg: dict, seed: int = 1234, precision: int = 3, step_list: Optional[Sequence[int]] = None,
                     eq_bands: int = 32) -> "MultiBandDiffusionBands":
        n_models = int(pkg["n_bands"])
        sample_rate = int(pkg.get("sample_rate", 24_000))
        step_list = list(step_list) if step_list is not None else list(range(1000))[::-50] + [0]
        u0 = _get(pkg[0]["cfg"], "diffusion_unet", {})
        hidden, depth, growth = int(_get(u0, "hidden", 48)), int(_get(u0, "depth", 4)), float(_get(u0, "growth", 4.0))
        max_ch = int(_get(u0, "max_channels", 10_000))
        if _get(u0, "transformer", False) or _get(u0, "cross_attention", False):
            raise NotImplementedError("transformer / cross-attention bottlenecks are not supported")
        proc0 = _get(pkg[0]["cfg"], "processor", {})
        proc_bands = int(_get(proc0, "n_bands", 8)) if _get(proc0, "use", True) and pkg[0].get("processor_state") else 1
        cfg = N.mv_mbd_config()
        cfg.n_models, cfg.chin, cfg.depth = n_models, 1, depth
        ch = hidden
        for i in range(depth):
            cfg.channels[i] = ch
            ch = min(int(ch * growth), max_ch)
        cfg.kernel, cfg.stride = int(_get(u0, "kernel", 8)), int(_get(u0, "stride", 4))
        cfg.norm_groups, cfg.res_blocks = int(_get(u0, "norm_groups", 4)), int(_get(u0, "res_blocks", 1))
        cfg.act = _ACT[str(_get(u0, "activation", "relu")).lower()]
        cfg.codec_dim = int(_get(u0, "codec_dim", 0) or 0)
        cfg.emb_all_layers = int(bool(_get(u0, "emb_all_layers", False)))
        cfg.bilstm = int(bool(_get(u0, "bilstm", False)))
        cfg.num_steps = int(_get(_get(pkg[0]["cfg"], "schedule", {}), "num_steps", 1000))
        cfg.n_eval, cfg.proc_bands, cfg.eq_bands, cfg.precision = len(step_list) - 1, proc_bands, eq_bands, precision
        h = C.c_void_p()
        N.check(N.lib().mv_mbd_create(C.byref(cfg), C.byref(h)))
        self = cls(h, n_models, cfg.n_eval, sample_rate, seed)
        self.codec_dim = cfg.codec_dim
        fp = C.POINTER(C.c_float)
        for b in range(n_models):
            entry = pkg[b]
            for name, t in entry["model_state"].items():
                if "num_batches_tracked" in name:
                    continue
                a = np.ascontiguousarray(t.detach().float().cpu().numpy())
                N.check(N.lib().mv_mbd_load_weight(h, b, name.encode(), a.ctypes.data_as(fp), a.size))
            sch = _get(entry["cfg"], "schedule", {})
            steps, est, isa, sig = subsampled_schedule(sch, step_list)
            n = len(steps)
            N.check(N.lib().mv_mbd_set_schedule(h, b, (C.c_int32 * n)(*steps), (C.c_float * n)(*est), (C.c_float * n)(*isa), (C.c_float * n)(*sig),
                                                float(_get(sch, "clip", 5.0) or 0.0), float(_get(sch, "rescale", 1.0)), float(_get(sch, "noise_scale", 1.0))))
            if proc_bands > 1:
                ps = entry["processor_state"]
                counts = ps["counts"].float()
                mean = ps["sum_x"].float() / counts
                std = (ps["sum_x2"].float() / counts - mean ** 2).clamp(min=0).sqrt()
                target = ps["sum_target_x2"].float() / counts
                power = float(_get(_get(entry["cfg"], "processor", {}), "power_std", 1.0))
                scale = ((std / target) ** power).numpy().astype(np.float32)
                mean = mean.numpy().astype(np.float32)
                N.check(N.lib().mv_mbd_set_processor(h, b, proc_bands, scale.ctypes.data_as(fp), mean.ctypes.data_as(fp)))
        for which, nb in ((0, proc_bands), (1, eq_bands)):
            if nb > 1:
                bank = lowpass_bank(nb, sample_rate)
                N.check(N.lib().mv_mbd_set_filterbank(h, which, bank.shape[0], bank.shape[1], bank.ctypes.data_as(fp)))
        N.check(N.lib().mv_mbd_finalize(h))
        return self

    def close(self):
        for lane in getattr(self, "_lanes", []):   # clones share this handle's weights: they go first
            lane.close()
        self._lanes = []
        if getattr(self, "_h", None):
            N.lib().mv_mbd_destroy(self._h)
            self._h = None

    def clone(self) -> "MultiBandDiffusionBands":
        """A second handle on the same device weights with its own stream and workspace (`mv_mbd_clone`)."""
        h = C.c_void_p()
        N.check(N.lib().mv_mbd_clone(self._h, C.byref(h)))
        other = MultiBandDiffusionBands(h, self.n_models, self.n_eval, self.sample_rate, self._seed)
        other.codec_dim = self.codec_dim
        return other

    def tokens_to_wav_many(self, items: Sequence, lanes: int = 4) -> List[torch.Tensor]:
        """`tokens_to_wav` for the utterances of a batch -- items = [(latent (1, codec_dim, frames), wav_encodec (1, 1, samples)), ...] -- on
        up to `lanes` handles at once (this one + clones, one host thread each: the C call releases the GIL).  Every utterance gets the
        seed the one-at-a-time path would have given it (call order), so the waveforms are the same samples."""
        from concurrent.futures import ThreadPoolExecutor

        n = len(items)
        if n == 0:
            return []
        if not hasattr(self, "_lanes"):
            self._lanes = []
        while len(self._lanes) + 1 < min(lanes, n):
            self._lanes.append(self.clone())
        handles = [self] + self._lanes[:max(0, min(lanes, n) - 1)]
        seeds = [(self._seed + 7919 * (self._calls + 1 + i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]
        self._calls += n
        fp = C.POINTER(C.c_float)

        def one(i: int, h) -> torch.Tensor:
            latent, wav_encodec = items[i]
            T = int(wav_encodec.shape[-1])
            lat = np.ascontiguousarray(latent[0].detach().float().cpu().numpy()) if self.codec_dim else None
            ref = np.ascontiguousarray(wav_encodec.reshape(-1).detach().float().cpu().numpy())
            out = np.empty(T, dtype=np.float32)
            N.check(N.lib().mv_mbd_generate(h._h, lat.ctypes.data_as(fp) if lat is not None else None, lat.shape[1] if lat is not None else 0,
                                            ref.ctypes.data_as(fp), T, None, seeds[i], out.ctypes.data_as(fp)))
            return torch.from_numpy(out).view(1, 1, -1)

        if len(handles) == 1:
            return [one(i, self) for i in range(n)]

        def lane_work(k: int):   # lane k takes utterances k, k + L, ...: a handle serves one call at a time
            return [(i, one(i, handles[k])) for i in range(k, n, len(handles))]

        out: List[Optional[torch.Tensor]] = [None] * n
        with ThreadPoolExecutor(max_workers=len(handles)) as pool:
            for part in pool.map(lane_work, range(len(handles))):
                for i, w in part:
                    out[i] = w
        return out

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tokens_to_wav(self, latent: torch.Tensor, wav_encodec: Optional[torch.Tensor], noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """latent (1, codec_dim, frames) and wav_encodec (1, 1, samples) from the codec -> (1, 1, samples).
        `noise`: (n_models, n_eval + 1, samples) injected Gaussian draws (initial noise, then one per evaluation)."""
        fp = C.POINTER(C.c_float)
        T = int(wav_encodec.shape[-1]) if wav_encodec is not None else int(noise.shape[-1])
        lat = np.ascontiguousarray(latent[0].detach().float().cpu().numpy()) if latent is not None else None
        ref = np.ascontiguousarray(wav_encodec.reshape(-1).detach().float().cpu().numpy()) if wav_encodec is not None else None
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(noise.detach().float().cpu().numpy())
            assert nz.shape == (self.n_models, self.n_eval + 1, T)
        out = np.empty(T, dtype=np.float32)
        self._calls += 1
        N.check(N.lib().mv_mbd_generate(self._h, lat.ctypes.data_as(fp) if lat is not None else None, lat.shape[1] if lat is not None else 0,
                                        ref.ctypes.data_as(fp) if ref is not None else None, T, nz.ctypes.data_as(fp) if nz is not None else None,
                                        (self._seed + 7919 * self._calls) & 0xFFFFFFFFFFFFFFFF, out.ctypes.data_as(fp)))
        return torch.from_numpy(out).view(1, 1, -1)

    def tokens_to_wav_device(self, d_latent: Optional[int], frames: int, d_wav_encodec: Optional[int], samples: int) -> torch.Tensor:
        """`tokens_to_wav` on the codec's DEVICE buffers (`codec.EncodecDecoder.decode_device`): no host round trip between the
        codec and the diffusion bands; the engine's own Philox noise stream.  -> (1, 1, samples) fp32 on the host."""
        out = np.empty(samples, dtype=np.float32)
        self._calls += 1
        N.check(N.lib().mv_mbd_generate_device(self._h, C.c_void_p(d_latent if self.codec_dim else None), frames if self.codec_dim else 0,
                                               C.c_void_p(d_wav_encodec), samples, (self._seed + 7919 * self._calls) & 0xFFFFFFFFFFFFFFFF,
                                               out.ctypes.data_as(C.POINTER(C.c_float))))
        return torch.from_numpy(out).view(1, 1, -1)

    def unet_forward(self, band: int, x: torch.Tensor, step: int, latent: Optional[torch.Tensor]) -> torch.Tensor:
        """Parity hook: one DiffusionUnet evaluation, x (samples,) -> (samples,)."""
        fp = C.POINTER(C.c_float)
        xa = np.ascontiguousarray(x.reshape(-1).float().numpy())
        lat = np.ascontiguousarray(latent[0].float().numpy()) if latent is not None else None
        out = np.empty_like(xa)
        N.check(N.lib().mv_mbd_unet_forward(self._h, band, xa.ctypes.data_as(fp), xa.size, int(step), lat.ctypes.data_as(fp) if lat is not None else None,
                                            lat.shape[1] if lat is not None else 0, out.ctypes.data_as(fp)))
        return torch.from_numpy(out)

    def unet_forward_sum(self, x: torch.Tensor, step: int, latent: Optional[torch.Tensor]) -> torch.Tensor:
        """Parity hook at the launch shapes of `tokens_to_wav`: one evaluation of EVERY band model in one pass (all bands per launch --
        which is what decides the GEMM kernel a layer runs on); x (n_models, samples) -> (samples,) = sum over the bands of unet_b(x_b)."""
        fp = C.POINTER(C.c_float)
        xa = np.ascontiguousarray(x.float().numpy())
        assert xa.ndim == 2 and xa.shape[0] == self.n_models
        lat = np.ascontiguousarray(latent[0].float().numpy()) if latent is not None else None
        out = np.empty(xa.shape[1], dtype=np.float32)
        N.check(N.lib().mv_mbd_unet_forward_sum(self._h, xa.ctypes.data_as(fp), xa.shape[1], int(step), lat.ctypes.data_as(fp) if lat is not None else None,
                                                lat.shape[1] if lat is not None else 0, out.ctypes.data_as(fp)))
        return torch.from_numpy(out)

    GEMM_KERNELS = ("mbd_gemm_t256_kernel", "mbd_gemm_glds3_kernel<64,192>", "mbd_gemm_glds3_kernel<128,128>", "mbd_gemm_thin_kernel",
                    "mbd_gemm_glds_kernel", "mbd_gemm_kernel<128>", "mbd_gemm_kernel<64>")

    def kernel_counts(self) -> dict:
        """Launches per convolution-GEMM kernel since the engine was built (the parity tests assert which kernels a shape ran on)."""
        out = (C.c_int64 * 8)()
        N.check(N.lib().mv_mbd_kernel_counts(self._h, out))
        return {k: int(out[i]) for i, k in enumerate(self.GEMM_KERNELS)}

    def fused_skip_additions(self) -> int:
        """Transposed convolutions since the engine was built whose epilogue wrote the decoder's `z + skip` (no pass of its own for the sum)."""
        out = (C.c_int64 * 8)()
        N.check(N.lib().mv_mbd_kernel_counts(self._h, out))
        return int(out[7])
