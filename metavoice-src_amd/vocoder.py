"""The vocoder `fam/llm/decoders.py:13` builds at import time -- `MultiBandDiffusion.get_mbd_24khz(bw=6)` (audiocraft==1.2.0,
third party) -- on the MI355X engine: EnCodec 24 kHz decode (`codec.py`, csrc/codec.hip) and, when their weights are present,
the four diffusion bands + 32-band re-equalisation (`mbd.py`, csrc/mbd.hip).

The reference downloads both checkpoints from the hub; there is no network here, so they are looked up in the model directory
next to first_stage.pt:

    encodec_24khz.pt   state dict of the EnCodec 24 kHz model: transformers' `EncodecModel` names, or the encodec / audiocraft
                       package names (`decoder.model.3.convtr.convtr.weight_g`, `quantizer.vq.layers.0._codebook.embed`, ...),
                       bare or under "state_dict" / "best_state" / "model"
    mbd_comp_8.pt      audiocraft's multi-band diffusion package ({i: {"model_state", "processor_state", "cfg"}, "n_bands", ...})

`get_mbd_24khz` raises FileNotFoundError when the codec weights are missing: a TTS without a vocoder cannot synthesise, so it
fails at construction, not at the first request.
"""
from __future__ import annotations

import os
import re
import warnings
from typing import Dict, Optional

import torch

from .codec import EncodecDecoder

ENCODEC_FILE = "encodec_24khz.pt"
MBD_FILE = "mbd_comp_8.pt"


def encodec_names_to_transformers(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """encodec / audiocraft package tensor names -> transformers' `EncodecModel` names (what `mv_codec_load_weight` expects).
    Names that already follow transformers pass through."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        n = k
        n = re.sub(r"^(encoder|decoder)\.model\.", r"\1.layers.", n)
        n = n.replace(".convtr.convtr.", ".conv.").replace(".conv.conv.", ".conv.")
        n = re.sub(r"^quantizer\.vq\.layers\.(\d+)\._codebook\.", r"quantizer.layers.\1.codebook.", n)
        out[n] = v
    return out


def _unwrap(obj) -> Dict[str, torch.Tensor]:
    for key in ("state_dict", "best_state", "model"):
        if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
            obj = obj[key]
    return obj


class MultiBandDiffusion:
    """The part of audiocraft's `MultiBandDiffusion` the reference uses: `tokens_to_wav` (`decoders.py:84-85`)."""

    def __init__(self, codec_model: EncodecDecoder, diffusion=None):
        self.codec_model = codec_model
        self.diffusion = diffusion   # mbd.MultiBandDiffusionBands or None
        self.sample_rate = 24_000

    @staticmethod
    def get_mbd_24khz(bw: float = 6.0, model_dir: Optional[str] = None, n_q: Optional[int] = None, seed: int = 1234) -> "MultiBandDiffusion":
        assert bw in (1.5, 3.0, 6.0), "bandwidth not available"
        n_q = n_q or {1.5: 2, 3.0: 4, 6.0: 8}[bw]
        if model_dir is None or not os.path.isfile(os.path.join(model_dir, ENCODEC_FILE)):
            raise FileNotFoundError(
                f"{ENCODEC_FILE} not found in {model_dir!r}: the EnCodec 24 kHz weights (facebook/encodec_24khz) must sit next to "
                "first_stage.pt -- the reference downloads them from the hub at import time (fam/llm/decoders.py:13)")
        sd = _unwrap(torch.load(os.path.join(model_dir, ENCODEC_FILE), map_location="cpu", weights_only=False))
        codec = EncodecDecoder.from_state_dict(encodec_names_to_transformers(sd), n_q=n_q)
        diffusion = None
        mbd_path = os.path.join(model_dir, MBD_FILE)
        if os.path.isfile(mbd_path):
            from .mbd import MultiBandDiffusionBands

            diffusion = MultiBandDiffusionBands.from_package(torch.load(mbd_path, map_location="cpu", weights_only=False), seed=seed)
        else:
            warnings.warn(f"{MBD_FILE} not found in {model_dir}: waveforms are EnCodec's own reconstruction, without the "
                          "multi-band diffusion refinement of the reference")
        return MultiBandDiffusion(codec, diffusion)

    def tokens_to_wav(self, tokens: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(1, n_q, frames) int -> (1, 1, frames * 320) fp32 at 24 kHz.  `noise`: injected Gaussian tensors for the diffusion
        bands (parity tests); default = the engine's own Philox stream."""
        if self.diffusion is None:
            return self.codec_model.decode(tokens)
        if noise is None and hasattr(self.codec_model, "decode_device") and hasattr(self.diffusion, "tokens_to_wav_device"):
            # the codec's waveform and latent stay in device memory for the diffusion bands (no PCIe leg between the stages)
            d_wav, d_lat, samples, frames = self.codec_model.decode_device(tokens)
            return self.diffusion.tokens_to_wav_device(d_lat, frames, d_wav, samples)
        wav_encodec, latent = self.codec_model.decode(tokens, return_latent=True)
        return self.diffusion.tokens_to_wav(latent, wav_encodec, noise=noise)  # band processes, sum, re_eq against wav_encodec

    __call__ = tokens_to_wav

    def tokens_to_wav_many(self, tokens_list, lanes: int = 4):
        """`tokens_to_wav` for the utterances of a batch: the EnCodec decodes one after the other (4-5 ms each), the diffusion bands of up to
        `lanes` utterances concurrently (`mbd.MultiBandDiffusionBands.tokens_to_wav_many`); the same samples as one call per utterance."""
        if self.diffusion is None or not hasattr(self.diffusion, "tokens_to_wav_many"):
            return [self.tokens_to_wav(t) for t in tokens_list]
        items = []
        for t in tokens_list:
            wav_encodec, latent = self.codec_model.decode(t, return_latent=True)
            items.append((latent, wav_encodec))
        return self.diffusion.tokens_to_wav_many(items, lanes=lanes)

    def close(self):
        self.codec_model.close()
        if self.diffusion is not None:
            self.diffusion.close()
