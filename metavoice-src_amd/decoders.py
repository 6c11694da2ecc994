"""Mirror of `fam/llm/decoders.py`: token grid -> waveform file.

The reference builds `audiocraft.MultiBandDiffusion.get_mbd_24khz(bw=6)` at import time (`decoders.py:13`) and calls
`mbd.tokens_to_wav` (`:84-85`).  Here the vocoder is `vocoder.MultiBandDiffusion` (HIP EnCodec decode + diffusion bands), built by
`TTS` from the weights in the model directory and handed in as a callable
`vocoder(tokens: LongTensor (1, 8, f)) -> FloatTensor (1, 1, samples)` at 24 kHz.  There is no CPU fallback: an
`EncodecDecoder` without a vocoder is refused at construction.
"""
from __future__ import annotations

import os
import pathlib
import struct
import uuid
from abc import ABC, abstractmethod
from datetime import datetime
from typing import Callable, List, Optional, Tuple, Union

import torch

from .loudness import normalize_loudness


class Decoder(ABC):
    @abstractmethod
    def decode(self, tokens: List[int], ref_audio_path: Optional[str] = None, causal: Optional[bool] = None):
        raise NotImplementedError


def write_wav_pcm16(path, wav: torch.Tensor, sample_rate: int) -> None:
    """Mono 16-bit PCM RIFF writer (stands in for audiocraft's `audio_write`, `decoders.py:40-47`); `path` may be a file
    name or a binary file object."""
    x = wav.detach().reshape(-1).float().cpu().clamp(-1.0, 1.0)
    pcm = (x * 32767.0).round().to(torch.int16).numpy().tobytes()
    blob = (b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16) + b"data" + struct.pack("<I", len(pcm)) + pcm)
    if hasattr(path, "write"):
        path.write(blob)
    else:
        with open(path, "wb") as f:
            f.write(blob)


class EncodecDecoder(Decoder):
    def __init__(self, tokeniser_decode_fn: Callable[[List[int]], str],
                 data_adapter_fn: Callable[[List[List[int]]], Tuple[List[int], List[List[int]]]], output_dir: str,
                 vocoder: Optional[Callable[[torch.Tensor], torch.Tensor]] = None):
        self._mbd_sample_rate = 24_000
        self._end_of_audio_token = 1024
        self._num_codebooks = 8
        if vocoder is None:
            raise ValueError("EncodecDecoder needs a vocoder (vocoder.MultiBandDiffusion.get_mbd_24khz(model_dir=...)); "
                             "the reference builds one at import time (fam/llm/decoders.py:13)")
        self.vocoder = vocoder
        self.tokeniser_decode_fn = tokeniser_decode_fn
        self._data_adapter_fn = data_adapter_fn
        self.output_dir = pathlib.Path(output_dir).resolve()
        os.makedirs(self.output_dir, exist_ok=True)

    def _save_audio(self, name, wav: torch.Tensor):
        # audio_write(..., strategy="loudness", loudness_compressor=True) of the reference (`decoders.py:40-47`): BS.1770
        # loudness to -14 LKFS + tanh compressor (loudness.py), then 16-bit PCM; waveform parity is stated before this step
        out = normalize_loudness(wav.detach().float().cpu().reshape(1, -1), self._mbd_sample_rate, loudness_headroom_db=14.0,
                                 loudness_compressor=True)
        write_wav_pcm16(str(name) + ".wav", out, self._mbd_sample_rate)

    def _grid(self, tokens: List[List[int]]) -> Tuple[str, torch.Tensor]:
        text_ids, extracted_audio_ids = self._data_adapter_fn(tokens)
        text = self.tokeniser_decode_fn(text_ids)
        grid = torch.tensor(extracted_audio_ids, dtype=torch.long).unsqueeze(0)
        if grid.shape[1] < self._num_codebooks:
            pad = torch.zeros(1, self._num_codebooks - grid.shape[1], grid.shape[2], dtype=torch.long)
            grid = torch.cat([grid, pad], dim=1)
        return text, grid

    def decode_wav(self, tokens: List[List[int]]) -> torch.Tensor:
        """The waveform `decode(causal=False)` would save, (1, 1, samples) fp32, without touching the disk (long-form
        synthesis concatenates these; serving returns them through `wav_bytes`)."""
        wav = self.vocoder(self._grid(tokens)[1])
        if wav.shape[-1] < 9600:
            raise Exception("wav predicted is shorter than 400ms!")  # decoders.py:88-91
        return wav

    def decode_wavs(self, token_lists: List[List[List[int]]], return_exceptions: bool = False) -> list:
        """`decode_wav` for the utterances of a batch: the vocoder runs their diffusion bands on concurrent lanes where it can
        (`vocoder.MultiBandDiffusion.tokens_to_wav_many`; the same samples as one call per utterance).  With `return_exceptions` an
        utterance that fails ('wav predicted is shorter than 400ms!') leaves its exception in its slot."""
        out: list = [None] * len(token_lists)
        grids, idx = [], []
        for i, t in enumerate(token_lists):   # the data adapter / grid of one utterance failing is that utterance's
            try:
                grids.append(self._grid(t)[1])
                idx.append(i)
            except Exception as e:
                if not return_exceptions:
                    raise
                out[i] = e
        many = getattr(self.vocoder, "tokens_to_wav_many", None)
        wavs: list
        try:
            wavs = many(grids) if many is not None and len(grids) > 1 else [self.vocoder(g) for g in grids]
        except Exception:
            if not return_exceptions:
                raise
            # a lane of the grouped call failed (a native error on one utterance's band): vocode one by one so that the failure stays
            # with the utterance that caused it and the others of the group still come out
            wavs = []
            for g in grids:
                try:
                    wavs.append(self.vocoder(g))
                except Exception as e:
                    wavs.append(e)
        for i, wav in zip(idx, wavs):
            if not isinstance(wav, Exception) and wav.shape[-1] < 9600:
                wav = Exception("wav predicted is shorter than 400ms!")  # decoders.py:88-91
                if not return_exceptions:
                    raise wav
            out[i] = wav
        return out

    def wav_bytes(self, wav: torch.Tensor) -> bytes:
        """Loudness-normalised 16-bit PCM RIFF bytes of `wav`: what `_save_audio` writes, kept in memory."""
        import io

        out = normalize_loudness(wav.detach().float().cpu().reshape(1, -1), self._mbd_sample_rate, loudness_headroom_db=14.0,
                                 loudness_compressor=True)
        buf = io.BytesIO()
        write_wav_pcm16(buf, out, self._mbd_sample_rate)
        return buf.getvalue()

    def decode(self, tokens: List[List[int]], causal: bool = True, ref_audio_path: Optional[str] = None) -> Union[str, torch.Tensor]:
        """`decoders.py:66-102`: adapter -> (1, n, f) grid padded with 0 up to 8 codebooks; causal=True returns the grid,
        causal=False vocodes and SAVES `<output_dir>/synth_<timestamp>_<text[:25]>_<uuid>.wav`, returning the path stem."""
        text, grid = self._grid(tokens)
        if causal:
            return grid
        wav = self.vocoder(grid)
        if wav.shape[-1] < 9600:
            raise Exception("wav predicted is shorter than 400ms!")
        stamp = datetime.now().strftime("%y-%m-%d--%H-%M-%S")
        try:
            name = self.output_dir / f"synth_{stamp}_{text.replace(' ', '_')[:25]}_{uuid.uuid4()}"
            self._save_audio(name, wav)
            return name
        except Exception as e:
            print(f"Failed to save audio! Reason: {e}")
            name = self.output_dir / f"synth_{stamp}_{uuid.uuid4()}"
            self._save_audio(name, wav)
            return name
