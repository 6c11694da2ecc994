"""Mirror of `fam/quantiser/audio/speaker_encoder/model.py:21-117` on the MI355X engine: same class, constructor keywords and
methods; the 3-layer LSTM + projection (`forward`) runs in libmetavoice_hip.so (`mv_spkenc_*`, csrc/lstm.hip: one persistent
launch per layer), the mel front end and the slicing into partial utterances are host pre-processing like in the reference."""
from __future__ import annotations

import ctypes as C
from time import perf_counter as timer
from typing import List, Optional, Union

import numpy as np
import torch

from .. import _native as N
from . import audio

mel_window_step = 10
mel_n_channels = 40
sampling_rate = 16000
partials_n_frames = 160
model_hidden_size = 256
model_embedding_size = 256
model_num_layers = 3


class SpeakerEncoder:
    def __init__(self, weights_fpath: Optional[str] = None, device: Optional[Union[str, torch.device]] = None, verbose: bool = True,
                 eval: bool = False, state_dict: Optional[dict] = None):
        start = timer()
        if state_dict is None:
            checkpoint = torch.load(weights_fpath, map_location="cpu", weights_only=False)
            state_dict = checkpoint["model_state"]  # model.py:48-49 (strict=False there: extra keys are ignored)
        self.device = torch.device(device) if isinstance(device, str) else device
        cfg = N.mv_spkenc_config(n_mels=mel_n_channels, hidden=model_hidden_size, n_layers=model_num_layers, emb=model_embedding_size)
        self._h = C.c_void_p()
        N.check(N.lib().mv_spkenc_create(C.byref(cfg), C.byref(self._h)))
        for name, t in state_dict.items():
            if not (name.startswith("lstm.") or name.startswith("linear.")):
                continue
            a = np.ascontiguousarray(t.detach().float().cpu().numpy())
            N.check(N.lib().mv_spkenc_load_weight(self._h, name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size))
        N.check(N.lib().mv_spkenc_finalize(self._h))
        if verbose:
            print("Loaded the speaker embedding model on the GPU in %.2f seconds." % (timer() - start))

    def close(self):
        if getattr(self, "_h", None):
            N.lib().mv_spkenc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, mels: torch.Tensor) -> torch.Tensor:
        """`model.py:54-57`: (n_partials, frames, 40) -> (n_partials, 256), rows L2-normalised."""
        m = np.ascontiguousarray(mels.detach().float().cpu().numpy())
        assert m.ndim == 3 and m.shape[2] == mel_n_channels
        out = np.empty((m.shape[0], model_embedding_size), dtype=np.float32)
        N.check(N.lib().mv_spkenc_embed(self._h, m.ctypes.data_as(C.POINTER(C.c_float)), m.shape[0], m.shape[1],
                                        out.ctypes.data_as(C.POINTER(C.c_float))))
        return torch.from_numpy(out)

    __call__ = forward

    @staticmethod
    def compute_partial_slices(n_samples: int, rate, min_coverage):
        """`model.py:59-84`: partial utterances of 160 frames every round(16000 / rate / 160) frames; the last one is dropped
        when it covers less than `min_coverage` of real signal."""
        samples_per_frame = int(sampling_rate * mel_window_step / 1000)
        n_frames = int(np.ceil((n_samples + 1) / samples_per_frame))
        frame_step = int(np.round((sampling_rate / rate) / samples_per_frame))
        wav_slices, mel_slices = [], []
        steps = max(1, n_frames - partials_n_frames + frame_step + 1)
        for i in range(0, steps, frame_step):
            mel_range = np.array([i, i + partials_n_frames])
            wav_range = mel_range * samples_per_frame
            mel_slices.append(slice(*mel_range))
            wav_slices.append(slice(*wav_range))
        last = wav_slices[-1]
        coverage = (n_samples - last.start) / (last.stop - last.start)
        if coverage < min_coverage and len(mel_slices) > 1:
            mel_slices, wav_slices = mel_slices[:-1], wav_slices[:-1]
        return wav_slices, mel_slices

    def embed_utterance(self, wav: np.ndarray, return_partials=False, rate=1.3, min_coverage=0.75, numpy: bool = True):
        """`model.py:86-107`."""
        wav_slices, mel_slices = self.compute_partial_slices(len(wav), rate, min_coverage)
        max_wave_length = wav_slices[-1].stop
        if max_wave_length >= len(wav):
            wav = np.pad(wav, (0, max_wave_length - len(wav)), "constant")
        mel = audio.wav_to_mel_spectrogram(wav)
        mels = torch.from_numpy(np.array([mel[s] for s in mel_slices]))
        partial_embeds = self(mels)
        if numpy:
            raw_embed = np.mean(partial_embeds.numpy(), axis=0)
            embed = raw_embed / np.linalg.norm(raw_embed, 2)
        else:
            raw_embed = partial_embeds.mean(dim=0)
            embed = raw_embed / torch.linalg.norm(raw_embed, 2)
        if return_partials:
            return embed, partial_embeds, wav_slices
        return embed

    def embed_speaker(self, wavs: List[np.ndarray], **kwargs):
        """`model.py:109-111`."""
        raw_embed = np.mean([self.embed_utterance(wav, return_partials=False, **kwargs) for wav in wavs], axis=0)
        return raw_embed / np.linalg.norm(raw_embed, 2)

    def embed_utterance_from_file(self, fpath: str, numpy: bool):
        """`model.py:112-117`: load at 16 kHz, trim leading / trailing silence (top_db 20), embed."""
        wav_tgt = audio.load_audio(fpath, sr=sampling_rate)
        wav_tgt = audio.trim_silence(wav_tgt, top_db=20)
        return self.embed_utterance(wav_tgt, numpy=numpy)
