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
        if self.device is not None and self.device.type == "cuda" and self.device.index is not None:
            # the audio front end (mv_audio_*) is called from request threads, whose current HIP device would be 0 on every rank
            N.check(N.lib().mv_audio_set_device(int(self.device.index)))
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
    def partial_table(n_samples: int, rate: float, min_coverage: float) -> np.ndarray:
        """First mel frame of every partial utterance, as one int array (the windows are `partials_n_frames` long, so the whole
        slicing of `model.py:59-84` is this table): a window every round(16000 / rate / 160) frames while the start stays below
        n_frames - 160 + step + 1, at least one window; the last one goes when less than `min_coverage` of its 25 600 samples is
        real signal (and it is not the only one)."""
        hop = sampling_rate * mel_window_step // 1000                      # samples per mel frame
        n_frames = -(-(n_samples + 1) // hop)                              # ceil
        step = int(np.round(sampling_rate / rate / hop))
        starts = np.arange(0, max(1, n_frames - partials_n_frames + step + 1), step, dtype=np.int64)
        covered = (n_samples - starts[-1] * hop) / float(partials_n_frames * hop)
        return starts[:-1] if (covered < min_coverage and starts.size > 1) else starts

    @classmethod
    def compute_partial_slices(cls, n_samples: int, rate, min_coverage):
        """The reference's call form (`model.py:59-84`): (wav slices, mel slices) of the partial utterances."""
        hop = sampling_rate * mel_window_step // 1000
        starts = cls.partial_table(n_samples, rate, min_coverage).tolist()
        return ([slice(a * hop, (a + partials_n_frames) * hop) for a in starts], [slice(a, a + partials_n_frames) for a in starts])

    def embed_utterance(self, wav: np.ndarray, return_partials=False, rate=1.3, min_coverage=0.75, numpy: bool = True):
        """`model.py:86-107`: zero-pad the signal to the end of the last partial, one mel spectrogram, all partials gathered with
        one index table and embedded in ONE call of the HIP LSTM, mean of the partial embeddings, L2-normalised."""
        hop = sampling_rate * mel_window_step // 1000
        n_samples = len(wav)
        starts = self.partial_table(n_samples, rate, min_coverage)
        need = int(starts[-1] + partials_n_frames) * hop
        if need >= len(wav):
            wav = np.concatenate([wav, np.zeros(need - len(wav), dtype=wav.dtype)])
        mel = audio.wav_to_mel_spectrogram(wav)
        frames = starts[:, None] + np.arange(partials_n_frames)[None, :]   # (n_partials, 160) frame indices
        partial_embeds = self(torch.from_numpy(np.ascontiguousarray(mel[frames])))
        mean = partial_embeds.numpy().mean(axis=0) if numpy else partial_embeds.mean(dim=0)
        embed = mean / (np.linalg.norm(mean, 2) if numpy else torch.linalg.norm(mean, 2))
        if return_partials:
            return embed, partial_embeds, self.compute_partial_slices(n_samples, rate, min_coverage)[0]
        return embed

    def embed_speaker(self, wavs: List[np.ndarray], **kwargs):
        """`model.py:109-111`: mean of the utterance embeddings, L2-normalised again."""
        centre = np.stack([self.embed_utterance(w, return_partials=False, **kwargs) for w in wavs]).mean(axis=0)
        return centre / np.linalg.norm(centre, 2)

    def embed_utterance_from_file(self, fpath: str, numpy: bool):
        """`model.py:112-117`: load at 16 kHz, trim leading / trailing silence (top_db 20), embed."""
        wav_tgt = audio.load_audio(fpath, sr=sampling_rate)
        wav_tgt = audio.trim_silence(wav_tgt, top_db=20)
        return self.embed_utterance(wav_tgt, numpy=numpy)
