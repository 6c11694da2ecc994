"""Mirror of `fam/llm/fast_inference.py`: the `TTS` class (`:38-195`) on the MI355X engine.

Same constructor keywords and `synthesise(text, spk_ref_path, top_p, guidance_scale, temperature) -> str`.  Differences,
all deliberate:
  * no telemetry (the reference posts the text of every request to PostHog, `:173-193`);
  * no network: `model_name` may be a local directory holding first_stage.pt / second_stage.pt / speaker_encoder.pt; a hub id
    is resolved through huggingface_hub only if that package can reach the hub;
  * `spk_ref_path` may also be a tensor / `.pt` / `.npy` file holding a precomputed (1, 256) speaker embedding (the LSTM
    speaker encoder runs once per voice and is cached on disk by the reference, `fam/llm/inference.py:419-435`; it is outside
    the timed path);
  * the DeepFilterNet enhancer (`:158-163`, third-party Rust) is an optional pluggable post-step, off by default;
  * the vocoder is injected (`decoders.EncodecDecoder(vocoder=...)`) until the HIP EnCodec / MBD stage lands.
"""
from __future__ import annotations

import os
import time
from pathlib import Path
from typing import Callable, Literal, Optional

import numpy as np
import torch

from .adapters import FlattenedInterleavedEncodec2Codebook, TiltedEncodec
from .decoders import EncodecDecoder
from .fast_inference_utils import build_model, main
from .inference import InferenceConfig, Model
from .tokenise import TrainedBPETokeniser
from .utils import check_audio_file, get_default_dtype, get_device, normalize_text


def _resolve_model_dir(model_name: str) -> str:
    if os.path.isdir(model_name):
        return model_name
    from huggingface_hub import snapshot_download  # network; same call as fast_inference.py:71

    return snapshot_download(repo_id=model_name)


class TTS:
    END_OF_AUDIO_TOKEN = 1024

    def __init__(self, model_name: str = "metavoiceio/metavoice-1B-v0.1", *, seed: int = 1337, output_dir: str = "outputs",
                 quantisation_mode: Optional[Literal["int4", "int8"]] = None, first_stage_path: Optional[str] = None,
                 telemetry_origin: Optional[str] = None, vocoder: Optional[Callable] = None, enhancer: Optional[Callable] = None):
        self._dtype = get_default_dtype()
        self._device = get_device()
        self._model_dir = _resolve_model_dir(model_name)
        self.first_stage_adapter = FlattenedInterleavedEncodec2Codebook(end_of_audio_token=self.END_OF_AUDIO_TOKEN)
        self.output_dir = output_dir
        os.makedirs(self.output_dir, exist_ok=True)
        if first_stage_path:
            print(f"Overriding first stage checkpoint via provided model: {first_stage_path}")
        self._first_stage_ckpt = first_stage_path or f"{self._model_dir}/first_stage.pt"
        config_second_stage = InferenceConfig(ckpt_path=f"{self._model_dir}/second_stage.pt", num_samples=1, seed=seed,
                                              device=self._device, dtype=self._dtype, compile=False, init_from="resume",
                                              output_dir=self.output_dir)
        data_adapter_second_stage = TiltedEncodec(end_of_audio_token=self.END_OF_AUDIO_TOKEN)
        self.llm_second_stage = Model(config_second_stage, TrainedBPETokeniser, EncodecDecoder,
                                      data_adapter_fn=data_adapter_second_stage.decode, vocoder=vocoder)
        self.enhancer = enhancer
        self.precision = {"float16": torch.float16, "bfloat16": torch.bfloat16}[self._dtype]
        self.model, self.tokenizer, self.smodel, self.model_size = build_model(
            precision=self.precision, checkpoint_path=Path(self._first_stage_ckpt),
            spk_emb_ckpt_path=Path(f"{self._model_dir}/speaker_encoder.pt"), device=self._device, compile=True,
            compile_prefill=True, quantisation_mode=quantisation_mode)
        self._seed = seed
        self._quantisation_mode = quantisation_mode
        self._model_name = model_name
        self._n_calls = 0

    def _speaker_embedding(self, spk_ref) -> torch.Tensor:
        if isinstance(spk_ref, torch.Tensor):
            return spk_ref.reshape(1, -1)
        p = str(spk_ref)
        if p.endswith(".pt"):
            return torch.load(p, map_location="cpu").reshape(1, -1)
        if p.endswith(".npy"):
            return torch.from_numpy(np.load(p)).reshape(1, -1)
        if self.smodel is None:
            raise FileNotFoundError(f"no speaker encoder available to embed {p}; pass a precomputed (1, 256) embedding instead")
        check_audio_file(p)
        return self.smodel.embed_utterance_from_file(p, numpy=False).unsqueeze(0)

    def synthesise(self, text: str, spk_ref_path, top_p=0.95, guidance_scale=3.0, temperature=1.0) -> str:
        """`fast_inference.py:111-195`: text -> path of the synthesised `.wav`."""
        text = normalize_text(text)
        spk_emb = self._speaker_embedding(spk_ref_path).to(dtype=self.precision)
        start = time.time()
        self._n_calls += 1
        tokens = main(model=self.model, tokenizer=self.tokenizer, model_size=self.model_size, prompt=text, spk_emb=spk_emb,
                      top_p=torch.tensor(top_p, dtype=self.precision), guidance_scale=torch.tensor(guidance_scale, dtype=self.precision),
                      temperature=torch.tensor(temperature, dtype=self.precision), device=self._device,
                      seed=self._seed + self._n_calls)
        _, extracted_audio_ids = self.first_stage_adapter.decode([tokens])
        wav_files = self.llm_second_stage(
            texts=[text], encodec_tokens=[torch.tensor(extracted_audio_ids, dtype=torch.int32).unsqueeze(0)],
            speaker_embs=spk_emb.unsqueeze(0), batch_size=1, guidance_scale=None, top_p=None, top_k=200, temperature=1.0,
            max_new_tokens=None)
        wav_file = wav_files[0]
        if wav_file is None:
            raise Exception("second stage / vocoder failed (see the message above)")
        if self.enhancer is not None:
            self.enhancer(str(wav_file) + ".wav", str(wav_file) + ".wav")
        print(f"\nSaved audio to {wav_file}.wav")
        time_to_synth_s = time.time() - start
        n_frames = len(extracted_audio_ids[0])
        duration_s = n_frames / 75.0  # EnCodec 24 kHz: 75 frames per second
        print(f"\nTotal time to synth (s): {time_to_synth_s}")
        print(f"Real-time factor: {time_to_synth_s / max(duration_s, 1e-9):.2f}")
        return str(wav_file) + ".wav"
