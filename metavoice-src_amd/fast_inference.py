"""Mirror of `fam/llm/fast_inference.py`: the `TTS` class (`:38-195`) on the MI355X engine.

Same constructor keywords and `synthesise(text, spk_ref_path, top_p, guidance_scale, temperature) -> str`.  Differences,
all deliberate:
  * no telemetry (the reference posts the text of every request to PostHog, `:173-193`);
  * `spk_ref_path` may be a local path or a public http(s) URI (`get_cached_file`, fetched once) and its embedding is cached on
    disk like the reference's (`get_cached_embedding`, `inference.py`);
  * `model_name` may be a local directory holding first_stage.pt / second_stage.pt / speaker_encoder.pt; a hub id
    is resolved through huggingface_hub only if that package can reach the hub;
  * `spk_ref_path` may also be a tensor / `.pt` / `.npy` file holding a precomputed (1, 256) speaker embedding (the LSTM
    speaker encoder runs once per voice and is cached on disk by the reference, `fam/llm/inference.py:419-435`; it is outside
    the timed path);
  * the DeepFilterNet enhancer (`:158-163`, third-party Rust) is an optional pluggable post-step, off by default;
  * the vocoder (`fam/llm/decoders.py:13`: `MultiBandDiffusion.get_mbd_24khz(bw=6)`, fetched from the hub there) is built from
    `encodec_24khz.pt` (+ `mbd_comp_8.pt`) in the model directory -- `vocoder.py`; a callable can be injected instead;
  * additions with no counterpart in the reference: `synthesise_long` / `synthesise_long_stream` (BASELINE configs[4]: the
    reference stops at 220 characters, `fam/llm/inference.py:535-541`; the stream yields every chunk's waveform as soon as its
    vocoder pass ends) and `synthesise_batch` (several utterances decoded together).
"""
from __future__ import annotations

import os
import time
from pathlib import Path
from typing import Callable, List, Literal, Optional, Sequence

import numpy as np
import torch

from .adapters import FlattenedInterleavedEncodec2Codebook, TiltedEncodec
from .decoders import EncodecDecoder
from .fast_inference_utils import build_model, main
from .inference import InferenceConfig, Model, get_cached_embedding, get_cached_file
from .longform import MAX_CHARS, split_text
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
                 telemetry_origin: Optional[str] = None, vocoder: Optional[Callable] = None, enhancer: Optional[Callable] = None,
                 max_utterances: int = 1):
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
        if vocoder is None:  # decoders.py:13 builds it at import time; a TTS that cannot vocode fails here, not at the first request
            from .vocoder import MultiBandDiffusion

            vocoder = MultiBandDiffusion.get_mbd_24khz(bw=6.0, model_dir=self._model_dir, seed=seed)
        self.vocoder = vocoder
        self.llm_second_stage = Model(config_second_stage, TrainedBPETokeniser, EncodecDecoder,
                                      data_adapter_fn=data_adapter_second_stage.decode, vocoder=vocoder)
        self.enhancer = enhancer
        self.precision = {"float16": torch.float16, "bfloat16": torch.bfloat16}[self._dtype]
        self.model, self.tokenizer, self.smodel, self.model_size = build_model(
            precision=self.precision, checkpoint_path=Path(self._first_stage_ckpt),
            spk_emb_ckpt_path=Path(f"{self._model_dir}/speaker_encoder.pt"), device=self._device, compile=True,
            compile_prefill=True, quantisation_mode=quantisation_mode, max_utterances=max_utterances)
        self.max_utterances = max_utterances
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
        p = get_cached_file(p)  # fast_inference.py:122-127: fetch a URI once, length check, embedding cached per file
        check_audio_file(p)
        return get_cached_embedding(p, self.smodel)

    def check_speaker_reference(self, spk_ref) -> None:
        """Raises what `synthesise` would raise for this speaker reference before any GPU work (serving.py validates requests with
        it on rank 0 before a batch is announced to the other ranks): missing file, unreadable format, shorter than 30 s
        (`fam/llm/utils.py:55-70`).  Precomputed embeddings (.pt / .npy / tensors) only have to exist."""
        if isinstance(spk_ref, torch.Tensor):
            return
        p = str(spk_ref)
        if p.endswith((".pt", ".npy")):
            if not os.path.exists(p):
                raise FileNotFoundError(p)
            return
        if self.smodel is None:
            raise FileNotFoundError(f"no speaker encoder available to embed {p}; pass a precomputed (1, 256) embedding instead")
        check_audio_file(get_cached_file(p))

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

    # ---- additions: several utterances per call, long texts ------------------------------------------------------------
    def _first_stage(self, texts: Sequence[str], spk_embs: torch.Tensor, top_p, guidance_scale, temperature) -> List[List[int]]:
        """Token streams of up to `max_utterances` texts decoded TOGETHER (one weight stream for all of them; per-utterance
        positions, end-of-audio and seeds -- the semantics of the reference's batched slow path, `mixins/causal.py:179-287`)."""
        from .engine import SamplingParams

        engine = self.model.engine
        out: List[List[int]] = []
        for i in range(0, len(texts), self.max_utterances):
            group = list(texts[i:i + self.max_utterances])
            prompts = [self.tokenizer.encode(t) for t in group]
            sps = []
            for _ in group:
                self._n_calls += 1
                sps.append(SamplingParams(guidance_scale=float(guidance_scale), temperature=float(temperature), top_p=float(top_p),
                                          seed=self._seed + self._n_calls))
            engine.begin(prompts, spk_embs[i:i + len(group)], sps, end_of_audio_token=2048, prefill_mode=1, dtype=self.precision)
            engine.run()
            out.extend(engine.tokens(u) for u in range(len(group)))
        return out

    def _expand_tokens(self, text: str, tokens: List[int], emb: torch.Tensor) -> torch.Tensor:
        """First-stage stream of one text -> (1, 1, samples) fp32 waveform: adapter, second stage, vocoder."""
        _, ids = self.first_stage_adapter.decode([tokens])
        return self.llm_second_stage.non_causal_wavs(
            texts=[text], encodec_tokens=[torch.tensor(ids, dtype=torch.int32).unsqueeze(0)], speaker_embs=emb.unsqueeze(0),
            batch_size=1, top_k=200, temperature=1.0)[0]

    def synthesise_wavs(self, texts: Sequence[str], spk_refs, top_p=0.95, guidance_scale=3.0, temperature=1.0,
                        return_exceptions: bool = False) -> list:
        """`synthesise` for a list of texts, waveforms kept in memory: (1, 1, samples) fp32 at 24 kHz each, in input order.
        `spk_refs`: one reference for all texts or one per text.  With `return_exceptions` a text that fails (characters
        `normalize_text` refuses, an unreadable or too short speaker reference, 'wav predicted is shorter than 400ms!') leaves
        its exception in its slot and the other texts of the call are still synthesised -- what a batching server needs: one bad
        request must not take down the requests it happened to be grouped with."""
        n = len(texts)
        if isinstance(spk_refs, (str, os.PathLike, torch.Tensor)):
            spk_refs = [spk_refs] * n
        out: list = [None] * n
        norm: List[Optional[str]] = [None] * n
        embs: List[Optional[torch.Tensor]] = [None] * n
        cache = {}
        for i, (t, r) in enumerate(zip(texts, spk_refs)):
            try:
                norm[i] = normalize_text(t)
                key = r if isinstance(r, (str, os.PathLike)) else id(r)
                if key not in cache:
                    cache[key] = self._speaker_embedding(r).to(dtype=self.precision)
                embs[i] = cache[key]
            except Exception as e:
                if not return_exceptions:
                    raise
                out[i] = e
        good = [i for i in range(n) if out[i] is None]
        if good:
            try:
                streams = self._first_stage([norm[i] for i in good], torch.cat([embs[i] for i in good], dim=0), top_p, guidance_scale, temperature)
            except Exception as e:  # the group decodes together: an engine failure is everybody's
                if not return_exceptions:
                    raise
                for i in good:
                    out[i] = e
                return out
            # adapter + second stage per utterance (a failure is that utterance's), then ONE vocoder call for the group: the diffusion bands
            # of up to four utterances run concurrently (`decoders.decode_wavs`); samples and seeds as with one call per utterance
            expanded = []
            for i, tokens in zip(good, streams):
                try:
                    _, ids = self.first_stage_adapter.decode([tokens])
                    tl = next(iter(self.llm_second_stage._second_stage_token_lists(
                        texts=[norm[i]], encodec_tokens=[torch.tensor(ids, dtype=torch.int32).unsqueeze(0)], speaker_embs=embs[i].unsqueeze(0),
                        batch_size=1, top_k=200, temperature=1.0)))
                    expanded.append((i, tl))
                except Exception as e:
                    if not return_exceptions:
                        raise
                    out[i] = e
            if expanded:
                # decode_wavs keeps a failure (the grid of one utterance, a native error on one vocoder lane, a too short waveform) in the
                # slot of the utterance it belongs to when return_exceptions is set; without it the first failure propagates
                wavs = self.llm_second_stage.decoder.decode_wavs([tl for _, tl in expanded], return_exceptions=return_exceptions)
                for (i, _), w in zip(expanded, wavs):
                    out[i] = w
        return out

    def synthesise_batch(self, texts: Sequence[str], spk_refs, top_p=0.95, guidance_scale=3.0, temperature=1.0,
                         return_exceptions: bool = False) -> list:
        """-> loudness-normalised 16-bit PCM RIFF bytes per text (what `synthesise` writes to disk), in input order; the
        batched-serving entry point (serving.py).  `return_exceptions`: see `synthesise_wavs`."""
        dec = self.llm_second_stage.decoder
        return [w if isinstance(w, Exception) else dec.wav_bytes(w)
                for w in self.synthesise_wavs(texts, spk_refs, top_p, guidance_scale, temperature, return_exceptions=return_exceptions)]

    def synthesise_long_stream(self, text: str, spk_ref_path, top_p=0.95, guidance_scale=3.0, temperature=1.0, max_chars: int = MAX_CHARS,
                               gap_s: float = 0.15):
        """Streaming long-form synthesis (BASELINE configs[4] "streaming decode"): a generator that yields
        `(chunk_index, n_chunks, wav)` -- wav (1, 1, samples) fp32 at 24 kHz, loudness-normalised per chunk, followed by `gap_s` of
        silence except after the last chunk -- as soon as that chunk's vocoder pass ends, so playback (or the chunked HTTP response
        of `serving.py`) starts after ONE chunk instead of after the whole text.  Chunks are independent generations with the same
        speaker embedding on the same resident engine; the stages of a chunk cannot overlap the next chunk's first stage on one
        GPU (the persistent decode kernel needs every CU), so the stream is chunk-sequential."""
        from .loudness import normalize_loudness

        chunks = split_text(normalize_text(text), max_chars)
        if not chunks:
            raise ValueError("empty text")
        emb = self._speaker_embedding(spk_ref_path).to(dtype=self.precision)
        gap = torch.zeros(1, 1, int(gap_s * 24000))
        for i, chunk in enumerate(chunks):
            tokens = self._first_stage([chunk], emb, top_p, guidance_scale, temperature)[0]
            wav = self._expand_tokens(chunk, tokens, emb)
            wav = normalize_loudness(wav.detach().float().cpu().reshape(1, -1), 24000, loudness_headroom_db=14.0,
                                     loudness_compressor=True).reshape(1, 1, -1)
            yield i, len(chunks), (torch.cat([wav, gap], dim=-1) if i + 1 < len(chunks) else wav)

    def synthesise_long(self, text: str, spk_ref_path, top_p=0.95, guidance_scale=3.0, temperature=1.0, max_chars: int = MAX_CHARS,
                        gap_s: float = 0.15) -> str:
        """Long-form synthesis: `text` is split at sentence boundaries into chunks of <= `max_chars` characters (`longform.py`),
        every chunk is an independent generation with the same speaker embedding -- the decode graph captured at load time is
        replayed for all of them -- and the waveforms are joined with `gap_s` of silence, loudness-normalised once and saved."""
        chunks = split_text(normalize_text(text), max_chars)
        if not chunks:
            raise ValueError("empty text")
        start = time.time()
        emb = self._speaker_embedding(spk_ref_path)
        wavs = self.synthesise_wavs(chunks, emb, top_p, guidance_scale, temperature)
        gap = torch.zeros(1, 1, int(gap_s * 24000))
        pieces = []
        for i, w in enumerate(wavs):
            pieces.append(w)
            if i + 1 < len(wavs):
                pieces.append(gap)
        wav = torch.cat(pieces, dim=-1)
        dec = self.llm_second_stage.decoder
        import uuid

        name = dec.output_dir / f"synth_long_{uuid.uuid4()}"
        dec._save_audio(name, wav)
        dur = wav.shape[-1] / 24000.0
        print(f"\nSaved audio to {name}.wav ({len(chunks)} chunks, {dur:.1f} s)")
        print(f"Real-time factor: {(time.time() - start) / max(dur, 1e-9):.2f}")
        return str(name) + ".wav"


def _parse_cli(argv=None):
    """`fast_inference.py:198-199` builds a `TTS` from the command line (`tyro.cli(TTS)`): the same constructor keywords here, plus an
    optional one-shot synthesis (`--text` with `--spk_ref_path`) so that the module is usable as a command."""
    import argparse

    ap = argparse.ArgumentParser(description="MetaVoice-1B text to speech on MI355X")
    ap.add_argument("--model_name", default="metavoiceio/metavoice-1B-v0.1", help="hub id or a local directory with the checkpoints")
    ap.add_argument("--seed", type=int, default=1337)
    ap.add_argument("--output_dir", default="outputs")
    ap.add_argument("--quantisation_mode", choices=["int4", "int8"], default=None)
    ap.add_argument("--first_stage_path", default=None)
    ap.add_argument("--telemetry_origin", default=None, help="accepted for compatibility; nothing is sent anywhere")
    ap.add_argument("--text", default=None, help="synthesise this text once and print the path of the .wav")
    ap.add_argument("--spk_ref_path", default=None, help="speaker reference (path, http(s) URI or a precomputed .pt / .npy embedding)")
    ap.add_argument("--top_p", type=float, default=0.95)
    ap.add_argument("--guidance_scale", type=float, default=3.0)
    ap.add_argument("--temperature", type=float, default=1.0)
    a = ap.parse_args(argv)
    if (a.text is None) != (a.spk_ref_path is None):
        ap.error("--text and --spk_ref_path go together")
    return a


if __name__ == "__main__":
    _a = _parse_cli()
    tts = TTS(model_name=_a.model_name, seed=_a.seed, output_dir=_a.output_dir, quantisation_mode=_a.quantisation_mode,
              first_stage_path=_a.first_stage_path, telemetry_origin=_a.telemetry_origin)
    if _a.text is not None:
        print(tts.synthesise(_a.text, _a.spk_ref_path, top_p=_a.top_p, guidance_scale=_a.guidance_scale, temperature=_a.temperature))
