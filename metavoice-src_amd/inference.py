"""Mirror of the inference-side of `fam/llm/inference.py`: `InferenceConfig` and `Model` for the second (non-causal) stage.

`Model(config, tokenizer_cls, decoder_cls, data_adapter_fn)` loads `second_stage.pt` ({"model", "model_args", "config",
"meta"}, `fam/llm/finetune.py:300-313`) into the HIP second-stage engine and `Model(...)(texts=..., encodec_tokens=..., ...)`
packs the (2, block_size) input grid, samples the 6 remaining codebooks and hands the (8, t) grid to the decoder -- the
contract of `inference.py:53-156,248-377`.  The causal batched slow path (`causal_sample`, `:158-246`) is not part of the
`TTS.synthesise` hot path and is not provided.
"""
from __future__ import annotations

import dataclasses
import hashlib
import os
import tempfile
import urllib.parse
import urllib.request
from typing import Callable, List, Optional, Type

import torch

from .decoders import Decoder
from .model import GPT, GPTConfig
from .tokenise import TrainedBPETokeniser


@dataclasses.dataclass
class InferenceConfig:
    """`fam/llm/inference.py:28-50`."""

    ckpt_path: str
    output_dir: str
    num_samples: int = 10
    seed: int = 1337
    device: str = "cuda"
    dtype: str = "bfloat16"
    compile: bool = False
    init_from: str = "resume"

    def __str__(self):
        return "InferenceConfig:\n" + "\n".join(f"  {f.name}: {getattr(self, f.name)}" for f in dataclasses.fields(self))


def cache_root(kind: str) -> str:
    """`~/.cache/metavoice` (downloaded speaker references) and `~/.cache/fam` (speaker embeddings) are the reference's
    directories (`inference.py:403-404,426-427`), so a machine that ran the reference keeps its cache; `METAVOICE_CACHE_DIR`
    moves both under one root (tests, read-only homes)."""
    root = os.environ.get("METAVOICE_CACHE_DIR")
    d = os.path.join(root, kind) if root else os.path.expanduser(f"~/.cache/{kind}")
    os.makedirs(d, exist_ok=True)
    return d


MAX_REFERENCE_BYTES = 256 << 20  # a speaker reference is a minute or two of audio; anything bigger is not one
FETCH_TIMEOUT_S = 60.0


def get_cached_file(file_or_uri: str) -> str:
    """`fam/llm/inference.py:392-416`: an http(s) URI is fetched once into `~/.cache/metavoice/audio_<md5(uri)><ext>` and that
    path is returned; anything else must be an existing local file (FileNotFoundError otherwise).  The reference shells out
    (`curl -o <path> <uri>` through a shell); here the fetch is urllib with a time-out and a size cap, written to a temporary
    name and renamed, so that an interrupted or concurrent fetch never leaves half a file under the cached name and the URI
    never reaches a command line."""
    file_or_uri = str(file_or_uri)
    if not file_or_uri.lower().startswith(("http://", "https://")):   # a local file may well be called http_take3.wav
        if os.path.exists(file_or_uri):
            return file_or_uri
        raise FileNotFoundError(f"File {file_or_uri} not found!")
    parts = urllib.parse.urlsplit(file_or_uri)
    if parts.scheme not in ("http", "https") or not parts.netloc:
        raise FileNotFoundError(f"File {file_or_uri} not found!")
    ext = os.path.splitext(parts.path)[1]
    if len(ext) > 8 or not ext[1:].isalnum():
        ext = ""
    d = cache_root("metavoice")
    cache_path = os.path.join(d, "audio_" + hashlib.md5(file_or_uri.encode("utf-8")).hexdigest() + ext)
    if os.path.exists(cache_path):
        return cache_path
    fd, tmp = tempfile.mkstemp(dir=d, prefix=".fetch_")
    try:
        with os.fdopen(fd, "wb") as out, urllib.request.urlopen(file_or_uri, timeout=FETCH_TIMEOUT_S) as src:
            total = 0
            while True:
                block = src.read(1 << 20)
                if not block:
                    break
                total += len(block)
                if total > MAX_REFERENCE_BYTES:
                    raise ValueError(f"{file_or_uri}: more than {MAX_REFERENCE_BYTES >> 20} MiB, not a speaker reference")
                out.write(block)
        os.replace(tmp, cache_path)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
    return cache_path


UPLOAD_PREFIX = "mv_upload_"   # == serving.UPLOAD_PREFIX (serving imports this module, not the other way round)


def get_cached_embedding(local_file_path: str, spkemb_model) -> torch.Tensor:
    """`fam/llm/inference.py:419-435`: the (1, 256) speaker embedding of an audio file, computed once and kept in
    `~/.cache/fam/embedding_<md5(path)>.pt` (the reference's name, so its cache stays valid here).  One difference: a cache entry
    older than the audio file is recomputed -- the reference keys on the path alone and would keep returning the embedding of
    the file's previous contents."""
    local_file_path = str(local_file_path)
    if not os.path.exists(local_file_path):
        raise FileNotFoundError(f"File {local_file_path} not found!")
    # an UPLOAD lives under a fresh temporary name that is never seen again: caching its embedding by path would leave one file per request
    # in ~/.cache/fam for ever (the reference does exactly that under its server).  Only the files serving.py itself stored
    # (`serving.UPLOAD_PREFIX` under the system temp directory) are exempt -- a user's own reference kept under /tmp is cached like any other
    tmp_root = os.path.realpath(tempfile.gettempdir()) + os.sep
    if os.path.realpath(local_file_path).startswith(tmp_root) and os.path.basename(local_file_path).startswith(UPLOAD_PREFIX):
        return spkemb_model.embed_utterance_from_file(local_file_path, numpy=False).unsqueeze(0).detach().to("cpu")
    d = cache_root("fam")
    cache_path = os.path.join(d, "embedding_" + hashlib.md5(local_file_path.encode("utf-8")).hexdigest() + ".pt")
    if os.path.exists(cache_path) and os.path.getmtime(cache_path) >= os.path.getmtime(local_file_path):
        try:
            emb = torch.load(cache_path, map_location="cpu")
            if isinstance(emb, torch.Tensor) and emb.numel() and emb.dim() == 2 and emb.shape[0] == 1:
                return emb
        except Exception:
            pass  # unreadable entry (interrupted writer of an older version): recompute below
    emb = spkemb_model.embed_utterance_from_file(local_file_path, numpy=False).unsqueeze(0).detach().to("cpu")  # (b=1, c)
    fd, tmp = tempfile.mkstemp(dir=d, prefix=".emb_")
    try:
        with os.fdopen(fd, "wb") as out:
            torch.save(emb, out)
        os.replace(tmp, cache_path)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
    return emb


def pack_second_stage_input(text_tokens: List[int], codebooks: List[List[int]], ctx_window: int, pad_token: int = 1024) -> List[List[int]]:
    """`inference.py:283-306`: row 0 = text ++ codebook 0 ++ [pad], row 1 = [pad]*len(text) ++ codebook 1 ++ [pad], each
    right-padded with `pad_token` or TRUNCATED to `ctx_window`."""
    rows = [list(text_tokens) + list(codebooks[0]) + [pad_token],
            [pad_token] * len(text_tokens) + list(codebooks[1]) + [pad_token]]
    assert len(rows[0]) == len(rows[1])
    return [(r + [pad_token] * (ctx_window - len(r)))[:ctx_window] for r in rows]


class Model:
    def __init__(self, config: InferenceConfig, tokenizer_cls: Type[TrainedBPETokeniser], decoder_cls: Type[Decoder],
                 data_adapter_fn: Callable, use_kv_cache=None, checkpoint: Optional[dict] = None, **decoder_kwargs):
        self._encodec_codes_pad_token = 1024
        self._num_encodec_codebooks = 8
        self.config = config
        if use_kv_cache is not None:
            raise Exception("kv_cache not supported for non-causal models!")  # inference.py:147-149
        if checkpoint is None:
            checkpoint = torch.load(config.ckpt_path, map_location="cpu", weights_only=False)
        self.checkpoint_config = checkpoint.get("config", {})
        if self.checkpoint_config.get("causal", True):
            raise NotImplementedError("Model wraps the non-causal second stage; the causal first stage runs through fast_inference_utils")
        self.meta = checkpoint["meta"]
        self.vocab_sizes = checkpoint["model_args"]["vocab_sizes"]
        self.speaker_cond = bool(self.meta.get("speaker_cond"))
        model_args = checkpoint["model_args"]
        self._encodec_ctx_window = model_args["block_size"]
        self.model = GPT(GPTConfig(**model_args), speaker_emb_dim=self.meta["speaker_emb_size"] if self.speaker_cond else None)
        self.model.load_state_dict(checkpoint["model"])
        self.tokenizer = tokenizer_cls(**self.meta["tokenizer"])
        self.decoder = decoder_cls(tokeniser_decode_fn=self.tokenizer.decode, output_dir=config.output_dir,
                                   data_adapter_fn=data_adapter_fn, **decoder_kwargs)
        self._calls = 0

    def _second_stage_token_lists(self, *, texts: List[str], encodec_tokens: List[torch.Tensor], batch_size: int,
                                  top_k: Optional[int], temperature: Optional[float], speaker_embs: Optional[torch.Tensor]):
        """Packing + sampling of `inference.py:264-329`: yields one (8 + ..., t) token list per (sample, text)."""
        if speaker_embs is not None:
            assert len(texts) == len(speaker_embs)
        grids = []
        for text, enc in zip(texts, encodec_tokens):
            assert enc.shape[0] == 1
            codebooks = enc[0].tolist()
            assert 1 <= len(codebooks) <= self._num_encodec_codebooks
            grids.append(pack_second_stage_input(self.tokenizer.encode(text), codebooks, self._encodec_ctx_window,
                                                 self._encodec_codes_pad_token))
        in_x = torch.tensor(grids, dtype=torch.long)
        if not self.speaker_cond:
            speaker_embs = None
        for _ in range(self.config.num_samples):
            self._calls += 1
            y = self.model.generate(in_x, None, temperature=temperature, top_k=top_k, top_p=None, speaker_embs=speaker_embs,
                                    batch_size=batch_size, guidance_scale=None, seed=self.config.seed + 7919 * self._calls)
            for tokens in torch.cat([in_x, y], dim=1):
                yield tokens.tolist()

    def non_causal_sample(self, *, texts: List[str], encodec_tokens: List[torch.Tensor], batch_size: int, top_k: Optional[int],
                          temperature: Optional[float], speaker_embs: Optional[torch.Tensor] = None) -> list:
        """`inference.py:248-338`: returns one decoder result (path of the saved audio) per text, `None` where decoding failed."""
        to_return = []
        for tokens in self._second_stage_token_lists(texts=texts, encodec_tokens=encodec_tokens, batch_size=batch_size, top_k=top_k,
                                                     temperature=temperature, speaker_embs=speaker_embs):
            try:
                to_return.append(self.decoder.decode(tokens=tokens, causal=False))
            except Exception as e:  # the reference swallows vocoder failures into a None entry (:331-336)
                print("failed to run MBD.")
                print(f"reason: {str(e)}")
                to_return.append(None)
        return to_return

    def non_causal_wavs(self, *, texts: List[str], encodec_tokens: List[torch.Tensor], batch_size: int, top_k: Optional[int],
                        temperature: Optional[float], speaker_embs: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        """Same sampling as `non_causal_sample`, but the (1, 1, samples) waveforms stay in memory (long-form concatenation,
        batched serving: SURVEY.md 8f row 4); errors propagate instead of turning into `None`."""
        return [self.decoder.decode_wav(tokens) for tokens in self._second_stage_token_lists(
            texts=texts, encodec_tokens=encodec_tokens, batch_size=batch_size, top_k=top_k, temperature=temperature,
            speaker_embs=speaker_embs)]

    def __call__(self, *, texts: List[str], batch_size: int, max_new_tokens: Optional[int], top_k: Optional[int],
                 top_p: Optional[float], temperature: Optional[float], encodec_tokens: Optional[List[torch.Tensor]] = None,
                 speaker_embs: Optional[torch.Tensor] = None, guidance_scale: Optional[float] = None):
        """`inference.py:340-377`, non-causal branch."""
        assert encodec_tokens is not None
        assert guidance_scale is None
        assert max_new_tokens is None
        assert top_p is None
        return self.non_causal_sample(texts=texts, encodec_tokens=encodec_tokens, batch_size=batch_size, speaker_embs=speaker_embs,
                                      top_k=top_k, temperature=temperature)
