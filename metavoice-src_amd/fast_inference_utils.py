"""Mirror of `fam/llm/fast_inference_utils.py` on the MI355X engine.

Same public names and argument meaning as the reference -- `build_model`, `generate`, `main`, `encode_tokens`,
`prefill`, `decode_n_tokens`, `device_sync` -- but the decode loop, the sampler and the per-token bookkeeping all run
inside libmetavoice_hip.so (one hipGraph replay per token, no per-token host sync).  Nothing here computes on the CPU.
"""
from __future__ import annotations

import time
import warnings
from pathlib import Path
from typing import Dict, Literal, Optional

import torch

from .engine import SamplingParams, _as_float
from .fast_model import ModelArgs, Transformer
from .tokenise import TrainedBPETokeniser


def device_sync(device):
    """fam/llm/fast_inference_utils.py:40-46."""
    if "cuda" in device:
        torch.cuda.synchronize()
    elif "cpu" in device:
        pass
    else:
        print(f"device={device} is not yet suppported")


def remap_checkpoint_keys(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """MetaVoice-1B (nanoGPT-style) tensor names -> gpt-fast names, as `_load_model` does
    (fam/llm/fast_inference_utils.py:246-278), including the `_orig_mod.` prefix torch.compile leaves behind."""
    renames = (
        ("transformer.h.", "layers."),
        (".attn.c_attn.", ".attention.wqkv."),
        (".attn.c_proj.", ".attention.wo."),
        (".mlp.swiglu.w1.", ".feed_forward.swiglu.w1."),
        (".mlp.swiglu.w3.", ".feed_forward.swiglu.w3."),
        (".ln_1.", ".attention_norm."),
        (".ln_2.", ".ffn_norm."),
        (".mlp.c_proj.", ".feed_forward.w2."),
    )
    top = {
        "transformer.wtes.0.weight": "tok_embeddings.weight",
        "transformer.wpe.weight": "pos_embeddings.weight",
        "lm_heads.0.weight": "output.weight",
        "transformer.ln_f.weight": "norm.weight",
    }
    out: Dict[str, torch.Tensor] = {}
    for key, value in state_dict.items():
        if key.startswith("_orig_mod."):
            key = key[len("_orig_mod."):]
        key = top.get(key, key)
        for old, new in renames:
            key = key.replace(old, new)
        out[key] = value
    return out


def _load_model(checkpoint_path, spk_emb_ckpt_path, device, precision,
                quantisation_mode: Optional[Literal["int4", "int8"]] = None, max_utterances: int = 1):
    """fam/llm/fast_inference_utils.py:236-321: checkpoint -> engine (+ tokenizer, speaker encoder handle)."""
    if quantisation_mode not in (None, "int4", "int8"):
        raise Exception(f"Invalid quantisation mode {quantisation_mode}! Must be either 'int4' or 'int8'!")
    checkpoint = torch.load(str(checkpoint_path), mmap=True, weights_only=False, map_location="cpu")
    state_dict = remap_checkpoint_keys(checkpoint["model"])
    margs = checkpoint.get("model_args")
    if margs is not None and "n_embd" in margs:
        args = ModelArgs(n_layer=margs["n_layer"], n_head=margs["n_head"], dim=margs["n_embd"],
                         vocab_size=margs["vocab_sizes"][0], block_size=margs["block_size"])
    else:
        args = ModelArgs.from_name("metavoice-1B")
    if quantisation_mode is not None:
        if quantisation_mode == "int8":
            warnings.warn("int8 quantisation is slower than bf16/fp16 for undebugged reasons! Please set optimisation_mode to `None` or to `int4`.")
        warnings.warn("quantisation will degrade the quality of the audio! Please set optimisation_mode to `None` for best quality.")
    model = Transformer(args, max_utterances=max_utterances, quantisation_mode=quantisation_mode)
    # the reference always casts the checkpoint to bf16 whatever `precision` says (:281)
    model.load_state_dict({k: v.to(torch.bfloat16) for k, v in state_dict.items()}, assign=True)
    tokenizer_info = checkpoint.get("meta", {}).get("tokenizer", {})
    tokenizer = TrainedBPETokeniser(**tokenizer_info)
    smodel = None
    if spk_emb_ckpt_path is not None and Path(spk_emb_ckpt_path).is_file():
        from .speaker_encoder import SpeakerEncoder  # host-side pre-processing, outside the timed path
        smodel = SpeakerEncoder(weights_fpath=spk_emb_ckpt_path, device=device, eval=True, verbose=False)
    return model.eval(), tokenizer, smodel


def build_model(*, precision: torch.dtype, checkpoint_path: Path = Path(""), spk_emb_ckpt_path: Path = Path(""),
                compile_prefill: bool = False, compile: bool = True, device: str = "cuda",
                quantisation_mode: Optional[Literal["int4", "int8"]] = None, max_utterances: int = 1):
    """fam/llm/fast_inference_utils.py:324-392.  `compile` / `compile_prefill` are accepted for call-site compatibility:
    the hipGraph capture in `mv_finalize` plays the role of `torch.compile(mode="max-autotune")` and takes well under a
    second, so there is no 30-90 s warm-up generation."""
    assert Path(checkpoint_path).is_file(), checkpoint_path
    print(f"Using device={device}")
    print("Loading model ...")
    t0 = time.time()
    model, tokenizer, smodel = _load_model(checkpoint_path, spk_emb_ckpt_path, device, precision,
                                           quantisation_mode=quantisation_mode, max_utterances=max_utterances)
    device_sync(device=device)
    print(f"Time to load model: {time.time() - t0:.02f} seconds")
    model.setup_spk_cond_mask()
    t0 = time.perf_counter()
    model.setup_caches(max_batch_size=2 * max_utterances, max_seq_length=model.config.block_size)
    model_size = model.engine.model_bytes
    print(f"Graph capture time: {time.perf_counter() - t0:.2f} seconds")
    return model, tokenizer, smodel, model_size


def _sampling_from_kwargs(*, temperature, guidance_scale, top_p=None, top_k=None, seed: int = 1337) -> SamplingParams:
    return SamplingParams(guidance_scale=_as_float(guidance_scale), temperature=_as_float(temperature),
                          top_p=None if top_p is None else _as_float(top_p), top_k=None if top_k is None else int(top_k),
                          seed=seed)


@torch.no_grad()
def generate(model: Transformer, prompt: torch.Tensor, spk_emb: torch.Tensor, *, max_new_tokens: Optional[int] = None,
             callback=lambda x: x, end_of_audio_token: int = 2048, seed: int = 1337, **sampling_kwargs) -> torch.Tensor:
    """fam/llm/fast_inference_utils.py:182-228: prompt (T,) int32 + spk_emb (1, 256) -> (T + n,) int32, end-of-audio
    token included.  Raises ValueError("Prompt is too long to generate more tokens") like :203-204.

    Differences that are deliberate: the whole loop is enqueued on the GPU (the end-of-audio test happens on the device and
    the host only polls a flag every 16 tokens); `callback` is invoked once per generated token AFTER the run instead of
    between tokens; when the very first sampled token is end-of-audio the reference crashes in `torch.cat([])` (:226),
    here the sequence simply ends with that token.
    """
    engine = model.engine
    sp = _sampling_from_kwargs(seed=seed, **sampling_kwargs)
    engine.begin([prompt], spk_emb, sp, max_new_tokens=max_new_tokens, end_of_audio_token=end_of_audio_token,
                 prefill_mode=1, dtype=model.config.dtype)  # batched (MFMA) prefill, fast_inference_utils.py:123-132
    engine.run()
    seq = engine.tokens(0)
    T = prompt.size(0)
    for t in seq[T:]:
        callback(torch.tensor([t], dtype=torch.int32))
    return torch.tensor(seq, dtype=prompt.dtype, device=prompt.device)


def encode_tokens(tokenizer: TrainedBPETokeniser, text: str, device="cuda") -> torch.Tensor:
    """fam/llm/fast_inference_utils.py:231-233."""
    return torch.tensor(tokenizer.encode(text), dtype=torch.int, device=device)


def main(*, model, tokenizer, model_size, prompt: str, guidance_scale: torch.Tensor, temperature: torch.Tensor,
         spk_emb: torch.Tensor, top_k: Optional[torch.Tensor] = None, top_p: Optional[torch.Tensor] = None,
         device: str = "cuda", seed: int = 1337) -> list:
    """fam/llm/fast_inference_utils.py:395-445: tokenise, generate, print tokens/s and achieved bandwidth."""
    encoded = encode_tokens(tokenizer, prompt, device="cpu")
    prompt_length = encoded.size(0)
    device_sync(device=device)
    t0 = time.perf_counter()
    y = generate(model, encoded, spk_emb, temperature=temperature, top_k=top_k, top_p=top_p, guidance_scale=guidance_scale,
                 seed=seed)
    device_sync(device=device)
    t = time.perf_counter() - t0
    tokens_generated = y.size(0) - prompt_length
    tokens_sec = tokens_generated / t
    print(f"Time for 1st stage LLM inference: {t:.02f} sec total, {tokens_sec:.02f} tokens/sec")
    print(f"Bandwidth achieved: {model_size * tokens_sec / 1e9:.02f} GB/s")
    print(f"Memory used: {torch.cuda.max_memory_reserved() / 1e9:.02f} GB\n")
    return y.tolist()
