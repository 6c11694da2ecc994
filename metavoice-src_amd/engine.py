"""Python handle on one `mv_engine` (C ABI in include/metavoice_hip.h): weights, KV caches and the decode-step hipGraph of
the first-stage model live on the GPU inside libmetavoice_hip.so; this class only marshals arguments."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import _native as N


@dataclass
class SamplingParams:
    """`**sampling_kwargs` of the reference's generate() (fam/llm/fast_inference_utils.py:85-120)."""

    guidance_scale: float = 3.0
    temperature: float = 1.0
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    seed: int = 1337

    def to_native(self, dtype: torch.dtype = torch.bfloat16) -> N.mv_sampling:
        # The reference hands these over as 0-d tensors in the model dtype (fam/llm/fast_inference.py:137-139), so
        # 0.95 becomes 0.94921875 on the bf16 path and `1 - guidance_scale` is evaluated in that dtype too.
        def rnd(v):
            return torch.tensor(float(v), dtype=dtype)

        g = rnd(self.guidance_scale)
        s = N.mv_sampling()
        s.guidance_scale = float(g)
        s.one_minus_guidance = float(1 - g)
        s.temperature = float(rnd(self.temperature))
        s.top_p = float(rnd(self.top_p)) if self.top_p is not None else -1.0
        s.top_k = int(self.top_k) if self.top_k is not None else -1
        s.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF
        return s


def _as_float(v) -> float:
    return float(v.item()) if isinstance(v, torch.Tensor) else float(v)


class Stage1Engine:
    """Owns a native engine for `args` (a `fast_model.ModelArgs`)."""

    def __init__(self, args, max_utterances: int = 1, quantisation_mode: Optional[str] = None):
        fmt = {None: N.MV_W_BF16, "int4": N.MV_W_INT4_G128, "int8": N.MV_W_INT8}.get(quantisation_mode, None)
        if fmt is None:
            # fam/llm/fast_inference_utils.py:307-308
            raise Exception(f"Invalid quantisation mode {quantisation_mode}! Must be either 'int4' or 'int8'!")
        self.args = args
        self.max_utterances = max_utterances
        # The persistent decode kernels exist for engines of 1, 2, 4 and (bf16) 8 utterances.  A request for 3 / 5..7 gets the next size
        # up: the spare slots stay idle (mv_begin marks them done, their rows cost a few cache lines), where a native engine of 3 would
        # decode on the per-layer launches and one of 5..7 does not exist.
        slots = max_utterances
        if max_utterances == 3:
            slots = 4
        elif 5 <= max_utterances <= 7 and fmt == N.MV_W_BF16:
            slots = 8
        self.slots = slots
        self.rows = 2 * slots
        cfg = N.mv_config(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, intermediate=args.intermediate_size,
                          vocab=args.vocab_size, block_size=args.block_size, spk_dim=args.speaker_emb_dim,
                          norm_eps=args.norm_eps, max_utterances=slots, weight_format=fmt)
        self._h = C.c_void_p()
        N.check(N.lib().mv_engine_create(C.byref(cfg), C.byref(self._h)))
        self._finalized = False

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            N.lib().mv_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------------------------------
    def load_weight(self, name: str, t: torch.Tensor):
        t = t.detach()
        if t.dtype not in (torch.bfloat16, torch.float32):
            t = t.to(torch.float32)
        t = t.contiguous()
        rows, cols = (1, t.numel()) if t.dim() == 1 else (t.shape[0], t.shape[1])
        dt = N.MV_BF16 if t.dtype == torch.bfloat16 else N.MV_F32
        N.check(N.lib().mv_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), dt, rows, cols))

    def load_state_dict(self, sd):
        """`sd` uses the gpt-fast names (after the key remap of fam/llm/fast_inference_utils.py:250-278)."""
        for k, v in sd.items():
            self.load_weight(k, v)

    def finalize(self):
        N.check(N.lib().mv_finalize(self._h))
        self._finalized = True

    @property
    def model_bytes(self) -> int:
        return int(N.lib().mv_model_bytes(self._h))

    # ---- generation ----------------------------------------------------------------------------------------------
    def begin(self, prompts: Sequence[Union[torch.Tensor, Sequence[int]]], spk_embs: torch.Tensor,
              sampling: Union[SamplingParams, Sequence[SamplingParams]], max_new_tokens: Optional[int] = None,
              end_of_audio_token: int = 2048, prefill_mode: int = 0, dtype: torch.dtype = torch.bfloat16):
        n = len(prompts)
        lens = [int(len(p)) for p in prompts]
        max_len = max(lens)
        buf = np.zeros((n, max_len), dtype=np.int32)
        for i, p in enumerate(prompts):
            buf[i, : lens[i]] = np.asarray(p.cpu() if isinstance(p, torch.Tensor) else p, dtype=np.int32)
        lens_a = np.asarray(lens, dtype=np.int32)
        spk = spk_embs.detach().to("cpu", torch.bfloat16).reshape(n, -1).contiguous().view(torch.int16).numpy().view(np.uint16)
        if isinstance(sampling, SamplingParams):
            sampling = [sampling] * n
        arr = (N.mv_sampling * n)(*[s.to_native(dtype) for s in sampling])
        N.check(N.lib().mv_begin(self._h, n, buf.ctypes.data_as(C.POINTER(C.c_int32)), lens_a.ctypes.data_as(C.POINTER(C.c_int32)),
                                 max_len, spk.ctypes.data_as(C.POINTER(C.c_uint16)), arr,
                                 -1 if max_new_tokens is None else int(max_new_tokens), int(end_of_audio_token), prefill_mode))
        self._n = n

    def run(self, max_steps: int = -1, poll_every: int = 16) -> int:
        steps = C.c_int32(0)
        N.check(N.lib().mv_run(self._h, max_steps, poll_every, C.byref(steps)))
        return steps.value

    @property
    def last_run_ms(self) -> float:
        return float(N.lib().mv_last_run_ms(self._h))

    def tokens(self, utt: int = 0) -> List[int]:
        cap = self.args.block_size
        out = (C.c_int32 * cap)()
        n = C.c_int32(0)
        N.check(N.lib().mv_get_tokens(self._h, utt, out, cap, C.byref(n)))
        return list(out[: n.value])

    def decode_path(self):
        """(path, timeouts): path 2 = persistent kernel drawing the tokens itself, 1 = persistent kernel + sampler launch per step,
        0 = per-layer launches; timeouts = persistent launches abandoned so far (each fell back to the launches, same bits)."""
        t = C.c_int32(0)
        return int(N.lib().mv_decode_path(self._h, C.byref(t))), int(t.value)

    def tune(self, **knobs: int):
        """mv_tune: launch knobs of the decode kernels / prefetcher (DESIGN.md "Tuning"); takes effect at the next begin()."""
        for k, v in knobs.items():
            N.check(N.lib().mv_tune(self._h, k.encode(), int(v)))

    # ---- parity hooks --------------------------------------------------------------------------------------------
    def capture_logits(self, enable: bool = True, max_steps: int = -1):
        N.check(N.lib().mv_set_logits_capture(self._h, int(enable), max_steps))

    def captured_logits(self, first_pos: int, n_pos: int) -> torch.Tensor:
        """fp32 (n_pos, rows, vocab): entry p holds the logits computed AT input position first_pos + p."""
        out = np.empty((n_pos, self.rows, self.args.vocab_size), dtype=np.float32)
        N.check(N.lib().mv_get_captured_logits(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), first_pos, n_pos))
        return torch.from_numpy(out)
