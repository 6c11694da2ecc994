#!/usr/bin/env python3
"""Generate the golden vectors in this directory by running the REFERENCE code itself.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference (`fam.llm.fast_model`, `fam.llm.fast_inference_utils`, `fam.llm.model`,
`fam.llm.fast_quantize`, `fam.llm.adapters`, `fam.llm.utils`) is imported straight from
/root/reference with `librosa` / `tiktoken` stubbed in `sys.modules` (they are imported at module top
but unused by the model code) and executed on CPU with this repo's seeded synthetic checkpoints
(`metavoice-src_amd/synthetic.py`).  Nothing from the reference is copied: only its OUTPUTS are stored.
`tests/test_oracle_golden.py` then checks `oracle/` against these files on any machine.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("MV_REFERENCE", "/root/reference")

for m in ("librosa", "tiktoken"):
    sys.modules.setdefault(m, types.ModuleType(m))
sys.path.insert(0, REFERENCE)
sys.path.insert(0, ROOT)

from fam.llm import fast_inference_utils as ref_fiu  # noqa: E402
from fam.llm import fast_quantize as ref_fq  # noqa: E402
from fam.llm.adapters import FlattenedInterleavedEncodec2Codebook, TiltedEncodec  # noqa: E402
from fam.llm.fast_model import ModelArgs, Transformer  # noqa: E402
from fam.llm.model import GPT, GPTConfig  # noqa: E402
from fam.llm.utils import normalize_text as ref_normalize_text  # noqa: E402

synthetic = importlib.import_module("metavoice-src_amd.synthetic")
from oracle import adapters as o_adapters  # noqa: E402
from oracle import quantize as o_quant  # noqa: E402
from oracle import stage1 as o_s1  # noqa: E402
from oracle import stage2 as o_s2  # noqa: E402

torch.set_grad_enabled(False)


def build_ref_stage1(cfg_kwargs, seed, dtype):
    """Reference `Transformer` holding the synthetic checkpoint (after the reference-style key remap)."""
    args = ModelArgs(n_layer=cfg_kwargs["n_layer"], n_head=cfg_kwargs["n_head"], dim=cfg_kwargs["dim"],
                     vocab_size=cfg_kwargs["vocab_size"], block_size=cfg_kwargs["block_size"])
    model = Transformer(args)
    sd = synthetic.stage1_state_dict(seed=seed, **cfg_kwargs)
    sd = o_s1.remap_checkpoint_keys(sd)
    missing = model.load_state_dict({k: v.to(dtype) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model = model.to(dtype).eval()
    model.config.dtype = dtype  # ModelArgs picks fp16 when no GPU is present (utils.py:83-84)
    model.setup_spk_cond_mask()
    model.setup_caches(max_batch_size=2, max_seq_length=args.block_size)
    return model, sd


def stage1_case(name, cfg_kwargs, seed, T, n_new, with_bf16):
    model, sd = build_ref_stage1(cfg_kwargs, seed, torch.float32)
    prompt = synthetic.prompt_tokens(T, seed=seed)
    spk = synthetic.speaker_embedding(seed)
    # (1) the reference's own generate(), greedy expressed as top_k=1
    torch.manual_seed(0)
    seq = ref_fiu.generate(model, prompt, spk, max_new_tokens=n_new, temperature=torch.tensor(1.0), top_k=1,
                           top_p=None, guidance_scale=torch.tensor(3.0), end_of_audio_token=9999)
    seq = seq.tolist()
    # (2) teacher-forced logits along that stream, fp32
    def forced_logits(m, dtype):
        m.setup_caches(2, cfg_kwargs["block_size"])
        for b in m.layers:  # fresh caches
            b.attention.kv_cache.k_cache.zero_()
            b.attention.kv_cache.v_cache.zero_()
        out = []
        x = prompt.view(1, -1).repeat(2, 1)
        lg = m(x, spk.to(dtype), torch.arange(T))
        out.append(lg[:, -1].float())
        for i in range(n_new - 1):
            tok = seq[T + i]
            lg = m(torch.full((2, 1), tok, dtype=torch.int32), spk.to(dtype), torch.tensor([T + i]))
            out.append(lg[:, -1].float())
        return torch.stack(out)
    logits32 = forced_logits(model, torch.float32)
    out = dict(cfg=json.dumps(cfg_kwargs), seed=seed, T=T, n_new=n_new, seq=np.array(seq, dtype=np.int32),
               logits_fp32=logits32.numpy())
    # cross-check the oracle right away
    cfg = o_s1.Stage1Config(**cfg_kwargs)
    orc = o_s1.Stage1Oracle({k: v.float() for k, v in sd.items()}, cfg, rounding="fp32", max_seq_length=cfg.block_size)
    oseq, ologits = o_s1.generate(orc, prompt, spk, max_new_tokens=n_new, end_of_audio_token=9999, top_k=1,
                                  noise=False, return_logits=True)
    print(f"[{name}] fp32: seq equal={oseq == seq}  max|dlogit|={float((ologits - logits32).abs().max()):.3e}")
    if with_bf16:
        mb, _ = build_ref_stage1(cfg_kwargs, seed, torch.bfloat16)
        logits16 = forced_logits(mb, torch.bfloat16)
        out["logits_bf16"] = logits16.numpy()
        ob = o_s1.Stage1Oracle({k: v.float() for k, v in sd.items()}, cfg, rounding="bf16", max_seq_length=cfg.block_size)
        ol = []
        x = prompt.view(1, -1).repeat(2, 1)
        ol.append(ob.forward(x, spk, torch.arange(T))[:, -1])
        for i in range(n_new - 1):
            ol.append(ob.forward(torch.full((2, 1), seq[T + i]), spk, torch.tensor([T + i]))[:, -1])
        ol = torch.stack(ol)
        print(f"[{name}] bf16-points oracle vs reference bf16 eager: max|d|={float((ol - logits16).abs().max()):.3e} "
              f"(vs fp32 ref: oracle {float((ol - logits32).abs().max()):.3e}, ref-bf16 {float((logits16 - logits32).abs().max()):.3e})")
    np.savez_compressed(os.path.join(HERE, f"stage1_{name}.npz"), **out)


FULL = dict(n_layer=24, n_head=16, dim=2048, vocab_size=2562, block_size=2048)


def _quantise_like_the_reference(model, mode):
    """Puts `model` (reference `Transformer`, bf16 weights) into the state the reference's `_load_model` reaches for
    `quantisation_mode` (fast_inference_utils.py:283-308), as far as that is possible on CPU:
      int8 -- the reference's own `WeightOnlyInt8QuantHandler` (create_quantized_state_dict + convert_for_runtime);
      int4 -- `aten::_convert_weight_to_int4pack` has no CPU form (SURVEY 8c gotcha 6), so every Linear the reference would
              replace (`out_features % 8 == 0`, fast_quantize.py:225) gets the weight the int4 kernel multiplies by:
              `group_dequantize_tensor_from_qparams(group_quantize_tensor_from_qparams(w))`, all reference functions."""
    if mode == "int8":
        h = ref_fq.WeightOnlyInt8QuantHandler(model)
        qsd = h.create_quantized_state_dict()
        model = h.convert_for_runtime()
        model.load_state_dict(qsd, assign=True)
        return model
    assert mode == "int4"
    for mod in model.modules():
        if isinstance(mod, torch.nn.Linear) and mod.out_features % 8 == 0 and ref_fq._check_linear_int4_k(mod.in_features, 128, 8):
            w = mod.weight.data.to(torch.bfloat16)
            scales, zeros = ref_fq.get_group_qparams(w, 4, 128)
            q = ref_fq.group_quantize_tensor_from_qparams(w, scales, zeros, 4, 128)
            mod.weight.data = ref_fq.group_dequantize_tensor_from_qparams(q, scales, zeros, 4, 128).to(torch.bfloat16)
    return model


def stage1_full_case(T=128, n_new=64, seed=0):
    """BASELINE.json's own configuration: 24 layers x 2048, the bench prompt (T = 128, seed 0), greedy through the reference's
    `generate()`, logits of every step recorded from the forward calls `generate` itself makes (fp32 compute on the
    bf16-rounded synthetic weights).  Three weight formats: bf16, int4 (group 128), int8 (per channel)."""
    out = dict(cfg=json.dumps(FULL), seed=seed, T=T, n_new=n_new)
    prompt = synthetic.prompt_tokens(T, seed=seed)
    spk = synthetic.speaker_embedding(seed)
    for mode in ("bf16", "int4", "int8"):
        import time
        t0 = time.time()
        model, _ = build_ref_stage1(FULL, seed, torch.bfloat16)
        if mode != "bf16":
            model = _quantise_like_the_reference(model, mode)
        model = model.to(torch.float32).eval()
        model.config.dtype = torch.float32
        model.setup_spk_cond_mask()
        model.setup_caches(max_batch_size=2, max_seq_length=FULL["block_size"])
        cap = []
        hook = model.register_forward_hook(lambda m, i, o: cap.append(o[:, -1].float().clone()))
        torch.manual_seed(0)
        seq = ref_fiu.generate(model, prompt, spk, max_new_tokens=n_new, temperature=torch.tensor(1.0), top_k=1, top_p=None,
                               guidance_scale=torch.tensor(3.0), end_of_audio_token=9999).tolist()
        hook.remove()
        logits = torch.stack(cap)  # (n_new, 2, V): entry i = logits at input position T - 1 + i
        assert logits.shape[0] == n_new and len(seq) == T + n_new
        mix = 3.0 * logits[:, 0] + (1 - 3.0) * logits[:, 1]
        assert mix.argmax(-1).tolist() == seq[T:]
        top2 = mix.topk(2, dim=-1).values
        print(f"[full24 {mode}] {time.time() - t0:.0f} s; raw logit std {float(logits.std()):.3f}, mix std {float(mix.std()):.3f}, "
              f"top-2 margin min {float((top2[:, 0] - top2[:, 1]).min()):.4f} median {float((top2[:, 0] - top2[:, 1]).median()):.3f}")
        out[f"seq_{mode}"] = np.array(seq, dtype=np.int32)
        out[f"logits_{mode}"] = logits.numpy() if mode == "bf16" else logits.numpy().astype(np.float16)
        del model
    out.update(stage1_full_noise_floor(out["seq_bf16"].tolist(), torch.from_numpy(out["logits_bf16"]), T, n_new, seed))
    np.savez_compressed(os.path.join(HERE, "stage1_full24.npz"), **out)


def stage1_full_noise_floor(seq, logits_fp32, T, n_new, seed):
    """What bf16 costs at this size, measured on the reference itself: its `Transformer` in bf16 end to end (eager, CPU),
    teacher-forced along the fp32 greedy stream, against its own fp32 logits.  Stored per step (max over both rows and the
    vocabulary; raw logits and the 3*cond - 2*uncond mix): the tolerance of tests/test_fullsize_golden_gpu.py derives from it."""
    model, _ = build_ref_stage1(FULL, seed, torch.bfloat16)
    prompt = synthetic.prompt_tokens(T, seed=seed)
    spk = synthetic.speaker_embedding(seed).to(torch.bfloat16)
    got = [model(prompt.view(1, -1).repeat(2, 1), spk, torch.arange(T))[:, -1].float()]
    for i in range(n_new - 1):
        got.append(model(torch.full((2, 1), seq[T + i], dtype=torch.int32), spk, torch.tensor([T + i]))[:, -1].float())
    lb = torch.stack(got)
    raw = (lb - logits_fp32).abs().amax(dim=(1, 2))
    mixed = ((3.0 * lb[:, 0] - 2.0 * lb[:, 1]) - (3.0 * logits_fp32[:, 0] - 2.0 * logits_fp32[:, 1])).abs().amax(dim=1)
    print(f"[full24] the reference's own bf16 vs fp32: raw max {float(raw.max()):.4f}, mix max {float(mixed.max()):.4f}")
    return dict(noise_raw_refbf16=raw.numpy(), noise_mix_refbf16=mixed.numpy())


def _ref_bf16_along(model16, prompt, spk16, seq, T, n_new):
    """The reference `Transformer` in bf16 end to end (eager, CPU), teacher-forced along `seq`: logits (n_new, 2, V) as fp32."""
    got = [model16(prompt.view(1, -1).repeat(2, 1), spk16, torch.arange(T))[:, -1].float()]
    for i in range(n_new - 1):
        got.append(model16(torch.full((2, 1), seq[T + i], dtype=torch.int32), spk16, torch.tensor([T + i]))[:, -1].float())
    return torch.stack(got)


def _ref_bf16_choice(lb):
    """What the reference picks when it runs in bf16: the CFG mix exactly as `sample()` evaluates it on bf16 tensors with bf16
    scalars (fast_inference_utils.py:113-117; every product and the sum rounded to bf16), its argmax (lowest index among equal
    maxima: `top_k=1` keeps every tied maximum and the Exp(1) race picks one of them, SURVEY 8c gotcha 8), the number of tied
    maxima and the top-2 margin of that bf16 mix."""
    lb16 = lb.to(torch.bfloat16)
    g = torch.tensor(3.0, dtype=torch.bfloat16)
    mixed = (g * lb16[:, 0] + (1 - g) * lb16[:, 1]).float()  # (n, V) bf16 values
    top2 = mixed.topk(2, dim=-1).values
    mx = mixed.amax(dim=-1, keepdim=True)
    first = (mixed == mx).float().argmax(dim=-1)
    return first.to(torch.int32).numpy(), (mixed == mx).sum(dim=-1).to(torch.int32).numpy(), (top2[:, 0] - top2[:, 1]).numpy()


def stage1_long_case(T, n_new=32, seed=0, modes=("bf16", "int4", "int8")):
    """Long-context goldens at BASELINE's own size: a context of T tokens (the bench's 128-token text prompt +
    T - 128 flattened audio tokens, `synthetic.long_prompt_tokens`), n_new greedy tokens through the reference's `generate()`
    (fp32 compute on the bf16-rounded / quantised weights), the logits of every forward call it makes -- and, along the same
    stream, the reference evaluated in bf16 end to end: what IT picks (argmax of its bf16 CFG mix, ties, margin) and how far its
    logits are from its own fp32 ones (the noise floor the GPU test's tolerance derives from).  T = 1100 is the bench's mean
    context (8-range split attention); T = 2016 ends at the last position the reference can generate (sequence index 2047)."""
    import time
    out = dict(cfg=json.dumps(FULL), seed=seed, T=T, n_new=n_new, n_text=128)
    prompt = synthetic.long_prompt_tokens(T, seed=seed)
    spk = synthetic.speaker_embedding(seed)
    for mode in modes:
        t0 = time.time()
        model, _ = build_ref_stage1(FULL, seed, torch.bfloat16)
        if mode != "bf16":
            model = _quantise_like_the_reference(model, mode)
        model16 = model  # bf16 end to end: the state the reference's `_load_model` reaches (fast_inference_utils.py:283-321)
        import copy
        model = copy.deepcopy(model16).to(torch.float32).eval()
        model.config.dtype = torch.float32
        model.setup_spk_cond_mask()
        model.setup_caches(max_batch_size=2, max_seq_length=FULL["block_size"])
        cap = []
        hook = model.register_forward_hook(lambda m, i, o: cap.append(o[:, -1].float().clone()))
        torch.manual_seed(0)
        seq = ref_fiu.generate(model, prompt, spk, max_new_tokens=n_new, temperature=torch.tensor(1.0), top_k=1, top_p=None,
                               guidance_scale=torch.tensor(3.0), end_of_audio_token=9999).tolist()
        hook.remove()
        del model
        logits = torch.stack(cap)
        assert logits.shape[0] == n_new and len(seq) == T + n_new
        mix = 3.0 * logits[:, 0] + (1 - 3.0) * logits[:, 1]
        assert mix.argmax(-1).tolist() == seq[T:]
        top2 = mix.topk(2, dim=-1).values
        t1 = time.time()
        # the reference in bf16 along that stream
        model16.config.dtype = torch.bfloat16
        model16.setup_spk_cond_mask()
        model16.setup_caches(max_batch_size=2, max_seq_length=FULL["block_size"])
        lb = _ref_bf16_along(model16, prompt, spk.to(torch.bfloat16), seq, T, n_new)
        del model16
        raw = (lb - logits).abs().amax(dim=(1, 2))
        pick, ties, margin16 = _ref_bf16_choice(lb)
        mix16 = 3.0 * lb[:, 0] - 2.0 * lb[:, 1]
        mixed = (mix16 - mix).abs().amax(dim=1)
        agree = int((torch.from_numpy(pick).long() == torch.tensor(seq[T:])).sum())
        print(f"[long24 T={T} {mode}] fp32 {t1 - t0:.0f} s + bf16 {time.time() - t1:.0f} s; raw logit std {float(logits.std()):.3f}, "
              f"mix std {float(mix.std()):.3f}, fp32 top-2 margin min {float((top2[:, 0] - top2[:, 1]).min()):.4f} median "
              f"{float((top2[:, 0] - top2[:, 1]).median()):.3f}; reference-bf16 vs its fp32: raw max {float(raw.max()):.4f}, mix max "
              f"{float(mixed.max()):.4f}, picks the fp32 token at {agree}/{n_new} steps ({int((ties > 1).sum())} steps with tied maxima)",
              flush=True)
        out[f"seq_{mode}"] = np.array(seq, dtype=np.int32)
        out[f"logits_{mode}"] = logits.numpy() if mode == "bf16" else logits.numpy().astype(np.float16)
        out[f"noise_raw_refbf16_{mode}"] = raw.numpy()
        out[f"noise_mix_refbf16_{mode}"] = mixed.numpy()
        out[f"pick_refbf16_{mode}"] = pick
        out[f"ties_refbf16_{mode}"] = ties
        out[f"margin_refbf16_{mode}"] = margin16
    np.savez_compressed(os.path.join(HERE, f"stage1_long24_T{T}.npz"), **out)


def stage1_full_refbf16_choice(seed=0):
    """For the T = 128 case of `stage1_full24.npz`: what the reference evaluated in bf16 picks along the stored fp32 stream
    (bf16 weights only; the int4 / int8 streams of that file keep their fp32 logits as the only comparator)."""
    g = np.load(os.path.join(HERE, "stage1_full24.npz"))
    T, n_new, seq = int(g["T"]), int(g["n_new"]), g["seq_bf16"].tolist()
    model, _ = build_ref_stage1(FULL, seed, torch.bfloat16)
    lb = _ref_bf16_along(model, synthetic.prompt_tokens(T, seed=seed), synthetic.speaker_embedding(seed).to(torch.bfloat16), seq, T, n_new)
    pick, ties, margin16 = _ref_bf16_choice(lb)
    raw = (lb - torch.from_numpy(g["logits_bf16"])).abs().amax(dim=(1, 2))
    agree = int((torch.from_numpy(pick).long() == torch.tensor(seq[T:])).sum())
    print(f"[full24 refbf16] raw max {float(raw.max()):.4f} (stored floor {float(g['noise_raw_refbf16'].max()):.4f}); "
          f"reference-bf16 picks the fp32 token at {agree}/{n_new} steps, {int((ties > 1).sum())} steps with tied maxima")
    np.savez_compressed(os.path.join(HERE, "stage1_full24_refbf16.npz"), T=T, n_new=n_new, pick_refbf16_bf16=pick,
                        ties_refbf16_bf16=ties, margin_refbf16_bf16=margin16)


def slow_vs_fast_selftest(cfg_kwargs, seed):
    """the nanoGPT-style `GPT` and the gpt-fast `Transformer` are two independent reference
    implementations of stage 1; they must agree on the same checkpoint."""
    model, _ = build_ref_stage1(cfg_kwargs, seed, torch.float32)
    gconf = GPTConfig(block_size=cfg_kwargs["block_size"], vocab_sizes=[cfg_kwargs["vocab_size"]], n_layer=cfg_kwargs["n_layer"],
                      n_head=cfg_kwargs["n_head"], n_embd=cfg_kwargs["dim"], bias=False, causal=True, norm_type="rmsnorm",
                      rmsnorm_eps=1e-5, nonlinearity_type="swiglu", swiglu_multiple_of=256)
    slow = GPT(gconf, speaker_emb_dim=256).eval()
    sd = synthetic.stage1_state_dict(seed=seed, **cfg_kwargs)
    slow.load_state_dict({k: v.float() for k, v in sd.items()})
    prompt = synthetic.prompt_tokens(10, seed=seed)
    spk = synthetic.speaker_embedding(seed)
    fast = model(prompt.view(1, -1).repeat(2, 1), spk, torch.arange(10))[:, -1]
    lg, _ = slow(prompt.long().view(1, 1, -1).repeat(2, 1, 1), speaker_embs=[spk[0], None])
    d = float((lg[0][:, -1] - fast).abs().max())
    print(f"[selftest] slow GPT vs fast Transformer max|d|={d:.3e}")
    assert d < 1e-4


def sampling_case():
    g = torch.Generator().manual_seed(7)
    V = 2562
    logits = torch.randn(6, V, generator=g) * 3.0
    logits[1, 100] = logits[1].max()  # an exact tie at the top
    combos = [dict(temperature=1.0, top_p=0.95, top_k=None), dict(temperature=0.7, top_p=0.9, top_k=None),
              dict(temperature=1.0, top_p=None, top_k=50), dict(temperature=1.3, top_p=0.95, top_k=200),
              dict(temperature=0.0, top_p=None, top_k=None), dict(temperature=1.0, top_p=None, top_k=1),
              # the GPU path passes bf16 scalars: 0.95 -> 0.94921875 (SURVEY 8c gotcha 9)
              dict(temperature=1.0, top_p=float(torch.tensor(0.95).bfloat16()), top_k=None)]
    probs = np.zeros((len(combos), logits.shape[0], V), dtype=np.float32)
    worst = 0.0
    for ci, c in enumerate(combos):
        for li in range(logits.shape[0]):
            p = ref_fiu.logits_to_probs(logits[li].clone(), temperature=torch.tensor(c["temperature"]),
                                        top_p=None if c["top_p"] is None else torch.tensor(c["top_p"]), top_k=c["top_k"])
            probs[ci, li] = p.numpy()
            o = o_s1.logits_to_probs(logits[li].clone(), c["temperature"], top_p=c["top_p"], top_k=c["top_k"])
            worst = max(worst, float((o - p).abs().max()))
    # CFG mix through the reference's sample(): (2,1,V) logits
    pair = torch.randn(2, 1, V, generator=g) * 3.0
    torch.manual_seed(0)
    _, pmix = ref_fiu.sample(pair, guidance_scale=torch.tensor(3.0), temperature=torch.tensor(1.0), top_p=torch.tensor(0.95))
    _, omix = o_s1.sample(pair, 3.0, 1.0, top_p=0.95)
    worst = max(worst, float((omix - pmix).abs().max()))
    print(f"[sampling] oracle vs reference max|dprob|={worst:.3e}")
    np.savez_compressed(os.path.join(HERE, "sampling.npz"), logits=logits.numpy(), combos=json.dumps(combos), probs=probs,
                        cfg_pair=pair.numpy(), cfg_probs=pmix.numpy())


def quant_case():
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(64, 512, generator=g) * 0.02).to(torch.bfloat16)
    w[3, :128] = 0.0  # a constant group: exercises the 1e-6 clamp
    q, sz = ref_fq.group_quantize_tensor(w, n_bit=4, groupsize=128)
    scales, zeros = ref_fq.get_group_qparams(w, 4, 128)
    deq = ref_fq.group_dequantize_tensor_from_qparams(q, scales, zeros, 4, 128)
    q8, s8, _ = ref_fq.dynamically_quantize_per_channel(w.float(), -128, 127, torch.int8)
    oq, osz = o_quant.group_quantize_tensor(w)
    od = o_quant.group_dequantize(oq, osz, out_dtype=torch.bfloat16)
    oq8, os8 = o_quant.quantize_per_channel_int8(w.float())
    print(f"[quant] int4 codes equal={bool((oq == q).all())} sz equal={bool((osz == sz).all())} "
          f"deq equal={bool((od == deq).all())} int8 equal={bool((oq8 == q8).all())} scales equal={bool((os8 == s8).all())}")
    np.savez_compressed(os.path.join(HERE, "quant.npz"), w=w.float().numpy(), q4=q.numpy().astype(np.int8),
                        scales_and_zeros=sz.float().numpy(), deq=deq.float().numpy(), q8=q8.numpy(), s8=s8.numpy())


def adapters_case():
    rng = np.random.default_rng(3)
    cases = []
    for n in (0, 1, 7, 40, 41):
        toks = [int(t) for t in rng.integers(2049, 2562, size=5)] + [int(t) for t in rng.integers(0, 2049, size=n)]
        toks += [2048] if n % 2 else []
        t_text, t_audio = FlattenedInterleavedEncodec2Codebook(end_of_audio_token=1024).decode([toks])
        cases.append(dict(kind="flattened", tokens=toks, text=t_text, audio=t_audio))
        assert o_adapters.flattened_interleaved_decode(toks) == (t_text, t_audio)
    for t in (3, 16):
        grid = [[int(x) for x in rng.integers(1025, 1538, size=4)] + [int(x) for x in rng.integers(0, 1025, size=t)]]
        for r in range(7):
            grid.append([1024] * 4 + [int(x) for x in rng.integers(0, 1026, size=t)])
        t_text, t_audio = TiltedEncodec(end_of_audio_token=1024).decode(grid)
        cases.append(dict(kind="tilted", tokens=grid, text=t_text, audio=t_audio))
        assert o_adapters.tilted_decode(grid) == (t_text, t_audio)
    texts = ["Hello  “world” — it’s 12…", "a\tb\nc  *d*  ", "plain ascii", "café naïve"]
    norm = []
    for s in texts:
        norm.append(dict(text=s, out=ref_normalize_text(s)))
        assert o_adapters.normalize_text(s) == norm[-1]["out"]
    try:
        ref_normalize_text("你好")
        raised = False
    except ValueError:
        raised = True
    norm.append(dict(text="你好", raises=raised))
    with open(os.path.join(HERE, "adapters.json"), "w") as f:
        json.dump(dict(cases=cases, normalize=norm), f)
    print(f"[adapters] {len(cases)} adapter cases + {len(norm)} normalize_text cases pinned")


def speaker_encoder_case():
    """The reference's `SpeakerEncoder` class itself (librosa stubbed: only `embed_utterance_from_file` / the mel front end need it)
    on this repo's synthetic `speaker_encoder.pt`: `forward` on random mel partials + `compute_partial_slices`."""
    import tempfile

    from fam.quantiser.audio.speaker_encoder.model import SpeakerEncoder as RefSpeakerEncoder
    from oracle import speaker_encoder as o_spk

    sd = synthetic.speaker_encoder_state_dict(seed=4)
    with tempfile.TemporaryDirectory() as d:
        torch.save({"model_state": sd}, os.path.join(d, "speaker_encoder.pt"))
        ref = RefSpeakerEncoder(weights_fpath=os.path.join(d, "speaker_encoder.pt"), device="cpu", eval=True, verbose=False)
    g = torch.Generator().manual_seed(12)
    mels = torch.rand(5, 160, 40, generator=g) * 3.0  # mel power spectra are non-negative
    emb = ref(mels)
    o = o_spk.forward(sd, mels)
    slices = {}
    for n in (1, 15999, 25600, 48000, 51234, 160000):
        w, m = RefSpeakerEncoder.compute_partial_slices(n, rate=1.3, min_coverage=0.75)
        slices[str(n)] = [[int(s.start), int(s.stop)] for s in m]
    print(f"[speaker encoder] oracle vs reference forward max|d|={float((o - emb).abs().max()):.3e}; "
          f"partial counts {[len(v) for v in slices.values()]}")
    np.savez_compressed(os.path.join(HERE, "speaker_encoder.npz"), seed=4, mels=mels.numpy(), emb=emb.numpy(), slices=json.dumps(slices))


def packing_case():
    """Stage-2 input packing: the reference's own `Model.non_causal_sample` (fam/llm/inference.py:248-338),
    executed with the modules it imports but does not need for packing (audiocraft, tyro, df, librosa, tiktoken, posthog) stubbed,
    on an instance whose model / decoder are recorders: what reaches `self.model.generate` is the packed (b, 2, ctx) grid."""
    import importlib.machinery

    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return type(name, (), {"__init__": lambda self, *a, **k: None, "get_mbd_24khz": staticmethod(lambda *a, **k: None)})

    for m in ("audiocraft", "audiocraft.data", "audiocraft.data.audio", "audiocraft.models", "tyro", "df", "df.enhance", "df.io", "julius",
              "posthog", "pydub"):
        if m not in sys.modules:
            mod = _Any(m)
            mod.__spec__ = importlib.machinery.ModuleSpec(m, None)
            mod.__path__ = []
            sys.modules[m] = mod
    from fam.llm import inference as ref_inf

    class Tok:
        def encode(self, text):
            return [1025 + (ord(ch) % 512) for ch in text] + [1537]

    class Rec:
        def __init__(self):
            self.seen = []

        def generate(self, in_x, *a, **k):
            self.seen.append(in_x.clone())
            return torch.zeros(in_x.shape[0], 6, in_x.shape[2], dtype=torch.long)

    class Dec:
        def decode(self, tokens, causal):
            return "x"

    from contextlib import nullcontext
    cases = []
    for ctx, texts, lens in [(64, ["hello there"], [20]), (32, ["a much longer piece of text, truncated"], [40]), (48, ["ab", "cdefg"], [7, 12])]:
        m = ref_inf.Model.__new__(ref_inf.Model)
        m.tokenizer, m.model, m.decoder = Tok(), Rec(), Dec()
        m.config = ref_inf.InferenceConfig(ckpt_path="", output_dir="", num_samples=1, device="cpu")
        m._encodec_codes_pad_token, m._num_encodec_codebooks, m._encodec_ctx_window = 1024, 8, ctx
        m.speaker_cond, m._ctx = True, nullcontext()
        g = torch.Generator().manual_seed(ctx)
        enc = [torch.randint(0, 1024, (1, 2, n), generator=g) for n in lens]
        m.non_causal_sample(texts=texts, encodec_tokens=enc, batch_size=len(texts), top_k=200, temperature=1.0,
                            speaker_embs=torch.zeros(len(texts), 1, 256))
        cases.append(dict(ctx=ctx, texts=texts, text_ids=[Tok().encode(t) for t in texts], codebooks=[e[0].tolist() for e in enc],
                          packed=m.model.seen[0].tolist()))
    with open(os.path.join(HERE, "packing.json"), "w") as f:
        json.dump(cases, f)
    print(f"[packing] {len(cases)} second-stage packing cases pinned from fam/llm/inference.py")


def stage2_case(name, **kw):
    args = synthetic.stage2_model_args(block_size=32, n_layer=2, n_head=2, n_embd=128, **kw)
    sd = synthetic.stage2_state_dict(args, speaker_emb_dim=256, seed=5)
    model = GPT(GPTConfig(**args), speaker_emb_dim=256).eval()
    model.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    t = args["block_size"]
    row0 = torch.cat([torch.randint(1025, 1538, (6,), generator=g), torch.randint(0, 1024, (t - 8,), generator=g), torch.tensor([1024, 1024])])
    row1 = torch.cat([torch.full((6,), 1024), torch.randint(0, 1024, (t - 8,), generator=g), torch.tensor([1024, 1024])])
    idx = torch.stack([row0, row1])[None].long()
    spk = synthetic.speaker_embedding(2)
    logits, _ = model(idx, speaker_embs=spk)
    torch.manual_seed(0)
    toks = model.generate(idx, None, temperature=1.0, top_k=1, speaker_embs=spk, batch_size=1)
    cfg = o_s2.Stage2Config(**{k: v for k, v in args.items() if k in o_s2.Stage2Config.__dataclass_fields__})
    orc = o_s2.Stage2Oracle(sd, cfg, rounding="fp32")
    ol = orc.forward(idx, spk)
    d = max(float((a - b).abs().max()) for a, b in zip(ol, logits))
    ot = orc.non_causal_sample(idx, spk, temperature=1.0, top_k=1)
    print(f"[stage2 {name}] max|dlogit|={d:.3e} greedy tokens equal={bool((ot == toks).all())}")
    np.savez_compressed(os.path.join(HERE, f"stage2_{name}.npz"), model_args=json.dumps(args), seed=5, idx=idx.numpy(),
                        logits_sub=torch.stack(logits)[:, :, ::4].numpy(), tokens=toks.numpy())


if __name__ == "__main__":
    if "--extra" in sys.argv:  # cases added in round 2 (the round-1 files are reproduced bit for bit by the default run)
        speaker_encoder_case()
        packing_case()
        sys.exit(0)
    if "--long24" in sys.argv:  # round 3: ~1 h of CPU; T = 1100 and T = 2016 contexts + the reference's own bf16 picks
        which = [int(a) for a in sys.argv[sys.argv.index("--long24") + 1:] if a.isdigit()] or [128, 1100, 2016]
        for T in which:
            stage1_full_refbf16_choice() if T == 128 else stage1_long_case(T)
        sys.exit(0)
    if "--full24" in sys.argv:  # ~5 min and ~12 GB of RAM: minted separately from the small cases
        stage1_full_case()
        sys.exit(0)
    small = dict(n_layer=2, n_head=2, dim=256, vocab_size=2562, block_size=64)
    wide = dict(n_layer=2, n_head=16, dim=2048, vocab_size=2562, block_size=2048)
    slow_vs_fast_selftest(small, 0)
    stage1_case("small", small, seed=0, T=12, n_new=20, with_bf16=True)
    stage1_case("wide2", wide, seed=1, T=16, n_new=6, with_bf16=False)
    sampling_case()
    quant_case()
    adapters_case()
    stage2_case("rms_swiglu")
    stage2_case("ln_gelu_bias", bias=True, norm_type="layernorm", nonlinearity_type="gelu")
    speaker_encoder_case()
    packing_case()
    print("golden vectors written to", HERE)
