#!/usr/bin/env python3
"""bench.py -- audio-seconds per wall-second (RTF^-1) of the MI355X-native MetaVoice-1B engine.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 outside a launcher re-executes itself under `torch.distributed.run` (one rank per GPU, 127.0.0.1); under a
launcher it checks that the launcher started exactly N ranks.  The line carries `rccl_ranks` = the size of the process group.
The default N = 1 line also carries `extra_configs`: short measured legs of the other BASELINE.json configurations (int4 and int8
first-stage decode, 4 utterances per GPU with the full vocoder, long-form 10 000 characters), each with its own value,
ms_per_step, roofline and decode_fallbacks (`--no-extra-configs` skips them).

A "step" is one pass of the hot path over one utterance per GPU: BASELINE.json configs[1] -- MetaVoice-1B bf16, one
utterance, synthetic 256-char-equivalent prompt (T = 128 text tokens), batched MFMA prefill, N = 1920 generated
first-stage tokens forced (end-of-audio unreachable) = 12.8 s of audio at 150 tokens/s, token adapter,
second-stage expansion to 8 codebooks (placeholder second-stage sizes: the real model_args are not reachable offline),
vocoder = EnCodec decode (8 codebooks -> 24 kHz waveform) + the multi-band diffusion refinement (4 band UNets x 20 DDPM
evaluations, processor, 32-band re_eq: what `mbd.tokens_to_wav` does in the reference), random-init weights of the exact
MetaVoice-1B first-stage and EnCodec shapes and of the recalled audiocraft MBD sizes.  Inputs (weights, prompt, speaker embedding) are resident
before the timed region; the timed region is bracketed by barrier + torch.cuda.synchronize on both sides and the MAX
over ranks is reported.  Utterances are independent, so N GPUs run N utterances with no data-path collective; the
finished token streams are all-gathered over RCCL once per step (batched-serving path), inside the timed region.

Other BASELINE.json configurations (not the default line; each prints the same JSON shape with its own `config.workload`):
  --quantisation_mode int4|int8   configs[3]: the first stage decodes from weight-only quantised weights (reference numerics)
  --utterances-per-gpu U          configs[2]: U utterances decoded TOGETHER per GPU (ragged prompts 96..160 tokens), the finished
                                  fp32 waveforms all-gathered over RCCL (with --gpus 8 --utterances-per-gpu 4: batch 32)
  --long-form [--long-form-chars C]  configs[4]: a C-character text (default 10 000) split at sentence boundaries into <= 220-char
                                  chunks, each an independent generation on the same captured decode graph; a "step" is the whole text

Extra objects on the JSON line:
  roofline      -- dominant kernel: the persistent decode-step kernel (one launch per token) where supported, else the
                   largest per-layer kernel of the decode step (fused RMSNorm + w1|w3 + SiLU-gate GEMV, 46 MB of weights per
                   launch): algorithmic bytes / average launch duration measured with HIP events on the engine stream.
  step_roofline -- the same ratio for a whole decode step (all ~122 launches), the number RTF actually follows.
  cpu_baseline  -- the CPU oracle (a torch-CPU fp32 restatement of the reference, golden-pinned to it) timed on this
                   host's cores on a bounded sample, rank 0, N=1 only.
"""
import argparse
import contextlib
import copy
import importlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured achievable)
TOKENS_PER_AUDIO_SECOND = 150.0
T_PROMPT, N_NEW = 128, 1920


def weight_bytes(cfg, fmt=None):
    """bytes of weights one decode step must stream.  bf16: 2 B per parameter; int4 (group 128): half a byte per
    weight + 4 B of (scale, zero) per 128 weights, LM head and norms stay bf16 (fast_quantize.py:225); int8: 1 B per weight + a
    bf16 scale per output row, LM head included.  The speaker projection is hoisted out of the step (step-invariant)."""
    lin = 3 * cfg.dim * cfg.dim + cfg.dim * cfg.dim + 3 * cfg.dim * cfg.intermediate_size   # weights per layer
    rows = 3 * cfg.dim + cfg.dim + 2 * cfg.intermediate_size + cfg.dim                      # output rows per layer
    norms = 2 * cfg.dim * 2
    head = cfg.vocab_size * cfg.dim
    if fmt == "int4":
        return cfg.n_layer * (lin // 2 + lin // 128 * 4 + norms) + head * 2 + cfg.dim * 2
    if fmt == "int8":
        return cfg.n_layer * (lin + rows * 2 + norms) + head + cfg.vocab_size * 2 + cfg.dim * 2
    return 2 * (cfg.n_layer * (lin + 2 * cfg.dim) + head + cfg.dim)


MEGA_STEPS_PER_LAUNCH = 512  # mv_tune("mega_steps") default: decode steps per launch of the persistent kernel that draws its own tokens


def decode_launches(work, steps_per_launch=MEGA_STEPS_PER_LAUNCH):
    """Launches of the multi-step persistent kernel in one bench step: a group of n new tokens takes n - 1 decode steps after the
    prefill (which draws the first token), `mv_run` cuts them into launches of at most `steps_per_launch`."""
    return sum(-(-(max(n for _, n in group) - 1) // steps_per_launch) for group in work)


def kv_bytes(cfg, pos):
    """K and V rows [0, pos] of both CFG rows of ONE utterance, all layers (bf16)."""
    return 2 * 2 * cfg.n_head * 128 * 2 * cfg.n_layer * (pos + 1)


def step_bytes(cfg, pos, fmt=None):
    """Algorithmic bytes of one decode step of one utterance at input position `pos`."""
    return weight_bytes(cfg, fmt) + kv_bytes(cfg, pos)


def make_work(mv, args, *, long_form, long_form_chars, utterances_per_gpu, rank, world):
    """The work list of one bench step on this rank: groups of (prompt tokens, tokens to generate), one group per engine run."""
    U = utterances_per_gpu
    if long_form:
        # configs[4]: C characters -> sentence chunks of <= 220 characters (longform.split_text), tokenised with the synthetic
        # 512-rank BPE table (the real ranks live in the unreachable checkpoint); every chunk speaks 7.5 tokens per character,
        # the rate of configs[1] (256 characters -> 1920 tokens), end-of-audio unreachable
        longform = importlib.import_module("metavoice-src_amd.longform")
        tok_mod = importlib.import_module("metavoice-src_amd.tokenise")
        tokenizer = tok_mod.TrainedBPETokeniser(**tok_mod.byte_level_tokenizer_info(offset=2049))
        sentence = "The quick brown fox jumps over the lazy dog near the quiet river bank, while the old mill turns slowly. "
        text = (sentence * (long_form_chars // len(sentence) + 1))[:long_form_chars]
        items = []
        for c in longform.split_text(text):
            ids = torch.tensor(tokenizer.encode(c), dtype=torch.int32)
            items.append((ids, min(args.block_size - len(ids), int(round(7.5 * len(c))))))
        # U chunks decode TOGETHER in one engine run (what `TTS.synthesise_long` -> `synthesise_wavs` does with max_utterances = U):
        # the chunks of a text are independent generations with the same speaker embedding; consecutive chunks share a group
        return [items[i:i + U] for i in range(0, len(items), U)]
    if U > 1:
        # configs[2]: ragged prompts, lengths uniform in [96, 160] (seed 0 stream, offset by the rank), 12.8 s each unless the
        # context window ends first (160 + 1920 > 2048)
        g = torch.Generator().manual_seed(0)
        lens = torch.randint(96, 161, (U * max(world, 1),), generator=g).tolist()[rank * U:(rank + 1) * U]
        return [[(mv.synthetic.prompt_tokens(t, seed=rank * 100 + u), min(N_NEW, args.block_size - t)) for u, t in enumerate(lens)]]
    return [[(mv.synthetic.prompt_tokens(T_PROMPT, seed=rank), N_NEW)]]


def reference_dir():
    """The reference checkout, when this machine has one ($MV_REFERENCE or /root/reference -- the build container does, the GPU box does not)."""
    for d in (os.environ.get("MV_REFERENCE"), "/root/reference"):
        if d and os.path.isfile(os.path.join(d, "fam", "llm", "fast_inference_utils.py")):
            return d
    return None


def pick_cpu_threads(step_fn, trial_steps=3):
    """A decode step streams 5 GB of fp32 weights: memory-bound, and on a many-core host torch's default (one thread per logical core)
    can be several times SLOWER than a fraction of the cores (round 4: 128 threads -> 2 tok/s on the GPU box's EPYC, 8 threads -> 13 tok/s
    on the build container).  Time `trial_steps` steps at all of the cores, then at half, a quarter, ... and go on halving UNTIL THE TIME PER
    TOKEN TURNS (a setting slower than the one before it), down to one thread: the minimum then lies inside the scanned range, not on its
    edge (round 5's scan stopped at an eighth of the cores while still improving).  -> (threads, {threads: s/token})."""
    n0 = torch.get_num_threads()
    seen = {}
    n = n0
    while n >= 1:
        torch.set_num_threads(n)
        step_fn()   # warm the pool at this size
        t = time.perf_counter()
        for _ in range(trial_steps):
            step_fn()
        seen[n] = (time.perf_counter() - t) / trial_steps
        if len(seen) >= 2 and seen[n] > 1.03 * min(v for k, v in seen.items() if k != n):
            break   # turned: fewer threads are slower again
        if n == 1:
            break
        n = max(1, n // 2)
    best = min(seen, key=seen.get)
    torch.set_num_threads(best)
    return best, {k: round(v, 4) for k, v in seen.items()}


def cpu_baseline_port(mv, cfgk, sample_steps=24):
    """Time the oracle (oracle/stage1.py, fp32, torch CPU) on the same workload shape: bounded sample."""
    from oracle import stage1 as o_s1

    sd = o_s1.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, **cfgk))
    cfg = o_s1.Stage1Config(**cfgk)
    orc = o_s1.Stage1Oracle({k: v.float() for k, v in sd.items()}, cfg, rounding="fp32", max_seq_length=T_PROMPT + sample_steps + 8)
    del sd
    prompt = mv.synthetic.prompt_tokens(T_PROMPT, seed=0)
    spk = mv.synthetic.speaker_embedding(0)
    n_default = torch.get_num_threads()
    t0 = time.perf_counter()
    logits = orc.forward(prompt.view(1, -1).repeat(2, 1), spk, torch.arange(T_PROMPT))
    tok = o_s1.sample(logits, 3.0, 1.0, top_k=1)[0]
    t1 = time.perf_counter()
    threads, scan = pick_cpu_threads(lambda: orc.forward(torch.full((2, 1), tok), spk, torch.tensor([T_PROMPT])))   # rewrites cache row T: harmless
    t1b = time.perf_counter()
    for i in range(sample_steps):
        logits = orc.forward(torch.full((2, 1), tok), spk, torch.tensor([T_PROMPT + i]))
        tok = o_s1.sample(logits, 3.0, 1.0, top_k=1)[0]
    t2 = time.perf_counter()
    torch.set_num_threads(n_default)
    tok_s = sample_steps / (t2 - t1b)
    return {
        "value": round(tok_s / TOKENS_PER_AUDIO_SECOND, 5), "unit": "audio-s/s", "cores": threads,
        "kind": "port",
        "sample": f"oracle/stage1.py fp32 eager torch-CPU, full-size synthetic MetaVoice-1B: prefill T={T_PROMPT} "
                  f"({t1 - t0:.2f} s, {n_default} threads) + {sample_steps} greedy decode steps on {threads} threads ({(t2 - t1b) / sample_steps:.3f} s/token, "
                  f"{tok_s:.2f} tok/s; thread scan s/token {scan}: the fastest setting is the one timed); value = decode tok/s / 150; "
                  f"host has {os.cpu_count()} logical cores",
    }


def cpu_baseline_reference(mv, cfgk, ref_dir, sample_steps=24):
    """Time the REFERENCE itself -- `fam.llm.fast_inference_utils.prefill` / `decode_n_tokens` (what `generate()` runs, `:182-228`) on
    `fam.llm.fast_model.Transformer`, eager fp32 on the CPU (BASELINE configs[0]) -- on the same full-size synthetic checkpoint,
    prompt and speaker embedding as `cpu_baseline_port`.  Only where the reference checkout exists; `librosa` / `tiktoken` are imported at
    module top by the reference but not used by the model code: stubbed, as in tests/golden/make_golden.py."""
    import types

    from oracle import stage1 as o_s1

    for m in ("librosa", "tiktoken"):
        sys.modules.setdefault(m, types.ModuleType(m))
    if ref_dir not in sys.path:
        sys.path.insert(0, ref_dir)
    from fam.llm import fast_inference_utils as ref_fiu
    from fam.llm.fast_model import ModelArgs as RefArgs
    from fam.llm.fast_model import Transformer as RefTransformer

    with torch.no_grad():
        model = RefTransformer(RefArgs(n_layer=cfgk["n_layer"], n_head=cfgk["n_head"], dim=cfgk["dim"], vocab_size=cfgk["vocab_size"],
                                       block_size=cfgk["block_size"]))
        sd = o_s1.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, **cfgk))
        model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
        del sd
        model = model.float().eval()
        model.config.dtype = torch.float32   # ModelArgs picks fp16 when no GPU is present (fam/llm/utils.py:83-84)
        model.setup_spk_cond_mask()
        model.setup_caches(max_batch_size=2, max_seq_length=T_PROMPT + sample_steps + 8)
        prompt = mv.synthetic.prompt_tokens(T_PROMPT, seed=0)
        spk = mv.synthetic.speaker_embedding(0)
        kw = dict(temperature=torch.tensor(1.0), top_k=1, top_p=None, guidance_scale=torch.tensor(3.0))
        n_default = torch.get_num_threads()
        t0 = time.perf_counter()
        nxt = ref_fiu.prefill(model, prompt.view(1, -1).repeat(2, 1), spk, torch.arange(T_PROMPT), **kw)
        t1 = time.perf_counter()
        cur = nxt.view(1, -1).repeat(2, 1)
        threads, scan = pick_cpu_threads(lambda: ref_fiu.decode_one_token(model, cur, spk, torch.tensor([T_PROMPT]), **kw))
        t1b = time.perf_counter()
        with contextlib.redirect_stderr(io.StringIO()):   # the reference's tqdm bar
            toks, _ = ref_fiu.decode_n_tokens(model, cur, spk, torch.tensor([T_PROMPT]), sample_steps,
                                              callback=lambda x: x, end_of_audio_token=9999, **kw)
        t2 = time.perf_counter()
        torch.set_num_threads(n_default)
    tok_s = len(toks) / (t2 - t1b)
    return {
        "value": round(tok_s / TOKENS_PER_AUDIO_SECOND, 5), "unit": "audio-s/s", "cores": threads,
        "kind": "reference",
        "sample": f"the reference's own fam.llm.fast_inference_utils.prefill + decode_n_tokens on fam.llm.fast_model.Transformer ({ref_dir}), fp32 eager "
                  f"torch-CPU, full-size synthetic MetaVoice-1B: prefill T={T_PROMPT} ({t1 - t0:.2f} s, {n_default} threads) + {len(toks)} greedy decode steps on "
                  f"{threads} threads ({(t2 - t1b) / len(toks):.3f} s/token, {tok_s:.2f} tok/s; thread scan s/token {scan}: the fastest setting is the one timed); "
                  f"value = decode tok/s / 150; host has {os.cpu_count()} logical cores",
    }


def cpu_baseline(mv, cfgk, sample_steps=24):
    """`cpu_baseline` of the bench line: the reference itself where its checkout exists (kind "reference"), else its golden-pinned port
    (kind "port": oracle/stage1.py).  profiles/cpu_baseline_pair.json holds both timed back to back on ONE box (the build container),
    so the ratio port / reference is a known number for lines that could only time the port (the GPU box has no reference checkout)."""
    ref = None if os.environ.get("MV_BENCH_CPU_PORT") else reference_dir()
    if ref is not None:
        try:
            return cpu_baseline_reference(mv, cfgk, ref, sample_steps)
        except Exception as e:   # e.g. a torch too new for the reference's calls: fall back to the port and say why
            out = cpu_baseline_port(mv, cfgk, sample_steps)
            out["sample"] += f"; the reference at {ref} could not be timed ({type(e).__name__}: {e})"[:300]
            return out
    out = cpu_baseline_port(mv, cfgk, sample_steps)
    try:
        with open(os.path.join(ROOT, "profiles", "cpu_baseline_pair.json")) as f:
            pair = json.load(f)
        out["port_over_reference_same_box"] = pair["port_over_reference"]
        out["sample"] += f"; no reference checkout on this machine -- on one box ({pair['host']}) port / reference = {pair['port_over_reference']}"
    except (OSError, KeyError, ValueError):
        pass
    return out


def measure(a, rank, local_rank, world, dist, shared, with_cpu_baseline):
    """One bench configuration (the flags in `a`): builds its first-stage engine, runs `a.warmup` + `a.steps` bench steps between
    fences and returns the JSON object of the line (rank 0; None elsewhere).  Second stage, codec and diffusion bands are built
    once per process (`shared`) and reused by every configuration."""
    U = a.utterances_per_gpu
    mv = importlib.import_module("metavoice-src_amd")
    args = mv.ModelArgs.from_name("metavoice-1B")
    cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)

    # ---- resident inputs: weights drawn on the GPU, copied into the engine's arena, then dropped -------------------
    fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
    sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=rank, device="cuda", **cfgk))
    engine = mv.Stage1Engine(args, max_utterances=U, quantisation_mode=a.quantisation_mode)
    engine.load_state_dict(sd)
    del sd
    torch.cuda.empty_cache()
    engine.finalize()
    if a.decode_path != "auto":
        engine.tune(mega=int(a.decode_path == "step"), mega_fold=0)
    spk = mv.synthetic.speaker_embedding(rank)
    sp = mv.SamplingParams(guidance_scale=3.0, temperature=1.0, top_p=0.95, seed=1337 + rank)  # TTS.synthesise defaults
    work = make_work(mv, args, long_form=a.long_form, long_form_chars=a.long_form_chars, utterances_per_gpu=U, rank=rank, world=world)
    prompt = work[0][0][0]
    dispatch = importlib.import_module("metavoice-src_amd.batch_dispatch")

    # second stage (fam/llm/inference.py Model + model.py GPT, non-causal): placeholder sizes, block_size 2048 so that the
    # 960 frames + 128 text ids of this workload fit untruncated
    adapters = importlib.import_module("metavoice-src_amd.adapters")
    inference = importlib.import_module("metavoice-src_amd.inference")
    if "stage2" not in shared:
        s2_args = mv.synthetic.stage2_model_args(block_size=2048)
        stage2 = mv.GPT(mv.GPTConfig(**s2_args), speaker_emb_dim=256, max_batch=1)
        stage2.load_state_dict(mv.synthetic.stage2_state_dict(s2_args, speaker_emb_dim=256, seed=100 + rank))
        # codec stage: EnCodec's own decoder (8 codebooks -> 24 kHz waveform + the latent the diffusion bands are conditioned on) in HIP
        codec_mod = importlib.import_module("metavoice-src_amd.codec")
        shared.update(s2_args=s2_args, stage2=stage2,
                      vocoder=codec_mod.EncodecDecoder.from_state_dict(mv.synthetic.encodec_state_dict(seed=rank)))
    s2_args, stage2, vocoder = shared["s2_args"], shared["stage2"], shared["vocoder"]
    first_adapter = adapters.FlattenedInterleavedEncodec2Codebook(end_of_audio_token=1024)
    second_adapter = adapters.TiltedEncodec(end_of_audio_token=1024)
    stage1_only = a.quantisation_mode is not None  # configs[3] is "first-stage decode" only
    # multi-band diffusion on top of the codec (fam/llm/decoders.py:13,84-85): 4 band models x 20 evaluations + processor + 32-band re_eq.
    # The released mbd_comp_8.pt is not reachable offline: the sizes are audiocraft's config/model/score/basic.yaml as recalled
    # (hidden 48, depth 4, growth 4, kernel 8, stride 4: 166 M parameters per band), synthetic weights.
    diffusion = None
    if a.mbd_precision and a.quantisation_mode is None:
        key = ("diffusion", a.mbd_precision)
        if key not in shared:
            mbd_mod = importlib.import_module("metavoice-src_amd.mbd")
            shared[key] = mbd_mod.MultiBandDiffusionBands.from_package(mv.synthetic.mbd_package(seed=rank), seed=1234 + rank, precision=a.mbd_precision)
        diffusion = shared[key]
    stage_ms = {"stage1_prefill+decode": 0.0, "stage1_begin(upload+prefill+first_token)": 0.0, "stage1_decode_gpu": 0.0, "adapters+stage2": 0.0, "encodec_decode": 0.0, "multiband_diffusion": 0.0}

    def fold_onto_codebooks(toks, n_prompt):
        """Random weights do not produce the cb0 / cb1 alternation a trained first stage does, and the adapter truncates both
        codebooks to the shorter one (`flattened_encodec.py:25-31`: 654 of 960 frames survived in round 1).  For the stages
        AFTER the first one the synthetic stream is folded onto strictly alternating codebooks so that they are timed on the
        frame count the metric counts; the first stage's own arithmetic and timing are untouched."""
        gen = toks[n_prompt:]
        return toks[:n_prompt] + [(t % 1024) + 1024 * (i & 1) for i, t in enumerate(gen)]

    def one_step():
        n_gen, run_ms, wavs = 0, 0.0, []
        for group in work:
            t0 = time.perf_counter()
            prompts = [p for p, _ in group]
            n_new = max(n for _, n in group)
            caps = [n for _, n in group]
            engine.begin(prompts, spk.expand(len(group), -1), [mv.SamplingParams(guidance_scale=3.0, temperature=1.0, top_p=0.95,
                                                                              seed=sp.seed + 7 * u) for u in range(len(group))],
                         max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
            stage_ms["stage1_begin(upload+prefill+first_token)"] += (time.perf_counter() - t0) * 1e3   # mv_begin returns after a stream sync
            engine.run(poll_every=1 << 20)
            streams = [engine.tokens(u)[:len(prompts[u]) + caps[u]] for u in range(len(group))]
            t1 = time.perf_counter()
            stage_ms["stage1_prefill+decode"] += (t1 - t0) * 1e3
            stage_ms["stage1_decode_gpu"] += engine.last_run_ms
            run_ms += engine.last_run_ms
            pending = []   # (latent, wav_encodec) of the group's utterances when their diffusion bands run on concurrent lanes
            many = diffusion is not None and len(group) > 1 and a.vocoder_lanes > 1
            for u, toks in enumerate(streams):
                n_gen += len(toks) - len(prompts[u])
                if stage1_only:
                    continue
                t1 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):  # "Truncating to min length": an utterance capped by the context window at an odd count
                    _, cbs = first_adapter.decode([fold_onto_codebooks(toks, len(prompts[u]))])
                text_ids = [int(t) - 1024 for t in prompts[u].tolist()]  # the same text in the second stage's id space (offset 1025)
                grid = inference.pack_second_stage_input(text_ids, cbs, s2_args["block_size"])
                in_x = torch.tensor([grid], dtype=torch.long)
                y = stage2.generate(in_x, None, temperature=1.0, top_k=200, speaker_embs=spk, batch_size=1, seed=sp.seed)
                with contextlib.redirect_stdout(io.StringIO()):  # the adapter's "Truncating to min length" warning: random weights emit pad ids
                    _, audio_ids = second_adapter.decode(torch.cat([in_x, y], dim=1)[0].tolist())
                t2 = time.perf_counter()
                grid8 = torch.tensor(audio_ids, dtype=torch.long).unsqueeze(0)
                if many:
                    # a batch: EnCodec decodes one after the other, the diffusion bands of up to --vocoder-lanes utterances concurrently below
                    wav_encodec, latent = vocoder.decode(grid8, return_latent=True)
                    pending.append((latent, wav_encodec))
                    t3 = time.perf_counter()
                elif diffusion is not None:
                    # the codec's waveform and latent stay in device memory for the diffusion bands (vocoder.MultiBandDiffusion.tokens_to_wav)
                    d_wav, d_lat, samples, frames = vocoder.decode_device(grid8)
                    t3 = time.perf_counter()
                    wav = diffusion.tokens_to_wav_device(d_lat, frames, d_wav, samples)
                    stage_ms["multiband_diffusion"] += (time.perf_counter() - t3) * 1e3
                else:
                    wav = vocoder.decode(grid8)
                    t3 = time.perf_counter()
                stage_ms["encodec_decode"] += (t3 - t2) * 1e3
                stage_ms["adapters+stage2"] += (t2 - t1) * 1e3
                if not many:
                    wavs.append(wav.reshape(-1))
            if pending:
                t3 = time.perf_counter()
                wavs += [w.reshape(-1) for w in diffusion.tokens_to_wav_many(pending, lanes=a.vocoder_lanes)]
                stage_ms["multiband_diffusion"] += (time.perf_counter() - t3) * 1e3
        if world > 1 and wavs:
            # batched-serving path: every rank ends up with every utterance's fp32 waveform, in input order
            dev = torch.device("cuda", local_rank)
            everyone = dispatch.gather_variable([w.to(dev) for w in wavs], world * len(wavs), device=dev)
            assert len(everyone) == world * len(wavs)
        return n_gen, run_ms

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    for k in stage_ms:
        stage_ms[k] = 0.0
    fence()
    t0 = time.perf_counter()
    gpu_ms = 0.0
    n_tok = 0
    for _ in range(a.steps):
        n_gen, ms = one_step()
        gpu_ms += ms
        n_tok += n_gen
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    audio_s = world * n_tok / TOKENS_PER_AUDIO_SECOND  # every rank generates the same number of tokens
    value = audio_s / elapsed

    if rank == 0:
        # ---- roofline of the dominant kernel, live, HIP events on the engine stream -------------------------------
        import ctypes as C
        N = importlib.import_module("metavoice-src_amd._native")
        ms, nbytes = C.c_float(0), C.c_double(0)
        # the decode step is ONE persistent launch per token (decode_mega.hip: 24 layers + LM head, all utterances of the GPU)
        # when the configuration supports it; probe 6 times that kernel alone.  Otherwise (per-layer launches) the dominant kernel
        # is the fused RMSNorm + w1|w3 + SiLU-gate GEMV (probe 3).
        # The kernel's duration grows linearly with the context; it is probed at the MEAN context of the first timed group, so
        # that `avg_launch_us` is the average launch of the timed region (what the rocprofv3 --stats average in profiles/ shows)
        # and `bytes_per_launch` the average algorithmic bytes of those launches.
        path, timeouts = engine.decode_path()   # of the timed region
        engine.tune(mega_fold=0)                # the probe times the one-step instance; its set-up step must not add a launch of the timed kernel
        g0 = work[0]
        ctxs = [len(p) + (n - 1) // 2 for p, n in g0]
        engine.begin([mv.synthetic.prompt_tokens(c, seed=rank + u) for u, c in enumerate(ctxs)], spk.expand(len(g0), -1), sp,
                     max_new_tokens=2, end_of_audio_token=9999, prefill_mode=1)
        engine.run(max_steps=1)
        if N.lib().mv_probe_kernel(engine._h, 6, 200, C.byref(ms), C.byref(nbytes)) == 0:
            kernel = (f"decode_mfma_kernel<{U}> (matrix-core instance of the persistent step: v_mfma_f32_16x16x32_bf16 on a tiled weight copy)" if path == 3
                      else "decode_mega_kernel<.., false>")
            kname = ("%s (persistent: 24 layers + LM head of one token for %d utterance(s); probed at context %s = the "
                     "mean of the timed utterances)" % (kernel, len(g0), [c + 1 for c in ctxs]))
        else:
            N.check(N.lib().mv_probe_kernel(engine._h, 3, 24 * 20, C.byref(ms), C.byref(nbytes)))
            kname = "gemv_kernel<SWIGLU> (RMSNorm + w1|w3 + SiLU gate; per-layer launches)"
        engine.tune(mega_fold=1)                # mv_tune settings are process-wide: the legs that follow draw their tokens in the launch again
        ach = nbytes.value / (ms.value * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kname,
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": None, "bytes_per_launch": nbytes.value, "avg_launch_us": round(ms.value * 1e3, 2)}
        # HBM traffic per launch comes from a separate rocprofv3 --pmc FETCH_SIZE pass (it cannot be collected from inside this
        # process); the committed summary of the last pass is quoted with the launches it was measured on
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "roofline_traffic.json")) as f:
                tr = json.load(f)
            if tr.get("launch_form") != "multi-step" and tr["kernel"] in kname and tr.get("mode", "bf16") == (a.quantisation_mode or "bf16") and tr.get("utterances", 1) == U:
                roofline["traffic"] = tr["traffic_bytes_per_launch"]
                roofline["traffic_note"] = (f"{tr['source']}; {tr['launch_contexts']}; algorithmic bytes of those launches "
                                            f"{tr['algorithmic_bytes_same_launches']} (ratio {tr['ratio']})")
        except (OSError, KeyError, ValueError):
            pass
        per_kernel = {}
        if a.quantisation_mode is None and U <= 4:   # the per-layer launches (what these probes time) stop at 8 rows
            for kind, name in [(0, "norm+wqkv+kv"), (1, "attention@pos"), (2, "wo+res"), (4, "w2+res"), (5, "norm+head")]:
                N.check(N.lib().mv_probe_kernel(engine._h, kind, 24 * 10, C.byref(ms), C.byref(nbytes)))
                per_kernel[name] = {"us": round(ms.value * 1e3, 2), "GB/s": round(nbytes.value / (ms.value * 1e-3) / 1e9, 1)}
        # the decode steps mv_run timed: weights once per step (shared by the utterances of a group), K/V per utterance
        total_bytes, n_steps = 0, 0
        for group in work:
            steps = max(n for _, n in group) - 1
            n_steps += steps
            total_bytes += steps * weight_bytes(args, a.quantisation_mode)
            for p, n in group:
                total_bytes += sum(kv_bytes(args, q) for q in range(len(p), len(p) + n - 1))
        step_frac = total_bytes / ((gpu_ms / a.steps) * 1e-3) / 1e9 / HBM_PEAK_GBS
        if timeouts:
            # an abandoned persistent launch is redone by the per-layer launches (same tokens): the wall-clock value stands, but the
            # timed region is then not the kernel `roofline` would describe -- say so instead of attributing the time to it
            print(f"bench: {timeouts} launch(es) of the persistent decode kernel were abandoned and redone by per-layer launches",
                  file=sys.stderr)
        if path == 2 and not timeouts:
            # The persistent kernel draws the tokens itself and runs up to MEGA_STEPS decode steps per launch: the dominant kernel's
            # launches ARE the timed decode region.  achieved = algorithmic bytes of those steps / HIP-event time of mv_run;
            # per launch = both divided by the launches of one bench step (what `rocprofv3 --stats` averages over).  The single-step
            # probe above (no token draw, one launch per step) stays on file as `step_probe`.
            launches = decode_launches(work)
            run_ms = gpu_ms / a.steps
            ach = total_bytes / (run_ms * 1e-3) / 1e9
            step_probe = {k: roofline[k] for k in ("kernel", "achieved", "frac", "bytes_per_launch", "avg_launch_us")}
            roofline = {"bound": "hbm",
                        "kernel": (f"decode_mega_kernel<.., true> (persistent: up to {MEGA_STEPS_PER_LAUNCH} decode steps per launch -- per step 24 layers + LM head + "
                                   f"the token draw and the next input rows -- for {U} utterance(s); the {launches} launches of one bench step = "
                                   f"its {n_steps} decode steps)"),
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": None, "bytes_per_launch": total_bytes / launches, "avg_launch_us": round(run_ms * 1e3 / launches, 1),
                        "launches_per_bench_step": launches, "avg_step_us": round(run_ms * 1e3 / n_steps, 2), "step_probe": step_probe}
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "roofline_traffic.json")) as f:
                    tr = json.load(f)
                if tr.get("launch_form") == "multi-step" and tr.get("mode", "bf16") == (a.quantisation_mode or "bf16") and tr.get("utterances", 1) == U:
                    # FETCH_SIZE is per dispatch; the ratio traffic / algorithmic bytes of the profiled launches carries over
                    roofline["traffic"] = round(tr["ratio"] * total_bytes / launches)
                    roofline["traffic_note"] = (f"{tr['source']}; {tr['launch_contexts']}; HBM bytes / algorithmic bytes of those launches = "
                                                f"{tr['ratio']}, applied to this run's bytes per launch")
            except (OSError, KeyError, ValueError):
                pass
        n_utt = sum(len(gp) for gp in work)
        tok_per_step = sum(n for gp in work for _, n in gp)
        mode = a.quantisation_mode or "bf16"
        voc = ("EnCodec decode + multi-band diffusion (4 bands x 20 evaluations of a 166 M-parameter UNet at the recalled audiocraft basic.yaml "
               f"sizes, {'bf16 hi/lo-split (fp32-class)' if a.mbd_precision == 3 else 'bf16'} MFMA convolutions, processor, 32-band re_eq)"
               if diffusion is not None else "EnCodec decode (multi-band diffusion switched off)")
        if a.long_form:
            workload = (f"long-form synthesis (BASELINE configs[4]): {a.long_form_chars} characters split at sentence boundaries into "
                        f"{n_utt} chunks of <= 220 characters, each an independent generation (MFMA prefill + 7.5 tokens per "
                        f"character, top_p=0.95, guidance 3.0), {U} chunk(s) decoded together per engine run ({len(work)} runs) on the same "
                        f"resident engine and speaker embedding, then token "
                        f"adapters, second stage (placeholder sizes), {voc}; {tok_per_step / 150.0:.1f} s of audio per step; "
                        "synthetic speaker embedding (assets/bria.mp3 cannot be decoded here)")
        elif a.quantisation_mode is not None:
            workload = (f"MetaVoice-1B {mode} weight-only first-stage decode (BASELINE configs[3]): prompt T={T_PROMPT} (MFMA prefill on "
                        f"the de-quantised weights) + {N_NEW} sampled tokens (top_p=0.95, guidance 3.0) = 12.8 s audio; first stage only")
        elif U > 1:
            workload = (f"MetaVoice-1B bf16 batched serving (BASELINE configs[2]): {U} utterances per GPU decoded together (ragged prompts "
                        f"96..160 tokens, up to {N_NEW} sampled tokens each), adapters, second stage (placeholder sizes), {voc}, "
                        f"fp32 waveforms all-gathered over RCCL; batch {U * world} over {world} GPU(s)")
        else:
            workload = (f"MetaVoice-1B synthesis, 1 utterance/GPU: prompt T={T_PROMPT} tokens (MFMA prefill) + {N_NEW} "
                        "sampled first-stage tokens (top_p=0.95, guidance 3.0) = 12.8 s audio, token adapters, "
                        f"second-stage expansion to 8 codebooks (placeholder sizes), vocoder on all 960 frames -> 24 kHz waveform: {voc} "
                        "(synthetic stream folded onto alternating codebooks after the first stage)")
        out = {
            "metric": "audio-sec/sec (RTF^-1) MetaVoice-1B bf16, 256-char prompt" if mode == "bf16" and not a.long_form and U == 1
                      else f"audio-sec/sec (RTF^-1) MetaVoice-1B {mode}",
            "value": round(value, 4), "unit": "audio-s/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if mode == "bf16" else f"{mode} weights x bf16 activations", "data": "synthetic",
            "config": {"workload": workload, "rows_per_gpu": 2 * U, "utterances_per_step_per_gpu": n_utt, "tokens_per_step": tok_per_step,
                       "parallelism": f"utterance-dp{world}"},
            "tokens_per_s_per_gpu": round((tok_per_step - n_utt) / (gpu_ms / a.steps * 1e-3), 1),
            "stage_ms_per_step": {k: round(v / a.steps, 2) for k, v in stage_ms.items()},
            "roofline": roofline,
            "decode_fallbacks": timeouts,   # persistent launches abandoned (time-out) and redone by per-layer launches inside the timed region
            "step_roofline": {"bound": "hbm", "achieved": round(step_frac * HBM_PEAK_GBS, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(step_frac, 4), "scope": f"the {n_steps} decode steps of one bench step (HIP-event time of mv_run)",
                              "algorithmic_bytes_per_step": total_bytes, "gpu_ms_per_step": round(gpu_ms / a.steps, 2)},
            "per_kernel": per_kernel,
        }
        # the prefill (mv_begin: state upload, speaker projection, MFMA GEMMs over 2T rows + MFMA attention, LM head, first token):
        # MFMA-bound by arithmetic intensity only from ~150 rows up; at the bench's 2T = 256 rows it is the 2.5 GB weight stream that
        # prices it, so both fractions are printed
        begin_s = stage_ms["stage1_begin(upload+prefill+first_token)"] / a.steps / len(work) * 1e-3
        rows = sum(2 * len(p) for p, _ in work[0])
        pf_flops = 2.0 * (weight_bytes(args, None) / 2) * rows
        out["prefill_roofline"] = {"bound": "hbm (weights, below ~300 rows) / mfma", "rows": rows, "ms": round(begin_s * 1e3, 3),
                                   "mfma_achieved_TFLOPs": round(pf_flops / begin_s / 1e12, 1), "mfma_frac": round(pf_flops / begin_s / 2.5e15, 4),
                                   "weight_stream_GBs": round(weight_bytes(args, a.quantisation_mode) / begin_s / 1e9, 1),
                                   "weight_stream_frac": round(weight_bytes(args, a.quantisation_mode) / begin_s / 1e9 / HBM_PEAK_GBS, 4),
                                   "note": "whole mv_begin call per engine run, host-timed (it ends in a stream synchronisation)"}
        if diffusion is not None:
            # the vocoder's dominant kernel is MFMA-bound: algorithmic flops of the band UNets' convolutions (2 x MACs, x the MFMA terms
            # per product) over the whole multi-band-diffusion stage time (which also contains its element-wise kernels: a lower bound)
            samples = 960 * 320
            chans, k_, s_ = [48, 192, 768, 3072], 8, 4
            t_len, macs, cin = samples, 0.0, 1
            for c_ in chans:
                t_len = (t_len + s_ - 1) // s_
                macs += 2 * (t_len * c_ * cin * k_) + 4 * (t_len * c_ * c_ * 3)   # strided conv + its transposed twin, 2 ResBlocks x 2 convs
                cin = c_
            useful = 2 * macs * 4 * 20                       # one product per multiply-accumulate: the arithmetic the reference's fp32 convolutions ask for
            flops = useful * a.mbd_precision                 # what the MFMA units execute: three bf16 terms per product (a_hi w_hi + a_hi w_lo + a_lo w_hi)
            mbd_s = stage_ms["multiband_diffusion"] / a.steps / n_utt * 1e-3
            out["vocoder_roofline"] = {"bound": "mfma", "kernel": "mbd_gemm_t256_kernel / mbd_gemm_glds3_kernel (overlapping-window conv GEMMs of the 4 band UNets, 20 evaluations)",
                                       "achieved": round(flops / mbd_s / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(flops / mbd_s / 2.5e15, 4),
                                       "useful_achieved": round(useful / mbd_s / 1e12, 1), "useful_frac_of_bf16_peak": round(useful / mbd_s / 2.5e15, 4),
                                       "note": f"hardware: {flops / 1e12:.0f} TFLOP of bf16 MFMA work per utterance ({a.mbd_precision} terms per product); useful: {useful / 1e12:.0f} TFLOP "
                                               "of fp32-class multiply-accumulates (for scale: gfx950's exact-fp32 MFMA peaks at 157 TFLOP/s); both over the WHOLE stage time, "
                                               "which also holds its element-wise / GroupNorm / FIR kernels (a lower bound for the GEMM kernels themselves: profiles/)"}
            # the stage's HBM-bound kernels as achieved GB/s (north_star): from the rocprofv3 --stats summary of one default bench step,
            # reduced by tools/vocoder_hbm.py (a process cannot profile itself; the committed summary is quoted with its source)
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "vocoder_hbm_kernels.json")) as f:
                    hk = json.load(f)
                # NOT measured in this run: quoted from the committed profile, labelled as such
                out["vocoder_roofline"]["hbm_kernels_quoted_from_profile"] = {
                    "source": hk["source"], "profile": "profiles/vocoder_hbm_kernels.json",
                    "kernels": {k: {x: v[x] for x in ("achieved_GBs", "frac_of_8TBs", "avg_us", "calls")} for k, v in hk["kernels"].items()}}
            except (OSError, KeyError, ValueError):
                pass
        if with_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(mv, cfgk)
            except Exception as e:  # e.g. a host without the memory for the fp32 oracle: the GPU line still stands
                out["cpu_baseline"] = {"value": None, "unit": "audio-s/s", "cores": torch.get_num_threads(), "kind": "port",
                                       "sample": f"not measured: {type(e).__name__}: {e}"[:300]}
        engine.close()
        return out
    engine.close()
    return None


BOUNDARY_TEXT = ("The quick brown fox jumps over the lazy dog while seven silver swans glide across a quiet lake, and far away a single "
                 "church bell counts the hours of an ordinary afternoon in late summer; nobody hurries, nothing changes, and yet the light "
                 "keeps moving on the water.")[:256]


def boundary_leg(rank: int, composed: dict, steps: int = 3):
    """The headline THROUGH the drop-in boundary: `TTS(model_dir).synthesise(text_256_chars, spk)` on a full-size synthetic model
    directory (`synthetic.write_model_dir`: 24 layers x 2048, the reference's checkpoint layout, written once under /tmp), timed
    with the reference's timer scope -- text normalisation, tokenise, first stage, adapters, second stage, EnCodec + multi-band
    diffusion, loudness normalisation, the `.wav` on disk (`fam/llm/fast_inference.py:129-166`, `decoders.py:40-47`).  `bench.py`'s
    main line composes the engine / second stage / codec / diffusion calls itself; this leg says what the boundary adds on top.
    The same two accommodations for random weights as the composed line, both outside the timed arithmetic: end-of-audio cannot end
    the utterance (the stream runs to the context limit, as `generate()` does without an EOA), and the stream is folded onto
    alternating codebooks before the adapter (random weights do not alternate; the adapter would truncate to the shorter one)."""
    import shutil
    import wave

    mv = importlib.import_module("metavoice-src_amd")
    fi = importlib.import_module("metavoice-src_amd.fast_inference")
    model_dir = os.environ.get("MV_BENCH_MODEL_DIR", "/tmp/mv_bench_fullsize_model")
    t_dir = time.perf_counter()
    if not os.path.exists(os.path.join(model_dir, ".complete")):
        shutil.rmtree(model_dir, ignore_errors=True)
        args = mv.ModelArgs.from_name("metavoice-1B")
        mv.synthetic.write_model_dir(model_dir, stage1=dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size,
                                                            block_size=args.block_size),
                                     stage2_args=mv.synthetic.stage2_model_args(block_size=2048), seed=rank, mbd={})
        open(os.path.join(model_dir, ".complete"), "w").close()
    t_load = time.perf_counter()
    out_dir = os.path.join(model_dir, "outputs")
    with contextlib.redirect_stdout(io.StringIO()):
        tts = fi.TTS(model_dir, seed=1337 + rank, output_dir=out_dir)
    t_ready = time.perf_counter()
    engine = tts.model.engine
    real_begin = engine.begin

    def begin_without_eoa(*args, **kw):
        kw["end_of_audio_token"] = 9999
        return real_begin(*args, **kw)

    engine.begin = begin_without_eoa
    real_decode = tts.first_stage_adapter.decode
    n_prompt = len(tts.tokenizer.encode(fi.normalize_text(BOUNDARY_TEXT)))

    def folded_decode(tokens):
        toks = tokens[0]
        return real_decode([toks[:n_prompt] + [(t % 1024) + 1024 * (i & 1) for i, t in enumerate(toks[n_prompt:])]])

    tts.first_stage_adapter.decode = folded_decode
    spk = mv.synthetic.speaker_embedding(rank)
    walls, decode_ms, begin_ms, audio_s, paths = [], [], [], [], []
    import gc
    gc.collect()   # this leg runs behind the other legs of the line: their garbage (engines, full-size state dicts) is collected here, not in a timed call
    gc.freeze()    # ... and what survives is not walked again by a full collection that happens to fall into one (r06_final5 / 6: one call of three +426 / +85 ms on the host)
    for it in range(steps + 1):   # one warm-up call
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            path = tts.synthesise(BOUNDARY_TEXT, spk)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        with wave.open(path, "rb") as w:
            secs = w.getnframes() / float(w.getframerate())
        os.unlink(path)
        if it:
            walls.append(wall * 1e3); decode_ms.append(engine.last_run_ms); audio_s.append(secs)
    path_now, timeouts = engine.decode_path()
    wall_ms, dec_ms, secs = sum(walls) / len(walls), sum(decode_ms) / len(decode_ms), sum(audio_s) / len(audio_s)
    comp_rest = composed["ms_per_step"] - composed["step_roofline"]["gpu_ms_per_step"]
    comp_per_tok = composed["step_roofline"]["gpu_ms_per_step"] / (composed["config"]["tokens_per_step"] - 1)
    n_tok = int(round(secs * TOKENS_PER_AUDIO_SECOND))
    leg = {"value": round(secs / (wall_ms * 1e-3), 4), "unit": "audio-s/s", "steps": steps, "warmup": 1, "ms_per_step": round(wall_ms, 2),
           "audio_s_per_step": round(secs, 3), "prompt_tokens": n_prompt, "generated_tokens": n_tok,
           "wall_ms_each": [round(w, 1) for w in walls], "median_wall_ms": round(sorted(walls)[len(walls) // 2], 2),
           "value_at_median_wall": round(secs / (sorted(walls)[len(walls) // 2] * 1e-3), 4),   # `value` is the mean over the calls; a host hiccup in one call (file write, page cache) shows here
           "stage1_decode_gpu_ms_each": [round(x, 1) for x in decode_ms],
           "stage1_decode_gpu_ms": round(dec_ms, 2), "rest_ms": round(wall_ms - dec_ms, 2),
           "composed_rest_ms": round(comp_rest, 2),
           "host_overhead_ms": round((wall_ms - dec_ms) - comp_rest, 2),
           "decode_us_per_token": round(dec_ms * 1e3 / max(n_tok - 1, 1), 2), "composed_decode_us_per_token": round(comp_per_tok * 1e3, 2),
           "gap_vs_composed_pct": round(100.0 * (composed["value"] - secs / (wall_ms * 1e-3)) / composed["value"], 2),
           "decode_fallbacks": timeouts, "decode_path": path_now,
           "model_dir_write_s": round(t_load - t_dir, 1), "tts_load_s": round(t_ready - t_load, 1),
           "workload": ("TTS(model_dir).synthesise(256 characters, speaker embedding) on a full-size synthetic model directory, the reference's timer scope "
                        "(normalise + tokenise -> first stage -> adapters -> second stage -> EnCodec + multi-band diffusion -> loudness -> .wav on disk); "
                        f"{n_prompt} prompt tokens (synthetic 512-rank BPE), generation to the context limit; rest_ms = wall - first-stage decode "
                        "(HIP events); host_overhead_ms = rest_ms - the composed line's non-decode time (prefill, second stage, codec, diffusion)")}
    gc.unfreeze()
    tts.first_stage_adapter.decode = real_decode
    engine.begin = real_begin
    for closer in (getattr(tts.vocoder, "close", None), getattr(engine, "close", None)):
        try:
            if closer:
                closer()
        except Exception:
            pass
    return leg


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short legs of the other BASELINE configurations the default N=1 line carries under `extra_configs`")
    ap.add_argument("--quantisation_mode", choices=["int4", "int8"], default=None)
    ap.add_argument("--utterances-per-gpu", type=int, default=1)
    ap.add_argument("--long-form", action="store_true")
    ap.add_argument("--long-form-chars", type=int, default=10000)
    ap.add_argument("--mbd-precision", type=int, default=3, choices=[0, 1, 3],
                    help="multi-band diffusion vocoder: MFMA terms per product (3 = bf16 hi/lo split, ~fp32; 1 = bf16 operands; 0 = EnCodec decode only)")
    ap.add_argument("--vocoder-lanes", type=int, default=4,
                    help="utterances of a batch whose diffusion bands run concurrently (mv_mbd_clone: shared weights, one stream and workspace per lane); 1 = one after the other")
    ap.add_argument("--decode-path", choices=["auto", "step", "launches"], default="auto",
                    help="A/B switch: auto = the engine's choice (persistent kernel drawing its own tokens where supported), step = persistent "
                         "kernel, one step per launch + sampler launch, launches = per-layer launches")
    a = ap.parse_args(argv)
    if not 1 <= a.utterances_per_gpu <= 8:
        ap.error("--utterances-per-gpu: 1..8 (3 and 5..7 run on engines of 4 / 8 slots with the spare slots idle; more than 4 is bf16 only)")
    if a.utterances_per_gpu > 4 and a.quantisation_mode is not None:
        ap.error("--utterances-per-gpu above 4 is bf16 only (engines of 8 utterances run on the matrix-core persistent kernel)")
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    return a


def self_launch_argv(a, argv, port=None):
    """`python bench.py --gpus N` (N > 1) outside a launcher: the command line this process replaces itself with -- one rank per GPU
    under `torch.distributed.run` on 127.0.0.1, the form the driver uses.  None when no re-launch is needed (N = 1, or the
    launcher's environment is already present)."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


# The other BASELINE.json configurations, as short measured legs of the default N = 1 run: flag overrides + steps.
EXTRA_CONFIGS = [
    ("configs[3] int4 first-stage decode", dict(quantisation_mode="int4", steps=3, warmup=1)),
    ("configs[3'] int8 first-stage decode", dict(quantisation_mode="int8", steps=3, warmup=1)),
    ("configs[2] 4 utterances per GPU, full vocoder", dict(utterances_per_gpu=4, steps=3, warmup=1)),
    ("configs[2'] 8 utterances per GPU (batch 64 over 8 GPUs), full vocoder", dict(utterances_per_gpu=8, steps=3, warmup=1)),
    ("configs[4] long-form 10000 characters, full vocoder, 8 chunks per engine run", dict(long_form=True, long_form_chars=10000, utterances_per_gpu=8, steps=3, warmup=1)),
]


def summarise_extra(o):
    r = o["roofline"]
    return {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"],
            "dtype": o["dtype"], "tokens_per_s_per_gpu": o["tokens_per_s_per_gpu"], "decode_fallbacks": o["decode_fallbacks"],
            "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_launch", "avg_launch_us") if k in r},
            "step_roofline_frac": o["step_roofline"]["frac"], "stage_ms_per_step": o["stage_ms_per_step"],
            "audio_s_per_step": round(o["config"]["tokens_per_step"] / TOKENS_PER_AUDIO_SECOND, 1), "workload": o["config"]["workload"]}


def main():
    a = parse_args()
    relaunch = self_launch_argv(a, sys.argv[1:])
    if relaunch is not None:
        os.execv(relaunch[0], relaunch)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)")
    if os.environ.get("MV_BENCH_LAUNCH_CHECK"):
        # test hook (tests/test_bench_cpu.py, no GPU): prove that `--gpus N` ends up as N ranks of one process group, then stop
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = torch.tensor([1.0])
        dist.all_reduce(seen)
        if rank == 0:
            print(json.dumps({"launch_check": True, "rccl_ranks": dist.get_world_size(), "ranks_seen": int(seen.item()), "n_gpus": a.gpus}))
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == a.gpus

    shared = {}
    default_line = a.quantisation_mode is None and a.utterances_per_gpu == 1 and not a.long_form and a.decode_path == "auto"
    out = measure(a, rank, local_rank, world, dist, shared, with_cpu_baseline=(world == 1 and not a.no_cpu_baseline))
    if rank == 0:
        out["rccl_ranks"] = dist.get_world_size() if world > 1 else 1
        if world == 1 and default_line and not a.no_extra_configs:
            extras = {}
            for name, over in EXTRA_CONFIGS:
                b = copy.copy(a)
                for k, v in over.items():
                    setattr(b, k, v)
                t0 = time.perf_counter()
                try:
                    extras[name] = summarise_extra(measure(b, rank, local_rank, world, dist, shared, with_cpu_baseline=False))
                except Exception as e:  # a leg that fails must not take the headline line with it: say so in its place
                    extras[name] = {"error": f"{type(e).__name__}: {e}"[:500]}
                extras[name]["leg_wall_s"] = round(time.perf_counter() - t0, 1)   # incl. weight generation, quantisation and warm-up
            t0 = time.perf_counter()
            try:   # the headline through the drop-in boundary
                extras["configs[1] through TTS.synthesise (the drop-in boundary)"] = boundary_leg(rank, out)
            except Exception as e:
                extras["configs[1] through TTS.synthesise (the drop-in boundary)"] = {"error": f"{type(e).__name__}: {e}"[:500]}
            extras["configs[1] through TTS.synthesise (the drop-in boundary)"]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
            out["extra_configs"] = extras
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
