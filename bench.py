#!/usr/bin/env python3
"""bench.py -- audio-seconds per wall-second (RTF^-1) of the MI355X-native MetaVoice-1B engine.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one utterance per GPU: BASELINE.json configs[1] -- MetaVoice-1B bf16, one
utterance, synthetic 256-char-equivalent prompt (T = 128 text tokens), batched MFMA prefill, N = 1920 generated
first-stage tokens forced (end-of-audio unreachable) = 12.8 s of audio at 150 tokens/s, token adapter,
second-stage expansion to 8 codebooks (placeholder second-stage sizes: the real model_args are not reachable offline),
EnCodec decode (8 codebooks -> 24 kHz waveform), random-init weights of the exact MetaVoice-1B first-stage and EnCodec
shapes.  The reference's multi-band-diffusion refinement (audiocraft, third party) is not built yet and not timed.  Inputs (weights, prompt, speaker embedding) are resident
before the timed region; the timed region is bracketed by barrier + torch.cuda.synchronize on both sides and the MAX
over ranks is reported.  Utterances are independent, so N GPUs run N utterances with no data-path collective; the
finished token streams are all-gathered over RCCL once per step (batched-serving path), inside the timed region.

Extra objects on the JSON line:
  roofline      -- dominant kernel: the persistent decode-step kernel (one launch per token) where supported, else the
                   largest per-layer kernel of the decode step (fused RMSNorm + w1|w3 + SiLU-gate GEMV, 46 MB of weights per
                   launch): algorithmic bytes / average launch duration measured with HIP events on the engine stream.
  step_roofline -- the same ratio for a whole decode step (all ~122 launches), the number RTF actually follows.
  cpu_baseline  -- the CPU oracle (a torch-CPU fp32 restatement of the reference, golden-pinned to it) timed on this
                   host's cores on a bounded sample, rank 0, N=1 only.
"""
import argparse
import importlib
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


def step_bytes(cfg, pos):
    """algorithmic bytes of one decode step at input position `pos` (bf16, one utterance = 2 CFG rows).
    The speaker projection is hoisted out of the step (it is step-invariant), so its 1 MB is not counted."""
    per_layer = (3 * cfg.dim * cfg.dim + cfg.dim * cfg.dim + 3 * cfg.dim * cfg.intermediate_size + 2 * cfg.dim)
    weights = 2 * (cfg.n_layer * per_layer + cfg.vocab_size * cfg.dim + cfg.dim)
    kv = 2 * 2 * cfg.n_head * 128 * 2 * cfg.n_layer * (pos + 1)
    return weights + kv


def cpu_baseline(mv, cfgk, sample_steps=8):
    """Time the oracle (oracle/stage1.py, fp32, torch CPU) on the same workload shape: bounded sample."""
    from oracle import stage1 as o_s1

    sd = o_s1.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, **cfgk))
    cfg = o_s1.Stage1Config(**cfgk)
    orc = o_s1.Stage1Oracle({k: v.float() for k, v in sd.items()}, cfg, rounding="fp32", max_seq_length=T_PROMPT + sample_steps + 8)
    del sd
    prompt = mv.synthetic.prompt_tokens(T_PROMPT, seed=0)
    spk = mv.synthetic.speaker_embedding(0)
    t0 = time.perf_counter()
    logits = orc.forward(prompt.view(1, -1).repeat(2, 1), spk, torch.arange(T_PROMPT))
    tok = o_s1.sample(logits, 3.0, 1.0, top_k=1)[0]
    t1 = time.perf_counter()
    for i in range(sample_steps):
        logits = orc.forward(torch.full((2, 1), tok), spk, torch.tensor([T_PROMPT + i]))
        tok = o_s1.sample(logits, 3.0, 1.0, top_k=1)[0]
    t2 = time.perf_counter()
    tok_s = sample_steps / (t2 - t1)
    return {
        "value": round(tok_s / TOKENS_PER_AUDIO_SECOND, 5), "unit": "audio-s/s", "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"oracle/stage1.py fp32 eager torch-CPU, full-size synthetic MetaVoice-1B: prefill T={T_PROMPT} "
                  f"({t1 - t0:.2f} s) + {sample_steps} greedy decode steps ({(t2 - t1) / sample_steps:.3f} s/token, "
                  f"{tok_s:.2f} tok/s); value = decode tok/s / 150; host has {os.cpu_count()} logical cores",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    mv = importlib.import_module("metavoice-src_amd")
    args = mv.ModelArgs.from_name("metavoice-1B")
    cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)

    # ---- resident inputs: weights drawn on the GPU, copied into the engine's arena, then dropped -------------------
    fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
    sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=rank, device="cuda", **cfgk))
    engine = mv.Stage1Engine(args, max_utterances=1)
    engine.load_state_dict(sd)
    del sd
    torch.cuda.empty_cache()
    engine.finalize()
    prompt = mv.synthetic.prompt_tokens(T_PROMPT, seed=rank)
    spk = mv.synthetic.speaker_embedding(rank)
    sp = mv.SamplingParams(guidance_scale=3.0, temperature=1.0, top_p=0.95, seed=1337 + rank)  # TTS.synthesise defaults
    dispatch = importlib.import_module("metavoice-src_amd.batch_dispatch")

    # second stage (fam/llm/inference.py Model + model.py GPT, non-causal): placeholder sizes, block_size 2048 so that the
    # 960 frames + 128 text ids of this workload fit untruncated
    adapters = importlib.import_module("metavoice-src_amd.adapters")
    inference = importlib.import_module("metavoice-src_amd.inference")
    s2_args = mv.synthetic.stage2_model_args(block_size=2048)
    stage2 = mv.GPT(mv.GPTConfig(**s2_args), speaker_emb_dim=256, max_batch=1)
    stage2.load_state_dict(mv.synthetic.stage2_state_dict(s2_args, speaker_emb_dim=256, seed=100 + rank))
    first_adapter = adapters.FlattenedInterleavedEncodec2Codebook(end_of_audio_token=1024)
    second_adapter = adapters.TiltedEncodec(end_of_audio_token=1024)
    text2 = [int(t) - 1024 for t in prompt.tolist()]  # the same text in the second stage's id space (offset 1025)
    # codec stage: EnCodec's own decoder (8 codebooks -> 24 kHz waveform) in HIP; the reference's multi-band diffusion
    # refinement on top of it is not built yet and is NOT in this number
    codec_mod = importlib.import_module("metavoice-src_amd.codec")
    vocoder = codec_mod.EncodecDecoder.from_state_dict(mv.synthetic.encodec_state_dict(seed=rank))
    stage_ms = {"stage1_prefill+decode": 0.0, "stage1_decode_gpu": 0.0, "adapters+stage2": 0.0, "encodec_decode": 0.0}

    def one_step():
        t0 = time.perf_counter()
        engine.begin([prompt], spk, sp, max_new_tokens=N_NEW, end_of_audio_token=9999, prefill_mode=1)
        engine.run(poll_every=1 << 20)
        toks = engine.tokens(0)
        t1 = time.perf_counter()
        _, cbs = first_adapter.decode([toks])
        grid = inference.pack_second_stage_input(text2, cbs, s2_args["block_size"])
        in_x = torch.tensor([grid], dtype=torch.long)
        y = stage2.generate(in_x, None, temperature=1.0, top_k=200, speaker_embs=spk, batch_size=1, seed=sp.seed)
        _, audio_ids = second_adapter.decode(torch.cat([in_x, y], dim=1)[0].tolist())
        t2 = time.perf_counter()
        wav = vocoder.decode(torch.tensor(audio_ids, dtype=torch.long).unsqueeze(0))
        t3 = time.perf_counter()
        stage_ms["encodec_decode"] += (t3 - t2) * 1e3
        stage_ms["stage1_prefill+decode"] += (t1 - t0) * 1e3
        stage_ms["stage1_decode_gpu"] += engine.last_run_ms
        stage_ms["adapters+stage2"] += (t2 - t1) * 1e3
        if world > 1:  # batched-serving path: every rank ends up with every utterance's (8, f) token grid, in input order
            mine = torch.tensor(audio_ids, dtype=torch.int32, device="cuda").flatten()
            everyone = dispatch.gather_variable([mine], world, device=torch.device("cuda", local_rank))
            assert len(everyone) == world
        return toks, engine.last_run_ms

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
        toks, ms = one_step()
        gpu_ms += ms
        n_tok += len(toks) - T_PROMPT
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
        # the decode step is ONE persistent launch per token (decode_mega.hip: 24 layers + LM head) when the configuration
        # supports it; probe 6 times that kernel alone at the position the utterance ended on.  Otherwise (per-layer launches)
        # the dominant kernel is the fused RMSNorm + w1|w3 + SiLU-gate GEMV (probe 3).
        # The kernel's duration grows linearly with the context; it is probed at the MEAN context of the timed utterance, so that
        # `avg_launch_us` is the average launch of the timed region (what the rocprofv3 --stats average in profiles/ shows) and
        # `bytes_per_launch` the average algorithmic bytes of those launches.
        mean_ctx = T_PROMPT + (N_NEW - 1) // 2
        engine.begin([mv.synthetic.prompt_tokens(mean_ctx, seed=rank)], spk, sp, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=1)
        engine.run(max_steps=1)
        if N.lib().mv_probe_kernel(engine._h, 6, 200, C.byref(ms), C.byref(nbytes)) == 0:
            kname = "decode_mega_kernel (persistent: 24 layers + LM head of one token; probed at context %d = the mean of the utterance)" % (mean_ctx + 1)
        else:
            N.check(N.lib().mv_probe_kernel(engine._h, 3, 24 * 20, C.byref(ms), C.byref(nbytes)))
            kname = "gemv_kernel<2,4,1,SWIGLU,nt> (RMSNorm + w1|w3 + SiLU gate)"
        ach = nbytes.value / (ms.value * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kname,
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": None, "bytes_per_launch": nbytes.value, "avg_launch_us": round(ms.value * 1e3, 2)}
        # HBM traffic per launch comes from a separate rocprofv3 --pmc FETCH_SIZE pass (it cannot be collected from inside this
        # process); the committed summary of the last pass is quoted with the launches it was measured on
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "roofline_traffic.json")) as f:
                tr = json.load(f)
            if tr["kernel"] in kname:
                roofline["traffic"] = tr["traffic_bytes_per_launch"]
                roofline["traffic_note"] = (f"{tr['source']}; {tr['launch_contexts']}; algorithmic bytes of those launches "
                                            f"{tr['algorithmic_bytes_same_launches']} (ratio {tr['ratio']})")
        except (OSError, KeyError, ValueError):
            pass
        per_kernel = {}
        for kind, name in [(0, "norm+wqkv+kv"), (1, "attention@pos"), (2, "wo+res"), (4, "w2+res"), (5, "norm+head")]:
            N.check(N.lib().mv_probe_kernel(engine._h, kind, 24 * 10, C.byref(ms), C.byref(nbytes)))
            per_kernel[name] = {"us": round(ms.value * 1e3, 2), "GB/s": round(nbytes.value / (ms.value * 1e-3) / 1e9, 1)}
        total_bytes = sum(step_bytes(args, p) for p in range(T_PROMPT, T_PROMPT + N_NEW - 1))  # the decode steps mv_run timed
        step_frac = total_bytes / ((gpu_ms / a.steps) * 1e-3) / 1e9 / HBM_PEAK_GBS
        out = {
            "metric": "audio-sec/sec (RTF^-1) MetaVoice-1B bf16, 256-char prompt",
            "value": round(value, 4), "unit": "audio-s/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"MetaVoice-1B synthesis, 1 utterance/GPU: prompt T={T_PROMPT} tokens (MFMA prefill) + {N_NEW} "
                                   "sampled first-stage tokens (top_p=0.95, guidance 3.0) = 12.8 s audio, token adapters, "
                                   "second-stage expansion to 8 codebooks (placeholder sizes), EnCodec decode "
                                   "to a 24 kHz waveform; the multi-band-diffusion refinement is not built yet",
                       "rows_per_gpu": 2, "tokens_per_step": N_NEW, "parallelism": f"utterance-dp{world}"},
            "tokens_per_s_per_gpu": round((N_NEW - 1) / (gpu_ms / a.steps * 1e-3), 1),
            "stage_ms_per_utterance": {k: round(v / a.steps, 2) for k, v in stage_ms.items()},
            "roofline": roofline,
            "step_roofline": {"bound": "hbm", "achieved": round(step_frac * HBM_PEAK_GBS, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(step_frac, 4), "scope": "the 1919 decode steps (HIP-event time of mv_run)",
                              "algorithmic_bytes_per_utterance": total_bytes,
                              "gpu_ms_per_utterance": round(gpu_ms / a.steps, 2)},
            "per_kernel": per_kernel,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mv, cfgk)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
