#!/usr/bin/env python3
"""Decode-step time of the full-size engine for the three decode paths: per-layer launches, persistent kernel + sample_kernel
(one step per launch), persistent kernel drawing the tokens itself (mega_fold, `mega_steps` steps per launch).
usage: fold_sweep.py [n_new] ; env FOLD_STEPS="8,64,512" """
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
n_new = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
T = int(os.environ.get("FOLD_PROMPT", "128"))
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
e = mv.Stage1Engine(args, max_utterances=1)
e.load_state_dict(sd); del sd
e.finalize()
prompt, spk, sp = mv.synthetic.prompt_tokens(T, seed=0), mv.synthetic.speaker_embedding(0), mv.SamplingParams(top_p=0.95, guidance_scale=3.0, seed=1)
paths = ([] if os.environ.get("FOLD_SKIP_LAUNCHES") else [("launches", dict(mega=0))]) + [("mega + sample_kernel", dict(mega=1, mega_fold=0))]
paths += [(f"mega_fold, {s} steps/launch", dict(mega=1, mega_fold=1, mega_steps=int(s))) for s in os.environ.get("FOLD_STEPS", "8,64,512").split(",")]
ref = None
for name, knobs in paths:
    e.tune(**knobs)
    e.begin([prompt], spk, sp, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
    steps = e.run(poll_every=1 << 20)
    toks = e.tokens(0)
    if ref is None: ref = toks
    print(f"{name:32s}: {steps} steps, {e.last_run_ms / max(steps, 1) * 1e3:7.1f} us/step, tokens {'identical' if toks == ref else 'DIFFER'}", flush=True)
