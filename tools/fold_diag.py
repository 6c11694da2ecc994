#!/usr/bin/env python3
"""Diagnostic: which decode path survives (no abandoned persistent launch) for a small and the full-size model."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
def run(n_layer, cases, steps=40):
    args = mv.ModelArgs.from_name("metavoice-1B")
    cfgk = dict(n_layer=n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
    sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
    a2 = mv.ModelArgs(**cfgk)
    e = mv.Stage1Engine(a2, max_utterances=1)
    e.load_state_dict(sd); del sd
    e.finalize()
    prompt, spk, sp = mv.synthetic.prompt_tokens(int(os.environ.get("DIAG_PROMPT", "128")), seed=0), mv.synthetic.speaker_embedding(0), mv.SamplingParams(top_p=0.95, guidance_scale=3.0, seed=1)
    for name, knobs in cases:
        e.tune(mega=1)   # clears a sticky fall-back
        e.tune(**knobs)
        t0 = e.decode_path()[1]
        e.begin([prompt], spk, sp, max_new_tokens=steps + 8, end_of_audio_token=9999, prefill_mode=1)
        n = e.run(max_steps=steps, poll_every=8)
        print(f"layers {n_layer:2d} {name:28s}: {n} steps {e.last_run_ms / max(n, 1) * 1e3:8.1f} us/step, path {e.decode_path()[0]}, new timeouts {e.decode_path()[1] - t0}", flush=True)
    e.close()
cases = [("mega+sampler", dict(mega_fold=0, gemv_ablate=0)), 
         ("fold 8/launch", dict(mega_fold=1, mega_steps=8, gemv_ablate=0)), ("fold 1/launch", dict(mega_fold=1, mega_steps=1, gemv_ablate=0))]
cases = cases[: int(os.environ.get("DIAG_CASES", "4"))]
for nl in [int(x) for x in os.environ.get("DIAG_LAYERS", "2,6,24").split(",")]:
    run(nl, cases)
