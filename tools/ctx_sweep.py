#!/usr/bin/env python3
"""Decode-step time of the full-size engine at fixed contexts: every utterance starts from a prompt of P tokens (MFMA prefill) and runs
`n_steps` decode steps.  usage: ctx_sweep.py [n_steps]   env CTX="200,520,1024,1600" NU_CASES="bf16:1,bf16:4" (MV_HIP_LIB selects the library)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
for case in os.environ.get("NU_CASES", "bf16:1,bf16:4").split(","):
    fmt, nu = case.split(":")
    nu = int(nu)
    e = mv.Stage1Engine(args, max_utterances=nu, quantisation_mode=None if fmt == "bf16" else fmt)
    e.load_state_dict(sd)
    e.finalize()
    spk = mv.synthetic.speaker_embedding(0).expand(nu, -1)
    out = []
    for P in [int(x) for x in os.environ.get("CTX", "200,520,1024,1600").split(",")]:
        prompts = [mv.synthetic.prompt_tokens(P, seed=u) for u in range(nu)]
        sps = [mv.SamplingParams(top_p=0.95, guidance_scale=3.0, seed=1 + 7 * u) for u in range(nu)]
        best = None
        for rep in range(2):
            e.begin(prompts, spk, sps, max_new_tokens=n_steps + 1, end_of_audio_token=9999, prefill_mode=1)
            steps = e.run(poll_every=1 << 20)
            t = e.last_run_ms / max(steps, 1) * 1e3
            best = t if best is None else min(best, t)
        out.append(f"ctx {P}: {best:7.1f}")
    print(f"{fmt} x {nu}: path {e.decode_path()} us/step at " + "; ".join(out), flush=True)
    e.close()
