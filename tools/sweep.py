#!/usr/bin/env python3
"""Tuning sweep on the GPU: one full-size synthetic MetaVoice-1B engine, one 2047-step utterance per knob setting
(mv_tune), HIP-event time of the run.  Prints a table; the winners become the defaults in kernels.h (MvTune)."""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="", help="JSON list of {knob: value} dicts; each is applied on top of the defaults")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--new", type=int, default=1920)
ap.add_argument("--utterances", type=int, default=1, help="utterances decoded together on this GPU (2 CFG rows each)")
ap.add_argument("--quant", default=None, help="int4 / int8: run the weight-only quantised engine (BASELINE config 4)")
a = ap.parse_args()
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
e = mv.Stage1Engine(args, max_utterances=a.utterances, quantisation_mode=a.quant)
e.load_state_dict(sd); del sd
e.finalize()
print(f"model bytes {e.model_bytes / 1e9:.3f} GB ({a.quant or 'bf16'})")
U = a.utterances
prompt = [mv.synthetic.prompt_tokens(128, seed=u) for u in range(U)]
spk = torch.cat([mv.synthetic.speaker_embedding(u) for u in range(U)])
sp = [mv.SamplingParams(top_p=0.95, seed=7 + u) for u in range(U)]
DEFAULTS = dict(gemv_nt=1, use_graph=1, mega=1, gemv_ablate=0,
                gemv_threads_plain=0, gemv_threads_resid=0, gemv_threads_swiglu=0, gemv_threads_qkv=0,
                gemv_nr_plain=0, gemv_nr_resid=0, gemv_nr_swiglu=0, gemv_nr_qkv=0)
configs = json.loads(a.configs) if a.configs else [{}]
ref_tokens = None
for cfg in configs:
    knobs = dict(DEFAULTS); knobs.update(cfg)
    e.tune(**knobs)
    best = 1e9
    for _ in range(a.reps):
        e.begin(prompt, spk, sp, max_new_tokens=a.new, end_of_audio_token=9999)
        steps = e.run(poll_every=1 << 20)
        best = min(best, e.last_run_ms)
    toks = e.tokens(0)
    if ref_tokens is None:
        ref_tokens = toks
    same = toks == ref_tokens  # knobs must never change the result
    print(f"{best:9.2f} ms  {U * steps / best * 1e3:8.1f} tok/s  {best / steps * 1e3:7.1f} us/step  same_tokens={same}  {json.dumps(cfg)}", flush=True)
