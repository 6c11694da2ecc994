#!/usr/bin/env python3
"""Launch each decode-step kernel kind a few dozen times on a full-size synthetic MetaVoice-1B engine (mv_probe_kernel).
Small enough to sit under `rocprofv3 --pmc ...` (counter passes serialise every dispatch)."""
import argparse, ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=48)
ap.add_argument("--kinds", default="0,1,2,3,4,5")
ap.add_argument("--pos", type=int, default=1024, help="context length the attention probe runs at")
ap.add_argument("--ablate", default="0", help="comma list of gemv_ablate masks (1 no weight loads, 2 no dots, 4 no prologue, 8 no reductions)")
a = ap.parse_args()
mv = importlib.import_module("metavoice-src_amd")
N = importlib.import_module("metavoice-src_amd._native")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
e = mv.Stage1Engine(args, max_utterances=1)
e.load_state_dict(sd); del sd
e.finalize()
e.begin([mv.synthetic.prompt_tokens(a.pos, seed=0)], mv.synthetic.speaker_embedding(0), mv.SamplingParams(top_p=0.95),
        max_new_tokens=4, end_of_audio_token=9999)
e.run(max_steps=a.pos)  # teacher-forced prompt: fills the KV cache up to `pos`
# kind 6 = the persistent decode-step kernel (decode_mega.hip), one launch = 24 layers + LM head at context `pos`
names = ["norm+wqkv+kv", "attention", "wo+res", "norm+w1|w3+silu", "w2+res", "norm+head", "persistent step"]
for ab in [int(x, 0) for x in a.ablate.split(",")]:
  e.tune(gemv_ablate=ab)
  if ab: print(f"--- gemv_ablate = {ab}")
  for k in [int(x, 0) for x in a.kinds.split(",")]:
    ms, nb = C.c_float(0), C.c_double(0)
    N.check(N.lib().mv_probe_kernel(e._h, k, a.iters, C.byref(ms), C.byref(nb)))
    mode = "warm" if k & 0x100 else "cold"
    print("  " + f"kind {k & 0xFF} {names[k & 0xFF]:18s} {mode:16s} {ms.value*1e3:8.2f} us  {nb.value/1e6:8.2f} MB  {nb.value/(ms.value*1e-3)/1e9:8.1f} GB/s", flush=True)
