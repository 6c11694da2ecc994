#!/usr/bin/env python3
"""Decode-step time of the full-size engine with the token drawn inside the persistent launch (mega_fold 1) vs one step per launch +
sample_kernel (mega_fold 0), for engines of 1 / 2 / 4 utterances and the three weight formats; token streams must be identical.
usage: nu_sweep.py [n_new]   env NU_CASES="bf16:2,bf16:4,int8:1,int4:1" """
import hashlib, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
n_new = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
for case in os.environ.get("NU_CASES", "bf16:2,bf16:4,int8:1,int4:1").split(","):
    fmt, nu = case.split(":")
    nu = int(nu)
    e = mv.Stage1Engine(args, max_utterances=nu, quantisation_mode=None if fmt == "bf16" else fmt)
    e.load_state_dict(sd)
    e.finalize()
    g = torch.Generator().manual_seed(0)
    lens = [128] if nu == 1 else torch.randint(96, 161, (nu,), generator=g).tolist()
    prompts = [mv.synthetic.prompt_tokens(t, seed=u) for u, t in enumerate(lens)]
    spk = mv.synthetic.speaker_embedding(0).expand(nu, -1)
    sps = [mv.SamplingParams(top_p=0.95, guidance_scale=3.0, seed=1 + 7 * u) for u in range(nu)]
    ref = None
    for fold in (1, 0, 1):
        e.tune(mega=1, mega_fold=fold, mfma=int(os.environ.get("NU_MFMA", "1")))   # NU_MFMA=0: engines of 4 utterances on decode_mega.hip's instance
        e.begin(prompts, spk, sps, max_new_tokens=min(n_new, args.block_size - max(lens)), end_of_audio_token=9999, prefill_mode=1)
        steps = e.run(poll_every=1 << 20)
        toks = [e.tokens(u) for u in range(nu)]
        ref = ref or toks
        print(f"{fmt} x {nu} utterance(s), mega_fold={fold}: path {e.decode_path()}, {steps} steps, {e.last_run_ms / max(steps, 1) * 1e3:7.1f} us/step, "
              f"{nu * steps / e.last_run_ms * 1e3:7.0f} tok/s, tokens {'identical' if toks == ref else 'DIFFER'} sha1 {hashlib.sha1(repr(toks).encode()).hexdigest()[:12]}", flush=True)
    e.close()
