#!/usr/bin/env python3
"""Diagnostic for decode_mfma.hip: teacher-forced logits of a 4-utterance engine on the matrix-core path (twice) and on the per-layer launches,
per utterance and position: is the path reproducible, and how far is it from the launches?   usage: mfma_diag.py [n_layer] [len0,len1,len2,len3]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
from oracle import stage1 as o_s1
n_layer = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lens = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [40, 23, 31, 17]
cfgk = dict(n_layer=n_layer, n_head=16, dim=2048, vocab_size=2562, block_size=2048)
sd = o_s1.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=17, **cfgk))
e = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=4)
e.load_state_dict(sd); e.finalize()
prompts = [mv.synthetic.prompt_tokens(t, seed=90 + u) for u, t in enumerate(lens)]
spks = torch.cat([mv.synthetic.speaker_embedding(95 + u) for u in range(4)])
sps = [mv.SamplingParams(guidance_scale=3.0, seed=70 + u, top_k=1) for u in range(4)]
e.capture_logits(True)
L = max(lens)
runs = []
for mfma, mega in ((1, 1), (1, 1), (0, 0), (0, 1)):
    e.tune(mfma=mfma, mega=mega)
    e.begin(prompts, spks, sps, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=0)
    e.run()
    runs.append(e.captured_logits(0, L).clone())
    print("path", (mfma, mega), e.decode_path(), flush=True)
for u, t in enumerate(lens):
    a, b, c, d = (r[:t, 2 * u:2 * u + 2] for r in runs)
    rr = (a - b).abs().amax(dim=(1, 2))       # run to run, per position
    dl = (a - c).abs().amax(dim=(1, 2))       # vs launches
    dm = (d - c).abs().amax(dim=(1, 2))       # decode_mega instance vs launches (must be 0)
    bad = [int(i) for i in torch.nonzero(rr > 0).flatten()[:12]]
    print(f"utt {u} (len {t}): logit std {float(c.std()):.3f}; run-to-run max {float(rr.max()):.4f} first nonzero positions {bad}; vs launches max {float(dl.max()):.4f} "
          f"at {int(dl.argmax())} (median {float(dl.median()):.4f}); decode_mega vs launches {float(dm.max()):.4f}")
    print("   vs launches per position:", " ".join(f"{float(x):.3f}" for x in dl[:48]))
