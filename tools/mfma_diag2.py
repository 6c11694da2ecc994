#!/usr/bin/env python3
"""Free-running reproducibility of decode_mfma.hip: the same batch three times on the matrix-core path; first step at which the captured logits of
two runs differ, per utterance.   usage: mfma_diag2.py [n_layer] [n_new] [len0,len1,len2,len3]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
from oracle import stage1 as o_s1
n_layer = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 600
lens = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [140, 142, 153, 155]
cfgk = dict(n_layer=n_layer, n_head=16, dim=2048, vocab_size=2562, block_size=2048)
sd = o_s1.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
e = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=4)
e.load_state_dict(sd); e.finalize(); del sd
prompts = [mv.synthetic.prompt_tokens(t, seed=u) for u, t in enumerate(lens)]
spks = mv.synthetic.speaker_embedding(0).expand(4, -1)
sps = [mv.SamplingParams(top_p=0.95, guidance_scale=3.0, seed=1 + 7 * u) for u in range(4)]
e.capture_logits(True)
L = max(lens)
runs = []
for mfma in (1, 1, 1, 0, 0):
    e.tune(mfma=mfma, mega=1)
    e.begin(prompts, spks, sps, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
    e.run(poll_every=1 << 20)
    runs.append(([e.tokens(u) for u in range(4)], e.captured_logits(L - 1, n_new - 1).clone()))
    print("path", mfma, e.decode_path(), flush=True)
def first_diff(a, b, u):
    d = (a[1][:, 2 * u:2 * u + 2] != b[1][:, 2 * u:2 * u + 2]).any(dim=2).any(dim=1)
    nz = torch.nonzero(d).flatten()
    return int(nz[0]) if len(nz) else None
for (i, j) in ((0, 1), (1, 2), (3, 4), (0, 3)):
    print(f"runs {i} vs {j}: tokens equal {[runs[i][0][u] == runs[j][0][u] for u in range(4)]}; first step with different logits per utterance {[first_diff(runs[i], runs[j], u) for u in range(4)]}")
a, b = runs[0], runs[1]
for u in range(4):
    s = first_diff(a, b, u)
    if s is not None:
        d = (a[1][s, 2 * u:2 * u + 2] - b[1][s, 2 * u:2 * u + 2]).abs()
        print(f"utt {u}: step {s} (position {L - 1 + s}): {int((d > 0).sum())} logits differ, max {float(d.max()):.4f}; tokens so far equal: {a[0][u][:L + s] == b[0][u][:L + s]}")
