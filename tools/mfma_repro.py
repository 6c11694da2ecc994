#!/usr/bin/env python3
"""Reproducibility stress of the matrix-core persistent kernels: the same teacher-forced batch REPS times on one engine; every run's logits must
equal the first run's bit for bit.  Prints where a run differs (utterance, positions, size of the difference) -- how a hand-off race shows.
usage: mfma_repro.py [n_utt] [n_layer] [reps]     REPRO_EXP=1000: the skew hook of decode_mfma.hip (mv_tune mega_fault 1000)     REPRO_ALTERNATE=1: a run on the per-layer launches between the runs, as the test does"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_layer = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfgk = dict(n_layer=n_layer, n_head=16, dim=2048, vocab_size=2562, block_size=2048)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=17, **cfgk))
e = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=nu)
e.load_state_dict(sd); e.finalize()
e.capture_logits(True)
if os.environ.get("REPRO_EXP"):
    e.tune(mega_fault=int(os.environ["REPRO_EXP"]))
lens = [300, 12, 2000, 255, 40, 700, 17, 1500][:nu]
prompts = [mv.synthetic.prompt_tokens(t, seed=90 + u) for u, t in enumerate(lens)]
spk = torch.cat([mv.synthetic.speaker_embedding(95 + u) for u in range(nu)])
sps = [mv.SamplingParams(guidance_scale=3.0, seed=70 + u, top_k=1) for u in range(nu)]
L = max(lens)
first = None
bad = 0
timeouts = 0
for rep in range(reps):
    if os.environ.get("REPRO_ALTERNATE") and rep % 2 == 1:
        e.tune(mega=0)
        e.begin(prompts, spk, sps, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=0); e.run()
        e.tune(mega=1)
    e.begin(prompts, spk, sps, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=0)
    e.run()
    if e.decode_path()[0] != 3:   # a hand-off timed out and the run fell back to the launches: count it, re-arm, do not compare this run
        timeouts += 1
        print(f"run {rep}: hand-off time-out, decode_path {e.decode_path()}", flush=True)
        e.tune(mega=1)
        continue
    lg = e.captured_logits(0, L)
    if first is None:
        first = lg
        continue
    for u, t in enumerate(lens):
        a, b = lg[:t, 2 * u:2 * u + 2], first[:t, 2 * u:2 * u + 2]
        if not torch.equal(a, b):
            bad += 1
            d = (a - b).abs().amax(dim=(1, 2))
            pos = torch.nonzero(d > 0).flatten()
            print(f"run {rep}: utterance {u} (prompt {t}) differs at {len(pos)} positions, first {pos[:8].tolist()}, max |d| {float(d.max()):.4f} (logit std {float(b.std()):.3f}), "
                  f"NaNs {int(torch.isnan(a).sum())}", flush=True)
print(f"{nu} utterances, {n_layer} layers, {reps} runs: {bad} (run, utterance) pairs differ, {timeouts} runs with a hand-off time-out")
