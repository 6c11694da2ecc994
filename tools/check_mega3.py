#!/usr/bin/env python3
"""Persistent kernel v3 (loader waves + LDS ring, mv_tune mega=2) against the per-layer launches: identical tokens and logits."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
args = mv.ModelArgs.from_name("metavoice-1B")
for n_layer, T, n_new in [(2, 16, 24), (3, 300, 40), (24, 128, 96), (24, 1500, 64)]:
    cfgk = dict(n_layer=n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
    a = mv.ModelArgs.from_name("metavoice-1B"); a.n_layer = n_layer
    sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=1, device="cuda", **cfgk))
    e = mv.Stage1Engine(a, max_utterances=1)
    e.load_state_dict(sd); del sd
    e.finalize()
    e.capture_logits(True)
    out = {}
    for mega in (0, 2, 1, 2):
        e.tune(mega=mega)
        e.begin([mv.synthetic.prompt_tokens(T, seed=2)], mv.synthetic.speaker_embedding(2), mv.SamplingParams(top_p=0.95, seed=3),
                max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        t0 = time.time(); e.run(); dt = time.time() - t0
        res = (e.tokens(0), e.captured_logits(T - 1, n_new).clone())
        if mega in out:
            assert res[0] == out[mega][0] and torch.equal(res[1], out[mega][1]), "v3 not reproducible run to run"
        out[mega] = res
        print(f"n_layer={n_layer} T={T} mega={mega}: {e.last_run_ms / n_new * 1e3:.1f} us/step", flush=True)
    for m in (1, 2):
        same_t = out[m][0] == out[0][0]
        dmax = float((out[m][1].float() - out[0][1].float()).abs().max())
        print(f"  mega={m} vs launches: tokens identical={same_t}, max|dlogit|={dmax}")
        assert same_t and dmax == 0.0
    e.tune(mega=1)
    e.close()
print("OK")
