import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
for n_layer in (2, 3):
    cfgk = dict(n_layer=n_layer, n_head=16, dim=2048, vocab_size=2562, block_size=2048)
    sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=7, **cfgk))
    e = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=1)
    e.load_state_dict(sd); e.finalize()
    T, n_new = 16, 120
    prompt, spk = mv.synthetic.prompt_tokens(T, seed=7), mv.synthetic.speaker_embedding(7)
    e.capture_logits(True)
    out = {}
    for mega in (1, 0):
        e.tune(mega=mega)
        e.begin([prompt], spk, mv.SamplingParams(top_k=1, seed=21), max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        e.run()
        out[mega] = (e.tokens(0), e.captured_logits(T - 1, n_new))
    for first, second in ((1, 0), (0, 1)):
        e.tune(mega=first)
        e.begin([prompt], spk, mv.SamplingParams(top_k=1, seed=21), max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        e.run(max_steps=1)
        e.tune(mega=second)
        e.run(max_steps=3)
        lg = e.captured_logits(T - 1, 5)
        dd = (lg - out[0][1][:5]).abs().amax(dim=(1, 2))
        print(f"  first step mega={first}, then mega={second}: per-step max diff vs launches {[round(float(x), 3) for x in dd]}")
    from oracle import stage1 as o_s1
    cfg = o_s1.Stage1Config(**cfgk)
    orc = o_s1.Stage1Oracle({k: v.float() for k, v in sd.items()}, cfg, rounding="bf16", max_seq_length=64)
    oseq, ol = o_s1.generate(orc, prompt, spk, max_new_tokens=n_new, end_of_audio_token=9999, top_k=1, seed=21, return_logits=True)
    for mega in (1, 0):
        dd = (out[mega][1] - ol).abs().amax(dim=(1, 2))
        print(f"  mega={mega} vs oracle per-step max diff: {[round(float(x), 3) for x in dd]}  tokens equal oracle: {out[mega][0] == oseq}")
    d = (out[1][1] - out[0][1]).abs()
    print(f"n_layer={n_layer} tokens equal: {out[1][0] == out[0][0]}")
    for s in range(n_new):
        nz = int((d[s] > 0).sum())
        print(f"  step {s}: max diff {float(d[s].max()):.5f}  differing logits {nz} / {d[s].numel()}  rows {[int((d[s][r] > 0).sum()) for r in range(2)]}")
    e.close()
