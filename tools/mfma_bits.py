#!/usr/bin/env python3
"""Fingerprint of what the matrix-core persistent kernels compute: full-size engines of 4 and 8 utterances, ragged prompts behind the MFMA
prefill, 48 sampled steps with the logits captured -- sha256 of every utterance's token stream and logits.  A change that only touches the
TRANSPORT of the hand-offs (round 6: tag-free rows through LDS-DMA instead of tagged granules) must leave every hash as it was.
usage: mfma_bits.py out.json   (compare two files with `diff`)"""
import hashlib, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
out = {}
for nu in (4, 8):
    e = mv.Stage1Engine(args, max_utterances=nu)
    e.load_state_dict(sd); e.finalize()
    e.capture_logits(True)
    lens = [300, 12, 1030, 255, 40, 700, 17, 1500][:nu]
    prompts = [mv.synthetic.prompt_tokens(t, seed=40 + u) for u, t in enumerate(lens)]
    spk = torch.cat([mv.synthetic.speaker_embedding(50 + u) for u in range(nu)])
    sps = [mv.SamplingParams(top_p=0.95, guidance_scale=3.0, seed=60 + u) for u in range(nu)]
    n_new = 48
    e.begin(prompts, spk, sps, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
    e.run()
    assert e.decode_path() == (3, 0), e.decode_path()
    for u, t in enumerate(lens):
        lg = e.captured_logits(t, n_new - 1)[:, 2 * u:2 * u + 2].contiguous()
        out[f"x{nu} utt{u} T{t}"] = dict(tokens=hashlib.sha256(str(e.tokens(u)).encode()).hexdigest()[:16],
                                          logits=hashlib.sha256(lg.numpy().tobytes()).hexdigest()[:16])
    e.close()
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
