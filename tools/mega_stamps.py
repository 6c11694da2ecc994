#!/usr/bin/env python3
"""Phase timeline of the persistent decode-step kernel (MV_MEGA_STAMPS=1): per layer, for the first and the last workgroup,
how long wave 0 waited for each hand-off and how long it computed.  Slots: 0 layer start, 1 x ready, 2 qkv published,
3 q/k/v ready, 4 attention done (P2 start), 5 attn vector ready, 6 P3 start, 7 h ready, 8 P4 start, 9 act ready, 10 x published."""
import ctypes as C, importlib, os, sys
os.environ["MV_MEGA_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
mv = importlib.import_module("metavoice-src_amd")
N = importlib.import_module("metavoice-src_amd._native")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
pos = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
e = mv.Stage1Engine(args, max_utterances=1)
e.load_state_dict(sd); del sd
e.finalize()
e.tune(mega=int(os.environ.get("MV_MEGA", "1")))
e.begin([mv.synthetic.prompt_tokens(pos, seed=0)], mv.synthetic.speaker_embedding(0), mv.SamplingParams(top_p=0.95), max_new_tokens=64,
        end_of_audio_token=9999, prefill_mode=1)
e.run(max_steps=32)
print(f"32 steps at pos~{pos}: {e.last_run_ms / 32 * 1e3:.1f} us/step")
L = args.n_layer
buf = np.zeros(2 * L * 12, dtype=np.uint64)
N.check(N.lib().mv_debug_stamps(e._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size))
st = buf.reshape(2, L, 12).astype(np.int64)
names = ["wait x", "norm+qkv", "wait qkv", "attention", "wait attn", "wo -> P3", "wait h", "norm+w13 -> P4", "wait act", "w2+publish"]
for blk in range(2):
    d = np.diff(st[blk, :, :11], axis=1)[2:-2]  # skip the first/last layers
    per = d.mean(axis=0)
    tot = (st[blk, 1:, 0] - st[blk, :-1, 0])[2:-2].mean()
    print(f"workgroup {'first' if blk == 0 else 'last'}: {tot:.0f} ticks per layer; per segment (ticks): " +
          ", ".join(f"{n} {v:.0f}" for n, v in zip(names, per)))
print("(s_memtime ticks; 100 MHz constant clock if the numbers are ~100x smaller than cycles)")
