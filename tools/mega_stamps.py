#!/usr/bin/env python3
"""Phase timeline of the persistent decode-step kernel (MV_MEGA_STAMPS=1): per layer, for the first and the last workgroup,
how long wave 0 waited for each hand-off and how long it computed.  Slots: 0 layer start, 1 x ready, 2 qkv published,
3 q/k/v ready, 4 attention done (P2 start), 5 attn vector ready, 6 P3 start, 7 h ready, 8 P4 start, 9 act ready, 10 x published."""
import ctypes as C, importlib, os, sys
os.environ["MV_MEGA_STAMPS"] = "1"
os.environ.setdefault("MV_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metavoice-src_amd", "lib", "libmetavoice_hip_instr.so"))  # stamps live in the instrumented build (make -C csrc instr)
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
NU = int(os.environ.get("MV_UTT", "1"))
e = mv.Stage1Engine(args, max_utterances=NU, quantisation_mode=os.environ.get("MV_QUANT") or None)
e.load_state_dict(sd); del sd
e.finalize()
e.tune(mega=int(os.environ.get("MV_MEGA", "1")), gemv_ablate=int(os.environ.get("MV_ABLATE", "0")))
e.begin([mv.synthetic.prompt_tokens(pos, seed=u) for u in range(NU)], mv.synthetic.speaker_embedding(0).expand(NU, -1), mv.SamplingParams(top_p=0.95), max_new_tokens=64,
        end_of_audio_token=9999, prefill_mode=1)
e.run(max_steps=32)
print(f"32 steps at pos~{pos}: {e.last_run_ms / 32 * 1e3:.1f} us/step")
L = args.n_layer
S = 24
buf = np.zeros(2 * L * S, dtype=np.uint64)
N.check(N.lib().mv_debug_stamps(e._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size))
st = buf.reshape(2, L, S).astype(np.int64)
TICK_US = 1 / 2340.0  # s_memtime ticks per us on this part (measured against HIP events)
# coarse segments (slots 0..10) and, for the register-prefetch kernel, the fine order of slots inside a layer
if NU >= 4 and e.decode_path()[0] == 3:   # the matrix-core instance (decode_mfma.hip): its own slot map
    mf_names = ["x ready (poll)", "norm", "wqkv MFMAs + exchange", "reduce + q/k/v published", "q/k/v ready (V rows, wo requested; poll)", "key ranges scored",
                "merge + partial / output published", "attention vector ready (w1 requested; combine + poll)", "wo MFMAs + reduce + h published", "h ready (w3 requested; poll)",
                "norm", "w1|w3 MFMAs + exchange", "reduce + act published", "act ready (w2 requested; poll; 8 utterances: both halves)", "w2 MFMAs + exchange", "reduce + x published"]
    for blk in range(2):
        seq = st[blk][:, :17][2:-2]
        dd = np.diff(seq, axis=1).mean(axis=0) * TICK_US
        tot = (st[blk, 1:, 0] - st[blk, :-1, 0])[2:-2].mean() * TICK_US
        nxt = tot - dd.sum()
        print(f"decode_mfma_kernel<{NU}>, workgroup {'first' if blk == 0 else 'last'}: {tot:.1f} us per layer")
        for nm, v in zip(mf_names, dd):
            print(f"    {v:6.2f}  {nm}")
        print(f"    {nxt:6.2f}  next layer top (next wqkv requested)")
        tr = st[blk][2:-2, 17:22].astype(np.float64)
        print("    attempts of each row fetch that came back with a sentinel (wave 0; mean over layers, max): " +
              ", ".join(f"{h} {tr[:, k].mean():.2f} ({int(tr[:, k].max())})" for k, h in enumerate(["x", "attention vector", "h", "act", "act 2nd half"])))
    sys.exit(0)
names = ["wait x", "norm+qkv", "wait qkv", "attention", "wait attn", "wo -> P3", "wait h", "norm+w13 -> P4", "wait act", "w2+publish"]
fine_order = [0, 1, 11, 12, 13, 2, 3, 14, 15, 4, 5, 16, 17, 6, 7, 18, 19, 20, 8, 9, 21, 10]
fine_names = {0: "layer top", 1: "x ready", 11: "K/V requested", 12: "norm done", 13: "wqkv dots done", 2: "qkv published", 3: "q/k/v ready",
              14: "wave partial done", 15: "wave 0: attn output / partial published", 4: "P2 top (attn combined+published, wo|w1 requested)", 5: "attn vector ready", 16: "wo dots done",
              17: "h published", 6: "P3 top (w3 requested)", 7: "h ready", 18: "norm done", 19: "w1|w3 dots done", 20: "act published",
              8: "P4 top (w2 requested)", 9: "act ready", 21: "w2 dots done", 10: "x published"}
for blk in range(2):
    d = np.diff(st[blk, :, :11], axis=1)[2:-2]  # skip the first/last layers
    per = d.mean(axis=0)
    tot = (st[blk, 1:, 0] - st[blk, :-1, 0])[2:-2].mean()
    print(f"workgroup {'first' if blk == 0 else 'last'}: {tot:.0f} ticks = {tot * TICK_US:.1f} us per layer; per segment (ticks): " +
          ", ".join(f"{n} {v:.0f}" for n, v in zip(names, per)))
    if st[blk, 2, 12] > 0 and blk == 0:  # fine stamps present (v2 kernel)
        seq = st[blk][:, fine_order][2:-2]
        dd = np.diff(seq, axis=1).mean(axis=0)
        nxt = (st[blk, 3:-1, 0] - st[blk, 2:-2, 10]).mean()
        print("  fine (us since the previous stamp): " + "; ".join(f"{fine_names[k]} +{v * TICK_US:.2f}" for k, v in zip(fine_order[1:], dd)) +
              f"; next layer top (next wqkv requested) +{nxt * TICK_US:.2f}")
for blk in range(2):
    # slot 22: attempts of each hand-off poll that came back incomplete (wave 0), 12 bits per hop
    tw = st[blk, 2:-2, 22].astype(np.uint64)
    hops = ["x", "q/k/v", "attention vector", "h", "activation"]
    tr = [((tw >> np.uint64(12 * k)) & np.uint64(4095)).astype(np.float64) for k in range(5)]
    print(f"  workgroup {'first' if blk == 0 else 'last'}, incomplete poll attempts per hand-off (mean over layers; 0 = the first look saw everything): " +
          ", ".join(f"{h} {t.mean():.2f} (max {int(t.max())})" for h, t in zip(hops, tr)))
print("(s_memtime ticks)")
