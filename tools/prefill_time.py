#!/usr/bin/env python3
"""mv_begin (state upload + MFMA prefill + first token) of the full-size engine at several prompt lengths, with the 64-column x all-rows
GEMM (`gemm_skinny_kernel`, round 4) and with the 64 x 64 tiles (the default: MV_GEMM_SKINNY unset); the first sampled tokens must agree.
usage: prefill_time.py [reps]   env PREFILL_T="50,128,256,512,1100,2016" """
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
e = mv.Stage1Engine(args, max_utterances=1)
e.load_state_dict(sd); del sd
e.finalize()
spk, sp = mv.synthetic.speaker_embedding(0), mv.SamplingParams(top_k=1, guidance_scale=3.0, seed=1)
for T in [int(x) for x in os.environ.get("PREFILL_T", "50,128,256,512,1100,2016").split(",")]:
    prompt = mv.synthetic.prompt_tokens(T, seed=0)
    row = []
    for tiles in os.environ.get("PREFILL_MODES", "skinny,skinny64,64x64").split(","):
        os.environ.pop("MV_GEMM_SKINNY", None); os.environ.pop("MV_GEMM_SKINNY_COLS", None)
        if tiles != "64x64": os.environ["MV_GEMM_SKINNY"] = "1"
        if tiles == "skinny64": os.environ["MV_GEMM_SKINNY_COLS"] = "64"
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.begin([prompt], spk, sp, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=1)   # returns after a stream sync
            best = min(best, (time.perf_counter() - t0) * 1e3)
        e.run(max_steps=1)
        row.append((tiles, best, e.tokens(0)[T]))
    os.environ.pop("MV_GEMM_SKINNY", None); os.environ.pop("MV_GEMM_SKINNY_COLS", None)
    print(f"T = {T:5d} ({2 * T} rows): " + "; ".join(f"{n} {ms:6.2f} ms (first token {tok})" for n, ms, tok in row), flush=True)
