#!/usr/bin/env python3
"""One bench-shaped utterance (prompt 128, 1920 new tokens) on the persistent kernel that draws its own tokens (4 launches of up to
512 steps): prints the algorithmic bytes of those launches.  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` to get the HBM
traffic of the same launches (tools/gpu_profile_r02b.sh -> profiles/roofline_traffic.json)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
mv = importlib.import_module("metavoice-src_amd")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
T, N_NEW, MEGA_STEPS = 128, 1920, 512
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))
e = mv.Stage1Engine(args, max_utterances=1)
e.load_state_dict(sd); del sd
e.finalize()
e.begin([mv.synthetic.prompt_tokens(T, seed=0)], mv.synthetic.speaker_embedding(0), mv.SamplingParams(top_p=0.95, guidance_scale=3.0, seed=1),
        max_new_tokens=N_NEW, end_of_audio_token=9999, prefill_mode=1)
steps = e.run(poll_every=1 << 20)
assert e.decode_path() == (2, 0), e.decode_path()
launches = -(-steps // MEGA_STEPS)
total = steps * bench.weight_bytes(args, None) + sum(bench.kv_bytes(args, q) for q in range(T, T + steps))
print(json.dumps({"steps": steps, "launches": launches, "algorithmic_bytes_total": total, "algorithmic_bytes_per_launch": total / launches,
                  "us_per_step": e.last_run_ms / steps * 1e3, "contexts": f"{T + 1}..{T + steps}"}))
