#!/usr/bin/env python3
"""The reference's own CPU path and its port (oracle/stage1.py) timed back to back on ONE box -> profiles/cpu_baseline_pair.json.

    python tools/cpu_baseline_pair.py        (build container: needs /root/reference or $MV_REFERENCE; no GPU needed)

`bench.py`'s `cpu_baseline` times the reference itself where its checkout exists (kind "reference") and the port elsewhere (the GPU box
has no reference checkout); this pair makes port / reference a known ratio.  Both legs: full-size synthetic
MetaVoice-1B, fp32 eager torch-CPU, prefill T = 128 + 24 greedy decode steps, the same prompt and speaker embedding."""
import importlib
import json
import os
import platform
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

mv = importlib.import_module("metavoice-src_amd")
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
ref = bench.reference_dir()
assert ref is not None, "no reference checkout here"
legs, seen = {}, {"reference": [], "port": []}
for rnd in range(3):   # A B A B A B on one box; the FASTEST run of each leg is reported (a shared host: slow runs are interference), all are listed
    for name, fn in (("reference", lambda: bench.cpu_baseline_reference(mv, cfgk, ref)), ("port", lambda: bench.cpu_baseline_port(mv, cfgk))):
        r = fn()
        seen[name].append(r["value"])
        if name not in legs or r["value"] > legs[name]["value"]:
            legs[name] = r
    print(rnd, seen, flush=True)
out = {"host": f"{platform.processor() or platform.machine()}, {os.cpu_count()} logical cores, {torch.get_num_threads()} torch threads",
       "torch": torch.__version__, "reference": legs["reference"], "port": legs["port"], "all_runs_audio_s_per_s": seen,
       "port_over_reference": round(legs["port"]["value"] / legs["reference"]["value"], 3)}
with open(os.path.join(ROOT, "profiles", "cpu_baseline_pair.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
