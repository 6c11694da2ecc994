#!/usr/bin/env python3
"""One-screen summary of a bench.py JSON line: python tools/bench_summary.py gpurun_out/<tag>/bench_line.json"""
import json, sys
o = json.loads(open(sys.argv[1]).read())
r = o.get("roofline", {})
print("value", o.get("value"), o.get("unit"), "| ms/step", o.get("ms_per_step"), "| roofline frac", r.get("frac"), "avg_launch_us", r.get("avg_launch_us"),
      "traffic", r.get("traffic"))
print("stages", o.get("stage_ms_per_step"))
v = o.get("vocoder_roofline", {})
print("vocoder", {k: v.get(k) for k in ("frac", "useful_frac_of_bf16_peak")}, "| prefill", o.get("prefill_roofline"), "| cpu", o.get("cpu_baseline", {}).get("value"))
for k, e in o.get("extra_configs", {}).items():
    print(" ", k, "| value", e.get("value"), "ms/step", e.get("ms_per_step"), "frac", e.get("roofline", {}).get("frac"), "step frac", e.get("step_roofline_frac"),
          "fallbacks", e.get("decode_fallbacks"), "wall s", e.get("leg_wall_s"), {x: e[x] for x in e if x.startswith("host_") or x.startswith("stage_")})
