#!/usr/bin/env python3
"""Achieved GB/s of the vocoder's HBM-bound kernels, from the rocprofv3 --kernel-trace --stats summary of one default bench step
(north_star: "coalesced HBM reads shown in rocprof as achieved GB/s against the 8 TB/s roofline"):

    python tools/vocoder_hbm.py profiles/r04_final/bench_kernel_stats.csv > profiles/vocoder_hbm_kernels.json

Algorithmic bytes per launch at the bench's size (12.8 s = 307 200 samples, 4 diffusion bands of the recalled basic.yaml UNet: channels
48 / 192 / 768 / 3072 at 76 800 / 19 200 / 4 800 / 1 200 rows -- every level holds the same 3 686 400 elements per band):
  norm_act_kernel<GM>  reads the fp32 activation, writes the next convolution's bf16 hi + lo operands (or one fp32 tensor): 8 B per element
                       (all instances together; <4> also leaves the GroupNorm records of its output)
  gn_partial_kernel    reads the fp32 activation once: 4 B per element            (round 6: no launch left at this shape)
  add_kernel           reads two fp32 tensors, writes one: 12 B per element       (round 6: none left either)
  conv_in_kernel / conv_in_rows_kernel<GM>   the 48-channel end of a band: 4 B per element written (+ the 1-channel signal read)
  convtr_out_kernel / convtr_out_rows_kernel  4 B per element read (+ the 1-channel signal written); the rows kernel also applies the decoder's
                       last GroupNorm + activation, which used to be a norm_act launch of its own
bench.py attaches the result to `vocoder_roofline.hbm_kernels` (source file named there)."""
import csv, json, sys

ELEMS = 4 * 3686400            # 4 bands
BYTES = {"norm_act_kernel": 8 * ELEMS, "gn_partial_kernel": 4 * ELEMS, "add_kernel": 12 * ELEMS,
         "conv_in_kernel": 4 * ELEMS + 4 * 4 * 307200, "convtr_out_kernel": 4 * ELEMS + 4 * 4 * 307200,
         "conv_in_rows_kernel": 4 * ELEMS + 4 * 4 * 307200, "convtr_out_rows_kernel": 4 * ELEMS + 4 * 4 * 307200}
PEAK = 8000.0
acc = {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        for k in BYTES:
            if k + "(" in r["Name"] or k + "<" in r["Name"]:   # template instances of one kernel are added up
                calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
                if k == "add_kernel" and calls < 40:   # the 1-channel adds of the band sum: not this shape
                    continue
                c0, t0 = acc.get(k, (0, 0.0))
                acc[k] = (c0 + calls, t0 + tot)
rows = {}
for k, (calls, tot) in acc.items():
    rows[k] = {"calls": calls, "avg_us": round(tot / calls / 1e3, 2), "bytes_per_launch": BYTES[k],
               "achieved_GBs": round(BYTES[k] / (tot / calls), 1), "frac_of_8TBs": round(BYTES[k] / (tot / calls) / PEAK, 3),
               "total_ms": round(tot / 1e6, 2)}
if "gn_partial_kernel" in rows:
    rows["gn_partial_kernel"]["note"] = ("since round 4 the launches in front of a decoder level also carry the skip addition (3 x the bytes of a plain "
                                         "statistics pass): priced at 4 B per element the figure is a lower bound")
print(json.dumps({"source": sys.argv[1], "peak_GBs": PEAK, "kernels": rows}, indent=1))
