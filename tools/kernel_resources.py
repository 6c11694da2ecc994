#!/usr/bin/env python3
"""VGPRs / scratch / spills of every kernel of one source, from hipcc's -Rpass-analysis=kernel-resource-usage remarks:
    python tools/kernel_resources.py decode_mega ["-DMEGA_EW2_CHUNKS=7"]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flags = sys.argv[2].split() if len(sys.argv) > 2 else []
csrc = os.path.join(ROOT, "metavoice-src_amd", "csrc")
p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    *flags, "-c", src + ".hip", "-o", "/tmp/kernel_resources.o", "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, capture_output=True, text=True)
cur, vals = None, {}
for line in p.stderr.split("\n"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur, vals = m.group(1), {}
    for k, label in (("VGPRs", "vgpr"), (r"ScratchSize \[bytes/lane\]", "scratch"), ("VGPRs Spill", "spill"), ("SGPRs Spill", "sgpr_spill"), ("Occupancy \[waves/SIMD\]", "occ")):
        m = re.search(r"remark:\s+" + k + r": (\d+)", line)
        if m:
            vals[label] = int(m.group(1))
    if "LDS Size" in line and cur:
        tmpl = re.search(r"kernelILi(\d+)ELi(\d+)ELb(\d)", cur)
        name = re.sub(r"^_ZN?\d*(_GLOBAL__N_1)?\d*", "", cur)[:40] + (f" <{tmpl.group(1)}, {tmpl.group(2)}, {tmpl.group(3)}>" if tmpl else "")
        print(f"{name:60s} {vals}")
        cur = None
if p.returncode:
    print(p.stderr[-2000:])
