#!/usr/bin/env python3
"""Per-kernel mean of one PMC counter from a rocprofv3 counter_collection.csv."""
import csv, sys, collections
path, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") == ctr:
            acc[row["Kernel_Name"][:110]].append(float(row["Counter_Value"]))
print(f"# {ctr}: per-dispatch mean over the run (raw counter units; FETCH_SIZE/WRITE_SIZE are KiB-like units of 1024 B? see DESIGN.md)")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/len(v):14.1f}  n={len(v):5d}  {k}")
