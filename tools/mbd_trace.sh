# per-launch durations of the vocoder's helper kernels by grid shape: rocprofv3 --kernel-trace over tools/mbd_time.py (one precision-3 run) -> gpurun_out/$1/mbd_trace.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-mbdtrace}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
MBD_PRECISIONS=3 MBD_REPS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/mbdtrace -o t -- python $ROOT/tools/mbd_time.py > $OUT/mbd_time.log 2>&1 < /dev/null
f=$(find /tmp/mbdtrace -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' | tee $OUT/mbd_trace.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for k in ("gn_partial_kernel", "gn_add_rows_kernel", "gn_final_kernel", "norm_act_kernel", "conv_in_kernel", "conv_in_rows_kernel", "convtr_out_kernel", "convtr_out_rows_kernel", "add_kernel", "mbd_gemm_thin", "mbd_gemm_glds3", "mbd_gemm_t256"):
        if k in n:
            acc[(k, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = collections.defaultdict(float)
for (k, gx, gy, gz), v in sorted(acc.items()):
    print(f"{k:22s} grid {gx:>8s} x {gy:>4s} x {gz:>2s}: {len(v):4d} launches, mean {sum(v) / len(v):8.1f} us, total {sum(v) / 1e3:7.2f} ms")
    tot[k] += sum(v) / 1e3
print({k: round(v, 2) for k, v in tot.items()})
PY
tail -3 $OUT/mbd_time.log
