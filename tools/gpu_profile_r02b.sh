#!/bin/bash
# rocprofv3 passes for profiles/r02b_fold: (A) kernel stats of one default bench step (decode_mega_kernel = 4 launches of up to 512
# decode steps each), (B) FETCH_SIZE PMC pass on the same launch shape (own run: PMC never shares a run with trace domains other
# than --kernel-trace).  Every command is bounded by `timeout` and reads nothing from stdin.
set -u
TAG=${1:-r02b_fold}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 < /dev/null
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -o fold -- python $GRAFT_REPO_ROOT/tools/fold_traffic.py > $OUT/fold_traffic.log 2>&1 < /dev/null
f=$(find /tmp/prof_fetch -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && timeout 60 python $GRAFT_REPO_ROOT/tools/summarise_pmc.py "$f" FETCH_SIZE > $OUT/pmc_FETCH_SIZE.txt 2>&1 < /dev/null
ls -la $OUT
