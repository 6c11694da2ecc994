#!/bin/bash
# rocprofv3 passes for profiles/r02*: (A) kernel stats of one default bench step (full pipeline incl. multi-band diffusion),
# (B) FETCH_SIZE PMC pass on the persistent decode kernel (own run: PMC never shares a run with trace domains other than
# --kernel-trace), (C) SQ PMC pass (wave cycles: parked / issue-stalled / active, VALU activity) on the same probe.
# Every command is bounded by `timeout` and reads nothing from stdin.
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 < /dev/null
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -o probe -- python $GRAFT_REPO_ROOT/tools/probe_kernels.py --kinds 6 --pos 2040 --iters 24 > $OUT/probe_FETCH_SIZE.log 2>&1 < /dev/null
f=$(find /tmp/prof_fetch -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && timeout 60 python $GRAFT_REPO_ROOT/tools/summarise_pmc.py "$f" FETCH_SIZE > $OUT/pmc_FETCH_SIZE.txt 2>&1 < /dev/null
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"
timeout 400 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d /tmp/prof_sq -o probe -- python $GRAFT_REPO_ROOT/tools/probe_kernels.py --kinds 6 --pos 1087 --iters 24 > $OUT/probe_SQ.log 2>&1 < /dev/null
f=$(find /tmp/prof_sq -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then for c in $SQ; do timeout 60 python $GRAFT_REPO_ROOT/tools/summarise_pmc.py "$f" $c 2>&1 < /dev/null | grep -E "^#|decode_mega" ; done > $OUT/pmc_SQ.txt; fi
ls -la $OUT
