#!/bin/bash
# rocprofv3 passes for profiles/: kernel stats of one bench step, then PMC passes (own runs) on the kernel probe.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
find /tmp/prof_stats -name '*stats*.csv' -exec cp {} $OUT/ \;
for ctr in ${PMC_CTRS:-}; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$ctr -o probe -- python $GRAFT_REPO_ROOT/tools/probe_kernels.py ${PROBE_ARGS:---iters 24} > $OUT/probe_$ctr.log 2>&1
  f=$(find /tmp/prof_$ctr -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/summarise_pmc.py "$f" $ctr > $OUT/pmc_$ctr.txt 2>&1
done
ls -la $OUT
