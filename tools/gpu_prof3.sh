#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/prof3
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -o sw -- python $GRAFT_REPO_ROOT/tools/sweep.py --quant int4 --reps 1 --configs '[{}]' > $GRAFT_REPO_ROOT/gpurun_out/prof3/log.txt 2>&1
find /tmp/p3 -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof3/ \;
head -9 $GRAFT_REPO_ROOT/gpurun_out/prof3/sw_kernel_stats.csv | cut -c1-140
