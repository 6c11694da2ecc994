#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/prof2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o sw -- python $GRAFT_REPO_ROOT/tools/sweep.py --reps 1 --configs '[{"mega":0}]' > $GRAFT_REPO_ROOT/gpurun_out/prof2/log.txt 2>&1
find /tmp/p2 -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof2/ \;
head -9 $GRAFT_REPO_ROOT/gpurun_out/prof2/sw_kernel_stats.csv | cut -c1-140
grep " ms " $GRAFT_REPO_ROOT/gpurun_out/prof2/log.txt
