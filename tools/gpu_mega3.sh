#!/bin/bash
set -u
mkdir -p gpurun_out/mega
timeout 300 python tools/check_mega3.py > gpurun_out/mega/check3.log 2>&1; echo "check rc=$?"
tail -25 gpurun_out/mega/check3.log
MV_MEGA=2 timeout 200 python tools/mega_stamps.py 1024 > gpurun_out/mega/stamps3.log 2>&1; tail -4 gpurun_out/mega/stamps3.log
timeout 600 python tools/sweep.py --reps 2 --configs '[{"mega":1},{"mega":2}]' > gpurun_out/mega/sweep3.log 2>&1; grep ms gpurun_out/mega/sweep3.log
