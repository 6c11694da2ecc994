#!/bin/bash
set -u
mkdir -p gpurun_out/mega
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/mega/pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/mega/pytest_all.log
tail -3 gpurun_out/mega/pytest_all.log
timeout 300 python tools/mega_stamps.py 1024 > gpurun_out/mega/stamps.log 2>&1; tail -4 gpurun_out/mega/stamps.log | head -2
timeout 600 python tools/sweep.py --reps 2 --configs '[{"mega":0},{"mega":1}]' > gpurun_out/mega/sweep.log 2>&1; grep ms gpurun_out/mega/sweep.log
