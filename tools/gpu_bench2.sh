#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -x -q tests > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 600 python tools/sweep.py --reps 2 --configs "$(cat ${1:-tools/sweep4.json})" > gpurun_out/sweep.log 2>&1
grep " ms " gpurun_out/sweep.log
timeout 300 python tools/mega_stamps.py 1024 2>&1 | tail -4
