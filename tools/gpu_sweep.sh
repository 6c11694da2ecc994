#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 900 python tools/sweep.py --configs "$(cat ${1:-tools/sweep1.json})" > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log | tail -40
