#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest -m gpu -x -q ${1:-tests} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -${2:-40} gpurun_out/pytest_gpu.log
