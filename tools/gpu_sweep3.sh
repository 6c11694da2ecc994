#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 600 python tools/probe_kernels.py --iters 240 --pos 1024 --kinds 0,1,0x101,2,3,4,5 > gpurun_out/probe.log 2>&1
grep '^kind' gpurun_out/probe.log
timeout 900 python tools/sweep.py --configs "$(cat tools/sweep3.json)" > gpurun_out/sweep.log 2>&1
grep ms gpurun_out/sweep.log
bash tools/gpu_profile.sh r01c > /dev/null 2>&1
head -8 gpurun_out/prof_r01c/bench_kernel_stats.csv | cut -c1-150
