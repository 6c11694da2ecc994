#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 600 python tools/probe_kernels.py --iters 240 --pos 1024 --kinds 0,0x100,0x200,1,0x101,2,0x102,0x202,3,0x103,0x203,4,0x104,0x204,5,0x105 > gpurun_out/probe.log 2>&1
grep '^kind' gpurun_out/probe.log
timeout 900 python tools/sweep.py --configs "$(cat tools/sweep2.json)" > gpurun_out/sweep.log 2>&1
grep ms gpurun_out/sweep.log
bash tools/gpu_profile.sh r01b > /dev/null 2>&1
head -12 gpurun_out/prof_r01b/bench_kernel_stats.csv
