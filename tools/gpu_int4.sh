#!/bin/bash
set -u
mkdir -p gpurun_out/mega
timeout 600 python -m pytest tests/test_int4_gpu.py -x -q -k "engine or persistent" > gpurun_out/mega/pytest_i4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/mega/pytest_i4.log
tail -4 gpurun_out/mega/pytest_i4.log
timeout 600 python tools/sweep.py --quant int4 --reps 2 --configs '[{"mega":0},{"mega":1}]' > gpurun_out/mega/sweep_i4.log 2>&1; grep ms gpurun_out/mega/sweep_i4.log; tail -3 gpurun_out/mega/sweep_i4.log | grep -i "error\|Trace" 
