#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 600 python tools/sweep.py --configs "$(cat ${1:-tools/sweep4.json})" > gpurun_out/sweep.log 2>&1
grep " ms " gpurun_out/sweep.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -1 gpurun_out/bench.log
tail -3 gpurun_out/bench.err
