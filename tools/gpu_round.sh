#!/bin/bash
# End-of-milestone GPU session: parity suite, bench line, rocprofv3 kernel stats (+ optional PMC passes) into gpurun_out/.
set -u
TAG=${1:-r01}
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -x -q tests > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -1 gpurun_out/bench.log | cut -c1-1500
PMC_CTRS="${PMC_CTRS:-}" bash tools/gpu_profile.sh $TAG > /dev/null 2>&1
head -8 gpurun_out/prof_$TAG/bench_kernel_stats.csv | cut -c1-130
