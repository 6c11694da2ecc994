#!/bin/bash
# persistent-kernel iteration: bit-identity tests, phase stamps, step time vs the per-layer launches
set -u
mkdir -p gpurun_out/mega
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/mega/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/mega/pytest.log
tail -3 gpurun_out/mega/pytest.log
timeout 300 python tools/mega_stamps.py 1024 > gpurun_out/mega/stamps.log 2>&1; tail -4 gpurun_out/mega/stamps.log
timeout 600 python tools/sweep.py --reps 2 --configs '[{"mega":0},{"mega":1}]' > gpurun_out/mega/sweep.log 2>&1; grep ms gpurun_out/mega/sweep.log
