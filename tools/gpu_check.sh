#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel stats.  Run through gpurun from the repo root.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -c "import torch; print('devices', torch.cuda.device_count(), torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
nproc >> gpurun_out/env.log; grep -m1 'model name' /proc/cpuinfo >> gpurun_out/env.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -2 gpurun_out/bench.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_under_rocprof.log 2>&1 )
find /tmp/prof -name '*stats*' -exec cp {} gpurun_out/prof/ \;
ls -la gpurun_out/prof
