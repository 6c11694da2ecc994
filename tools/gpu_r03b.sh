#!/bin/bash
# Round 3, second GPU call: (1) decode paths after the token draw went back inline, (2) the 256 x 256 MBD GEMM: parity tests, stage
# time with / without it on ONE box, rocprof kernel stats of the stage.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
FOLD_STEPS=512 FOLD_SKIP_LAUNCHES=1 timeout 200 python tools/fold_sweep.py 2>&1 < /dev/null | grep "us/step" | tee $OUT/fold.txt
timeout 600 python -m pytest tests/test_mbd_gpu.py -x -q -s > $OUT/pytest_mbd.log 2>&1 < /dev/null; echo "pytest mbd rc=$?"
grep -E "^\.?\[mbd" $OUT/pytest_mbd.log | cut -c1-300; tail -3 $OUT/pytest_mbd.log
echo "== 256x256 tiles" | tee $OUT/mbd_time.txt
MBD_PRECISIONS=3 MBD_REPS=4 timeout 300 python tools/mbd_time.py 2>&1 < /dev/null | grep tokens_to_wav | tee -a $OUT/mbd_time.txt
echo "== 128x128 tiles (MV_MBD_NO_T256)" | tee -a $OUT/mbd_time.txt
MV_MBD_NO_T256=1 MBD_PRECISIONS=3 MBD_REPS=4 timeout 300 python tools/mbd_time.py 2>&1 < /dev/null | grep tokens_to_wav | tee -a $OUT/mbd_time.txt
export TMPDIR=/tmp
cd /tmp
MBD_PRECISIONS=3 MBD_REPS=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mbd -o mbd -- python $GRAFT_REPO_ROOT/tools/mbd_time.py > $OUT/mbd_rocprof.log 2>&1 < /dev/null
find /tmp/prof_mbd -name '*kernel_stats.csv' -exec cp {} $OUT/mbd_kernel_stats.csv \;
head -14 $OUT/mbd_kernel_stats.csv | cut -c1-200
