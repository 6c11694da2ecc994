#!/bin/bash
# GroupNorm statistics from the GEMM epilogues: the diffusion tests, then the stage time with and without (MV_MBD_NO_GN_FUSE=1).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mbd_gpu.py -m gpu -x -q -s 2>&1 < /dev/null | grep -E "gn fuse|passed|failed|Error|error" | tee $OUT/gn_fuse.txt
for v in two_pass fused two_pass fused; do
  unset MV_MBD_NO_GN_FUSE
  if [ $v = two_pass ]; then export MV_MBD_NO_GN_FUSE=1; fi
  echo "== $v" | tee -a $OUT/gn_fuse.txt
  MBD_PRECISIONS=3 MBD_REPS=4 timeout 300 python tools/mbd_time.py 2>&1 < /dev/null | grep tokens_to_wav | tail -3 | tee -a $OUT/gn_fuse.txt
done
