#!/bin/bash
# Compiler-visible vmcnt(0) behind every poll (MEGA_POLL_SCOREBOARD) x abort word read through a flat or an LDS pointer (MEGA_ABORT_LDS):
# product = scoreboard 1 + flat; sb0 = scoreboard 0 + flat (the r03q product); sb1lds; sb0lds.  One utterance bf16 (both persistent
# paths), then 2 / 4 utterances and int8 / int4 for product and sb0.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
export FOLD_STEPS=512 FOLD_SKIP_LAUNCHES=1
for v in sb0 product sb1lds sb0lds sb0 product sb1lds; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/scoreboard_ab.txt
  timeout 200 python tools/fold_sweep.py 2>&1 < /dev/null | grep "us/step" | tee -a $OUT/scoreboard_ab.txt
done
for v in sb0 product sb1lds; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/scoreboard_ab.txt
  NU_CASES="bf16:2,bf16:4,int8:1,int4:1" timeout 400 python tools/nu_sweep.py 2>&1 < /dev/null | grep "us/step" | tee -a $OUT/scoreboard_ab.txt
done
