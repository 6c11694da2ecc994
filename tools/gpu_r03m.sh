#!/bin/bash
# (the variant libraries: git apply tools/experiments/t256_schedule_variants.patch; tools/build_variant.sh deep1 "-DT256_DEEP=1" mbd; ... deep2, deep1prio, prio1 "-DT256_SETPRIO=1", prio2, stag "-DT256_STAGGER=1")
# t256 schedule variants on ONE box: s_setprio forms, staggered piece issue, two-stages-ahead with counted vmcnt (T256_DEEP).
# Correctness first (the GEMM parity tests under each library), then the stage time.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
for v in product deep1 deep2 deep1prio prio1 prio2 stag product; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/t256_variants.txt
  timeout 300 python -m pytest tests/test_mbd_gpu.py -m gpu -x -q -k "256x256" 2>&1 < /dev/null | tail -2 | tee -a $OUT/t256_variants.txt
  MBD_PRECISIONS=3 MBD_REPS=4 timeout 300 python tools/mbd_time.py 2>&1 < /dev/null | grep tokens_to_wav | tail -3 | tee -a $OUT/t256_variants.txt
done
unset MV_HIP_LIB
