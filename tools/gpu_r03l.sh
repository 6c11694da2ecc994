#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
for v in product sp2 tord sp2tord sp0 product; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/t256_variants.txt
  MBD_PRECISIONS=3 MBD_REPS=4 timeout 300 python tools/mbd_time.py 2>&1 < /dev/null | grep tokens_to_wav | tail -3 | tee -a $OUT/t256_variants.txt
done
unset MV_HIP_LIB
