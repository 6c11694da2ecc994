#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
for v in product nukv product; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/nu_variants.txt
  NU_CASES="bf16:2,bf16:4" timeout 300 python tools/nu_sweep.py 2>&1 < /dev/null | grep "mega_fold=0" | tee -a $OUT/nu_variants.txt
done
for v in product abl1 abl2; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/t256_ablation.txt
  MBD_PRECISIONS=3 MBD_REPS=3 timeout 300 python tools/mbd_time.py 2>&1 < /dev/null | grep tokens_to_wav | tee -a $OUT/t256_ablation.txt
done
unset MV_HIP_LIB
