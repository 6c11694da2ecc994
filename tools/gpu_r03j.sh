#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
export FOLD_STEPS=512 FOLD_SKIP_LAUNCHES=1
for v in product kvnorm kvnormq product kvnorm; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/variants.txt
  timeout 200 python tools/fold_sweep.py 2>&1 < /dev/null | grep "us/step" | tee -a $OUT/variants.txt
done
export MV_HIP_LIB=$LIB/libmv_kvnorm.so
NU_CASES="bf16:4,int8:1" timeout 300 python tools/nu_sweep.py 2>&1 < /dev/null | grep "mega_fold=0" | tee $OUT/nu_kvnorm.txt
timeout 100 python tools/mega_stamps.py 1024 2>&1 < /dev/null | grep -E "us/step|fine" | cut -c1-1600 | tee $OUT/stamps_kvnorm.txt
unset MV_HIP_LIB
timeout 100 python tools/mega_stamps.py 1024 2>&1 < /dev/null | grep -E "us/step|fine" | cut -c1-1600 | tee $OUT/stamps_product.txt
