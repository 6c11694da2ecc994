#!/bin/bash
# K/V slices as buffer loads (out-of-range offsets return zeros: no fill, no branches, no copy behind a load) vs the previous library, one box.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03t
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
export FOLD_STEPS=512 FOLD_SKIP_LAUNCHES=1
for v in base product base product; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/kv_buffer_ab.txt
  timeout 200 python tools/fold_sweep.py 2>&1 < /dev/null | grep "us/step" | tee -a $OUT/kv_buffer_ab.txt
done
for v in base product; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/kv_buffer_ab.txt
  timeout 400 python tools/nu_sweep.py 2>&1 < /dev/null | grep "us/step" | grep "mega_fold=1" | tee -a $OUT/kv_buffer_ab.txt
done
unset MV_HIP_LIB FOLD_SKIP_LAUNCHES
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 < /dev/null | tail -2 | tee -a $OUT/kv_buffer_ab.txt
