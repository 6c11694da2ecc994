#!/bin/bash
# attn_publish_tagged without the scratch round trip (the split-partial publish of every layer above 256 keys): A/B on ONE box
# against the previous library (libmv_base.so), one utterance on both persistent paths, then 2 / 4 utterances and int8 / int4; token
# streams must be identical.  Then the persistent-kernel tests.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
export FOLD_STEPS=512 FOLD_SKIP_LAUNCHES=1
for v in base product base product; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/publish_ab.txt
  timeout 200 python tools/fold_sweep.py 2>&1 < /dev/null | grep "us/step" | tee -a $OUT/publish_ab.txt
done
for v in base product; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/publish_ab.txt
  timeout 400 python tools/nu_sweep.py 2>&1 < /dev/null | grep "us/step" | tee -a $OUT/publish_ab.txt
done
unset MV_HIP_LIB FOLD_SKIP_LAUNCHES
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_fullsize_golden_gpu.py -m gpu -x -q 2>&1 < /dev/null | tail -3 | tee -a $OUT/publish_ab.txt
