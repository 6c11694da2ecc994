#!/bin/bash
# Multi-utterance RMSNorm: statistics spread over the 8 waves, totals / reads / applies of all utterances overlapped -- vs the previous
# library, one box; token streams must stay identical; then the multi-utterance bit-identity tests.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03v
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
for v in base product base product; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/nu_norm_ab.txt
  NU_CASES="bf16:2,bf16:4" timeout 300 python tools/nu_sweep.py 2>&1 < /dev/null | grep "us/step" | grep "mega_fold=1" | tee -a $OUT/nu_norm_ab.txt
done
unset MV_HIP_LIB
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "several or utterances or batch" 2>&1 < /dev/null | tail -2 | tee -a $OUT/nu_norm_ab.txt
