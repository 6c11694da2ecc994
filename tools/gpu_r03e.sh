#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/attn_ab.py 2>&1 < /dev/null | grep -v amdgpu.ids | tee $OUT/attn_ab.txt
for v in mfma valu; do
  if [ $v = valu ]; then export MV_ATTN_VALU=1; else unset MV_ATTN_VALU; fi
  echo "== engines on the $v attention" | tee -a $OUT/sampling.txt
  timeout 300 python -m pytest tests/test_stage2_gpu.py -q -s -k "sampling or logits" 2>&1 < /dev/null | grep -E "identical|passed|failed|max\|d" | cut -c1-200 | tee -a $OUT/sampling.txt
done
unset MV_ATTN_VALU
