#!/bin/bash
# Round 6: decode-step time of one utterance (the headline kernel) under a list of libraries built by tools/build_variant.sh with different
# compiler flags on decode_mega.hip; token hashes must agree.   tools/flag_sweep.sh <tag> <rounds> <lib> [<lib> ...]   (`product` = the product library)
set -u
TAG=$1; ROUNDS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
for r in $(seq 1 $ROUNDS); do for v in "$@"; do
  if [ "$v" = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$ROOT/metavoice-src_amd/lib/libmv_$v.so; fi
  echo "== $v (round $r)" | tee -a $OUT/flag_ab.txt
  NU_CASES="${NU_CASES:-bf16:1}" timeout 200 python tools/nu_sweep.py 1920 2>&1 < /dev/null | grep -E "us/step|rror" | tee -a $OUT/flag_ab.txt
done; done
