#!/bin/bash
# Round 6: rocprofv3 --kernel-trace --stats over tools/prefill_time.py (mv_begin alone: upload + MFMA prefill + first token) -> gpurun_out/<tag>/prefill_kernel_stats.csv
#   tools/prefill_trace.sh <tag> [T list, default 128] [modes, default 64x64]
TAG=$1; ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PREFILL_T=${2:-128} PREFILL_MODES=${3:-64x64}
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_prefill -o pf -- python $ROOT/tools/prefill_time.py 10 > $OUT/prefill_time.txt 2>&1 < /dev/null
find /tmp/prof_prefill -name '*kernel_stats.csv' -exec cp {} $OUT/prefill_kernel_stats.csv \;
cat $OUT/prefill_time.txt | grep "T =" ; head -14 $OUT/prefill_kernel_stats.csv | cut -c1-150
