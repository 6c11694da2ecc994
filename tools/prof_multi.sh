# rocprofv3 --kernel-trace --stats of the 4- and 8-utterance bench legs (one step each) -> gpurun_out/<tag (default r06w)>/kernel_stats_u{4,8}.csv; run through gpurun
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-r06w}; mkdir -p $OUT; export TMPDIR=/tmp
for U in 4 8; do
  ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_u$U -o bench -- python $ROOT/bench.py --utterances-per-gpu $U --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $OUT/bench_u$U.log 2>&1 < /dev/null )
  find /tmp/prof_u$U -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_u$U.csv \;
  head -6 $OUT/kernel_stats_u$U.csv | cut -c1-150
  tail -1 $OUT/bench_u$U.log | cut -c1-300
done
