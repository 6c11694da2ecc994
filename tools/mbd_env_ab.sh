# vocoder stage time under a list of environment settings on ONE box, two rounds: tools/mbd_env_ab.sh <out> "A=1" "A=2 B=3" ...  -> gpurun_out/<out>/env_ab.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; shift
cd $ROOT
for round in 1 2; do
  for setting in "$@"; do
    echo "== [$setting] (round $round)" >> $OUT/env_ab.txt
    env $setting MBD_PRECISIONS=3 MBD_REPS=4 timeout 300 python tools/mbd_time.py 2>&1 | grep tokens_to_wav | tail -3 >> $OUT/env_ab.txt
  done
done
cat $OUT/env_ab.txt
