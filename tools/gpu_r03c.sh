#!/bin/bash
# Round 3, third GPU call: MBD stage -- L2-prefetch variants of the 256 x 256 GEMM on ONE box, device hand-off from the codec,
# persistent workspaces; parity tests of the MBD + codec + TTS paths.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
timeout 900 python -m pytest tests/test_mbd_gpu.py tests/test_codec_gpu.py tests/test_tts_gpu.py -x -q -s > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"
grep -E "^\.?\[mbd" $OUT/pytest.log | cut -c1-260; tail -3 $OUT/pytest.log
for v in product nopf pf2 pf5 spread product; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/mbd_variants.txt
  MBD_PRECISIONS=3 MBD_REPS=4 timeout 300 python tools/mbd_time.py 2>&1 < /dev/null | grep tokens_to_wav | tee -a $OUT/mbd_variants.txt
done
unset MV_HIP_LIB
export TMPDIR=/tmp
cd /tmp
MBD_PRECISIONS=3 MBD_REPS=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mbd -o mbd -- python $GRAFT_REPO_ROOT/tools/mbd_time.py > $OUT/mbd_rocprof.log 2>&1 < /dev/null
find /tmp/prof_mbd -name '*kernel_stats.csv' -exec cp {} $OUT/mbd_kernel_stats.csv \;
head -8 $OUT/mbd_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/bench_stderr.txt < /dev/null; echo "bench rc=$?"
python - <<'PY'
import json,os
o=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r03c/bench_line.json")).read().strip().splitlines()[-1])
print("bench value", o["value"], "ms/step", o["ms_per_step"], "roofline", o["roofline"]["frac"], "avg_step_us", o["roofline"].get("avg_step_us"), "stages", o["stage_ms_per_step"])
PY
