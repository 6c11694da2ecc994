#!/bin/bash
# Round 3: consolidated state -- the whole GPU suite, the default bench line (with extra_configs), rocprofv3 kernel stats of one
# bench step and the FETCH_SIZE pass on the folded decode kernel (profiles/r03_final).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?" | tee $OUT/pytest.rc
tail -4 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt < /dev/null; echo "bench rc=$?"
python - <<'PY'
import json,os
o=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r03f/bench_line.json")).read().strip().splitlines()[-1])
print("bench value", o["value"], "ms/step", o["ms_per_step"], "roofline", o["roofline"]["frac"], "avg_step_us", o["roofline"].get("avg_step_us"), "stages", o["stage_ms_per_step"])
print("vocoder", o.get("vocoder_roofline", {}).get("frac"), o.get("vocoder_roofline", {}).get("useful_frac_of_bf16_peak"), "cpu", o.get("cpu_baseline", {}).get("value"))
for k,v in o.get("extra_configs",{}).items(): print(" ", k, v["value"], v["ms_per_step"], v["roofline"]["frac"], v["decode_fallbacks"], v["leg_wall_s"])
PY
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1 < /dev/null
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
head -12 $OUT/bench_kernel_stats.csv | cut -c1-170
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -o fold -- python $GRAFT_REPO_ROOT/tools/fold_traffic.py > $OUT/fold_traffic.log 2>&1 < /dev/null
f=$(find /tmp/prof_fetch -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && timeout 60 python $GRAFT_REPO_ROOT/tools/summarise_pmc.py "$f" FETCH_SIZE > $OUT/pmc_FETCH_SIZE.txt 2>&1 < /dev/null
cat $OUT/pmc_FETCH_SIZE.txt | head -5; tail -3 $OUT/fold_traffic.log
