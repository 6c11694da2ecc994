#!/bin/bash
# Round 3 end state: smoke, the whole GPU suite, the default bench line (with extra_configs), rocprofv3 kernel stats of one bench step,
# FETCH_SIZE pass on the folded decode kernel, SQ pass on the one-step instance (profiles/r03_final).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_final
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $OUT/smoke.log 2>&1 < /dev/null; tail -6 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?" | tee $OUT/pytest.rc
tail -4 $OUT/pytest.log
timeout 1200 python bench.py > $OUT/bench_out.txt 2> $OUT/bench_stderr.txt < /dev/null; echo "bench rc=$?"
tail -1 $OUT/bench_out.txt > $OUT/bench_line.json
python - <<'PY'
import json,os
o=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r03_final/bench_line.json")).read())
print("bench value", o["value"], "ms/step", o["ms_per_step"], "roofline", o["roofline"]["frac"], "avg_step_us", o["roofline"].get("avg_step_us"), "stages", o["stage_ms_per_step"])
print("vocoder", o.get("vocoder_roofline", {}).get("frac"), o.get("vocoder_roofline", {}).get("useful_frac_of_bf16_peak"), "cpu", o.get("cpu_baseline", {}).get("value"), "stdout lines", sum(1 for _ in open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r03_final/bench_out.txt"))))
for k,v in o.get("extra_configs",{}).items(): print(" ", k, v["value"], v["ms_per_step"], v["roofline"]["frac"], v["step_roofline_frac"], v["decode_fallbacks"], v["leg_wall_s"])
PY
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1 < /dev/null
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
head -8 $OUT/bench_kernel_stats.csv | cut -c1-170
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -o fold -- python $GRAFT_REPO_ROOT/tools/fold_traffic.py > $OUT/fold_traffic.log 2>&1 < /dev/null
f=$(find /tmp/prof_fetch -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && timeout 60 python $GRAFT_REPO_ROOT/tools/summarise_pmc.py "$f" FETCH_SIZE 2>&1 < /dev/null | grep -E "^#|decode_mega" > $OUT/pmc_FETCH_SIZE.txt
grep us_per_step $OUT/fold_traffic.log | tail -1 > $OUT/fold_traffic.json
cat $OUT/pmc_FETCH_SIZE.txt
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"
timeout 400 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d /tmp/prof_sq -o probe -- python $GRAFT_REPO_ROOT/tools/probe_kernels.py --kinds 6 --pos 1087 --iters 24 > $OUT/probe_SQ.log 2>&1 < /dev/null
f=$(find /tmp/prof_sq -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then for c in $SQ; do timeout 60 python $GRAFT_REPO_ROOT/tools/summarise_pmc.py "$f" $c 2>&1 < /dev/null | grep -E "^#|decode_mega" ; done > $OUT/pmc_SQ.txt; fi
cat $OUT/pmc_SQ.txt | cut -c1-140
