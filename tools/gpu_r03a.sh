#!/bin/bash
# Round 3, first GPU call: (1) the whole GPU suite (new: long-context goldens, token draw inside the launch for 2 / 4 utterances and
# int8), (2) A/B of the early-row variants of the persistent decode kernel on ONE box, (3) fold on / off for batches and quantised
# formats, (4) phase stamps of the product kernel, (5) the default bench line with its extra_configs.  Everything bounded by timeout.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
timeout 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?" | tee $OUT/pytest.rc
tail -5 $OUT/pytest.log
grep -E "^\[(full24|long24)" $OUT/pytest.log > $OUT/golden_lines.txt
export FOLD_STEPS=512
for v in product noearly w1only w2only w12qkv w2qkv product; do
  if [ $v = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$v.so; fi
  echo "== $v" | tee -a $OUT/variants.txt
  FOLD_SKIP_LAUNCHES=$([ $v = noearly ] && echo "" || echo 1) timeout 200 python tools/fold_sweep.py 2>&1 < /dev/null | grep "us/step" | tee -a $OUT/variants.txt
done
unset MV_HIP_LIB
timeout 400 python tools/nu_sweep.py 2>&1 < /dev/null | grep "us/step" | tee $OUT/nu_sweep.txt
timeout 100 python tools/mega_stamps.py 1024 2>&1 < /dev/null | grep -E "us/step|fine" | cut -c1-1600 | tee $OUT/stamps_product.txt
MV_HIP_LIB=$LIB/libmv_w12qkv.so timeout 100 python tools/mega_stamps.py 1024 2>&1 < /dev/null | grep -E "us/step|fine" | cut -c1-1600 | tee $OUT/stamps_w12qkv.txt
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt < /dev/null; echo "bench rc=$?"
python - <<'PY'
import json,os
p=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r03a/bench_line.json")
try:
    o=json.loads(open(p).read().strip().splitlines()[-1])
    print("bench value", o["value"], "ms/step", o["ms_per_step"], "roofline", o["roofline"]["frac"], "stages", o["stage_ms_per_step"])
    for k,v in o.get("extra_configs",{}).items(): print(" ", k, v["value"], v["ms_per_step"], v["roofline"]["frac"], v["decode_fallbacks"], v["leg_wall_s"])
except Exception as e: print("bench parse failed", e); print(open(p.replace("bench_line.json","bench_stderr.txt")).read()[-3000:])
PY
