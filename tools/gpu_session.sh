#!/bin/bash
# ONE parameterised GPU session (replaces the per-call gpu_r03*.sh family): run through gpurun, every step bounded by `timeout`,
# nothing reads stdin, everything lands under gpurun_out/<tag>/.
#   tools/gpu_session.sh <tag> <step> [<step> ...]
# steps (':'-separated arguments):
#   ab:<libs>:<NU_CASES>[:n_new]   tools/nu_sweep.py once per library, twice round (A B A B) -- libs are comma-separated names: `product` or
#                                  X for metavoice-src_amd/lib/libmv_X.so (tools/build_variant.sh); cases as <format>x<utterances>:
#                                  e.g. ab:r03,product:bf16x1,bf16x4
#   fold:<libs>                    tools/fold_sweep.py (launches / one step per launch / token drawn in the launch) per library
#   tests[:<pytest -k expression>] the GPU suite (or a subset)
#   lines[:<pytest -k expression>] the figures the parity tests print (pytest -s), collected into pytest_lines.txt
#   smoke                          __graft_entry__.smoke()
#   bench[:<bench.py arguments>]   the default bench line (+ a one-line summary)
#   stamps[:<utterances>[:<pos>]]  phase stamps of the persistent kernel (instrumented library)
#   ablate                         no-weights ablation of the persistent kernel (instrumented library): the dependency chain alone
#   stats                          rocprofv3 --kernel-trace --stats of one bench step
#   fetch                          rocprofv3 --pmc FETCH_SIZE on one bench-shaped utterance (own pass)
#   sq                             rocprofv3 --pmc SQ_* on the one-step instance at the mean context (own pass)
#   py:<script>[:args]             any tools/*.py
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
LIB=$ROOT/metavoice-src_amd/lib
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
uselib() { if [ "$1" = product ]; then unset MV_HIP_LIB; else export MV_HIP_LIB=$LIB/libmv_$1.so; [ "$1" = instr ] && export MV_HIP_LIB=$LIB/libmetavoice_hip_instr.so; fi; }
for step in "$@"; do
  IFS=':' read -r kind a1 a2 a3 <<< "$step"
  echo "=== $step" | tee -a $OUT/session.log
  case $kind in
    ab)
      for round in 1 2; do for v in ${a1//,/ }; do
        uselib $v; echo "== $v (round $round)" | tee -a $OUT/ab.txt
        CASES=${a2:-bf16x1}; NU_CASES="${CASES//x/:}" timeout 600 python tools/nu_sweep.py ${a3:-1920} 2>&1 < /dev/null | grep -E "us/step|Error|error" | tee -a $OUT/ab.txt
      done; done; unset MV_HIP_LIB ;;
    fold)
      for v in ${a1//,/ }; do uselib $v; echo "== $v" | tee -a $OUT/fold.txt
        FOLD_SKIP_LAUNCHES=1 FOLD_STEPS=512 timeout 300 python tools/fold_sweep.py 2>&1 < /dev/null | grep -E "us/step|rror" | tee -a $OUT/fold.txt; done; unset MV_HIP_LIB ;;
    tests)
      if [ -n "${a1:-}" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$a1" > $OUT/pytest.log 2>&1 < /dev/null; else timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 < /dev/null; fi
      echo "pytest rc=$?" | tee $OUT/pytest.rc; tail -5 $OUT/pytest.log ;;
    lines)
      # the figures the parity tests print (-s): SNR / distance lines of the benched-size, matrix-core, int4 and front-end tests, for profiles/
      timeout 900 python -m pytest tests -m gpu -q -s -k "${a1:-benched or matrix_core or audio_front or int4 or fullsize_golden}" > $OUT/pytest_s.log 2>&1 < /dev/null
      grep -aE "\[[^]]+\]|passed|failed" $OUT/pytest_s.log | sed 's/^\.*//' | sort -u > $OUT/pytest_lines.txt; tail -3 $OUT/pytest_lines.txt ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $OUT/smoke.log 2>&1 < /dev/null; tail -4 $OUT/smoke.log ;;
    bench)
      timeout 1500 python bench.py ${a1:-} > $OUT/bench_out.txt 2> $OUT/bench_stderr.txt < /dev/null; echo "bench rc=$?"
      tail -1 $OUT/bench_out.txt > $OUT/bench_line.json
      python tools/bench_summary.py $OUT/bench_line.json 2>&1 | tee $OUT/bench_summary.txt ;;
    stamps)
      uselib instr
      MV_UTT=${a1:-1} timeout 300 python tools/mega_stamps.py ${a2:-1024} 2>&1 < /dev/null | tee $OUT/stamps_${a1:-1}utt.txt; unset MV_HIP_LIB ;;
    ablate)
      uselib instr
      for ab in 0 1; do echo "== gemv_ablate $ab" | tee -a $OUT/ablate.txt; MV_ABLATE=$ab timeout 300 python tools/mega_stamps.py 1024 2>&1 < /dev/null | tee -a $OUT/ablate.txt; done; unset MV_HIP_LIB ;;
    stats)
      ( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $OUT/bench_under_rocprof.log 2>&1 < /dev/null )
      find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
      head -12 $OUT/bench_kernel_stats.csv | cut -c1-170 ;;
    fetch)
      ( cd /tmp; timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -o fold -- python $ROOT/tools/fold_traffic.py > $OUT/fold_traffic.log 2>&1 < /dev/null )
      f=$(find /tmp/prof_fetch -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && timeout 60 python tools/summarise_pmc.py "$f" FETCH_SIZE 2>&1 < /dev/null | grep -E "^#|decode_mega" > $OUT/pmc_FETCH_SIZE.txt
      grep us_per_step $OUT/fold_traffic.log | tail -1 > $OUT/fold_traffic.json; cat $OUT/pmc_FETCH_SIZE.txt ;;
    sq)
      SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"
      ( cd /tmp; timeout 400 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d /tmp/prof_sq -o probe -- python $ROOT/tools/probe_kernels.py --kinds 6 --pos 1087 --iters 24 > $OUT/probe_SQ.log 2>&1 < /dev/null )
      f=$(find /tmp/prof_sq -name '*counter_collection.csv' | head -1)
      if [ -n "$f" ]; then for c in $SQ; do timeout 60 python tools/summarise_pmc.py "$f" $c 2>&1 < /dev/null | grep -E "^#|decode_mega"; done > $OUT/pmc_SQ.txt; fi
      cut -c1-140 $OUT/pmc_SQ.txt ;;
    py)
      timeout 900 python tools/$a1 ${a2:-} ${a3:-} 2>&1 < /dev/null | tee -a $OUT/${a1%.py}.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
