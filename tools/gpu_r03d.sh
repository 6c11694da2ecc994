#!/bin/bash
# Round 3, fourth GPU call: MFMA whole-row attention (prefill / second stage) -- parity + the engine suites that run through it;
# PMC passes on the 256 x 256 MBD GEMM (what bounds it at ~48 % of the MFMA peak?); prefill time before / after.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_stage2_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_fullsize_golden_gpu.py tests/test_tts_gpu.py -x -q -s > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"
tail -4 $OUT/pytest.log; grep -E "^\.?\[long24.*free-running" $OUT/pytest.log | cut -c1-220
python - <<'PY' 2>&1 | tee $OUT/prefill_time.txt
import importlib, os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
mv = importlib.import_module("metavoice-src_amd"); fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
e = mv.Stage1Engine(args, max_utterances=1); e.load_state_dict(fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))); e.finalize()
spk, sp = mv.synthetic.speaker_embedding(0), mv.SamplingParams(top_p=0.95, seed=1)
for T in (128, 512, 1100, 2016):
    p = mv.synthetic.prompt_tokens(T, seed=0)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.begin([p], spk, sp, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=1)   # mv_begin = upload + prefill + first token, synchronous
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"mv_begin with MFMA prefill, T={T}: {min(ts):.2f} ms (runs {[round(t, 2) for t in ts]})")
PY
export TMPDIR=/tmp
cd /tmp
echo "== available counters (MFMA / LDS / wait)" > $OUT/pmc_t256.txt
timeout 120 rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(SQ_[A-Z0-9_]*(MFMA|LDS|WAIT|BUSY|ACTIVE|VMEM|INSTS)[A-Z0-9_]*|TCC_[A-Z_]*(HIT|MISS|REQ)[A-Z0-9_\[\]]*|TCP_[A-Z_]*(REQ|READ|PENDING)[A-Z0-9_]*)\b" | sort -u | tr '\n' ' ' >> $OUT/pmc_t256.txt
echo >> $OUT/pmc_t256.txt
run_pmc() {  # $1 = tag, rest = counters
  tag=$1; shift
  rm -rf /tmp/prof_pmc_$tag
  MBD_PRECISIONS=3 MBD_REPS=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/prof_pmc_$tag -o mbd -- python $GRAFT_REPO_ROOT/tools/mbd_time.py > $OUT/pmc_$tag.log 2>&1 < /dev/null
  f=$(find /tmp/prof_pmc_$tag -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then for c in "$@"; do timeout 60 python $GRAFT_REPO_ROOT/tools/summarise_pmc.py "$f" $c 2>&1 < /dev/null | grep -E "^#|t256|glds3" | cut -c1-150; done >> $OUT/pmc_t256.txt; else echo "pass $tag produced no counters" >> $OUT/pmc_t256.txt; tail -5 $OUT/pmc_$tag.log >> $OUT/pmc_t256.txt; fi
}
run_pmc a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
run_pmc b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
run_pmc c TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum FETCH_SIZE
cat $OUT/pmc_t256.txt | cut -c1-400
