#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_stage2_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_fullsize_golden_gpu.py tests/test_int4_gpu.py -x -q > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"
tail -4 $OUT/pytest.log
cat > /tmp/prefill_time.py <<'PY'
import importlib, os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
mv = importlib.import_module("metavoice-src_amd"); fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
args = mv.ModelArgs.from_name("metavoice-1B")
cfgk = dict(n_layer=args.n_layer, n_head=args.n_head, dim=args.dim, vocab_size=args.vocab_size, block_size=args.block_size)
e = mv.Stage1Engine(args, max_utterances=1); e.load_state_dict(fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=0, device="cuda", **cfgk))); e.finalize()
spk, sp = mv.synthetic.speaker_embedding(0), mv.SamplingParams(top_p=0.95, seed=1)
for T in (50, 128, 256, 512, 1100):
    p = mv.synthetic.prompt_tokens(T, seed=0)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.begin([p], spk, sp, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=1)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"mv_begin with MFMA prefill, T={T}: {min(ts):.2f} ms (runs {[round(t, 2) for t in ts]})")
PY
echo "== split-K" | tee $OUT/prefill_time.txt
timeout 200 python /tmp/prefill_time.py 2>&1 < /dev/null | grep mv_begin | tee -a $OUT/prefill_time.txt
echo "== one range (MV_GEMM_NO_SPLITK)" | tee -a $OUT/prefill_time.txt
MV_GEMM_NO_SPLITK=1 timeout 200 python /tmp/prefill_time.py 2>&1 < /dev/null | grep mv_begin | tee -a $OUT/prefill_time.txt
