#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_codec_gpu.py tests/test_speaker_encoder_gpu.py tests/test_mbd_gpu.py tests/test_tts_gpu.py -x -q -s > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"
tail -4 $OUT/pytest.log
MBD_PRECISIONS=3 MBD_REPS=3 timeout 300 python tools/mbd_time.py 2>&1 < /dev/null | grep tokens_to_wav | tee $OUT/mbd_time.txt
