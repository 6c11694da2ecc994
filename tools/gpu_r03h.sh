#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_codec_gpu.py tests/test_tts_gpu.py tests/test_mbd_gpu.py -x -q -s > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"
tail -3 $OUT/pytest.log
timeout 200 python tools/codec_time.py 2>&1 < /dev/null | grep -v amdgpu.ids | tee $OUT/codec_time.txt
