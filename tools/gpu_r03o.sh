#!/bin/bash
# Single-layer persistent LSTM (speaker encoder, H = 256) with the gate rows in registers vs in LDS; then the LSTM users' tests.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03o
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for v in lds product lds product; do
  unset MV_LSTM_ROWS_IN_LDS
  if [ $v = lds ]; then export MV_LSTM_ROWS_IN_LDS=1; fi
  echo "== $v" | tee -a $OUT/spkenc.txt
  timeout 300 python tools/spkenc_time.py 2>&1 < /dev/null | grep "speaker encoder" | tee -a $OUT/spkenc.txt
done
unset MV_LSTM_ROWS_IN_LDS
timeout 900 python -m pytest tests/test_codec_gpu.py tests/test_speaker_encoder_gpu.py tests/test_mbd_gpu.py -m gpu -x -q 2>&1 < /dev/null | tail -3 | tee -a $OUT/spkenc.txt
