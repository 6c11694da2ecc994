#!/bin/bash
# Fused two-layer codec LSTM: gate rows in registers (lstm_seq2_kernel<8>), (W_ih2 h1_t) + b between the two waits of layer 2, gin / bias
# off the step's serial chain.  A/B on one box against the previous library (metavoice-src_amd/lib/libmv_base.so = the build before the
# change) and the rows-in-LDS kernel of the same library; the waveform must not change a bit.  Then the phase stamps of the new kernel
# (tools/build_variant.sh lstmstamps "-DLSTM_STAMPS" lstm) and the codec / speaker encoder tests.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/metavoice-src_amd/lib
for v in base product lds product; do
  unset MV_HIP_LIB MV_LSTM_ROWS_IN_LDS
  if [ $v = base ]; then export MV_HIP_LIB=$LIB/libmv_base.so; fi
  if [ $v = lds ]; then export MV_LSTM_ROWS_IN_LDS=1; fi
  echo "== $v" | tee -a $OUT/codec_lstm2.txt
  timeout 300 python tools/codec_time.py 2>&1 < /dev/null | grep "decode 960" | tee -a $OUT/codec_lstm2.txt
done
unset MV_HIP_LIB MV_LSTM_ROWS_IN_LDS
MV_HIP_LIB=$LIB/libmv_lstmstamps.so timeout 300 python tools/codec_time.py 2>&1 < /dev/null | grep -E "stamps|decode 960" | tail -6 | tee $OUT/lstm_stamps2.txt
timeout 900 python -m pytest tests/test_codec_gpu.py tests/test_speaker_encoder_gpu.py -m gpu -x -q 2>&1 < /dev/null | tail -3 | tee -a $OUT/codec_lstm2.txt
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_codec -o codec -- python $GRAFT_REPO_ROOT/tools/codec_time.py > $OUT/codec_under_rocprof.log 2>&1 < /dev/null
find /tmp/prof_codec -name '*kernel_stats.csv' -exec cp {} $OUT/codec_kernel_stats2.csv \;
head -6 $OUT/codec_kernel_stats2.csv | cut -c1-160
