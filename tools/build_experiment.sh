#!/bin/bash
# Builds a second copy of libmetavoice_hip.so with decode_mega.hip compiled under an experiment macro, for A/B runs on one box
# (the experiment itself lives as a patch under tools/experiments/, e.g. `git apply tools/experiments/decode_mega_w3_lds.patch`):
#   tools/build_experiment.sh MEGA_EXPERIMENT_W3_LDS   ->  metavoice-src_amd/lib/libmv_MEGA_EXPERIMENT_W3_LDS.so
#   MV_HIP_LIB=$PWD/metavoice-src_amd/lib/libmv_MEGA_EXPERIMENT_W3_LDS.so python tools/fold_sweep.py
# The product library (make -C metavoice-src_amd/csrc) never defines these macros and the product source carries no experiment code.
set -eu
MACRO=$1
cd "$(dirname "$0")/../metavoice-src_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -I../../include -D$MACRO -c decode_mega.hip -o "../lib/obj/decode_mega_${MACRO// /_}.oo"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls ../lib/obj/*.o | grep -v '/decode_mega.o') "../lib/obj/decode_mega_${MACRO// /_}.oo" -o "../lib/libmv_${MACRO// /_}.so"
ls -la "../lib/libmv_${MACRO// /_}.so"
