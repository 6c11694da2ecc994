#!/bin/bash
# Builds a second copy of libmetavoice_hip.so with decode_mega.hip compiled under extra -D flags, for A/B runs on ONE box:
#   tools/build_variant.sh noearly "-DMEGA_EARLY_W1=0 -DMEGA_EARLY_W2=0"   ->  metavoice-src_amd/lib/libmv_noearly.so
#   MV_HIP_LIB=$PWD/metavoice-src_amd/lib/libmv_noearly.so python tools/fold_sweep.py
# (compile-time switches of decode_mega.hip: MEGA_EARLY_W1 / MEGA_EARLY_W2 / MEGA_EARLY_QKV; the product build uses its defaults)
set -eu
NAME=$1
FLAGS=$2
cd "$(dirname "$0")/../metavoice-src_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -I../../include $FLAGS -c decode_mega.hip -o "../lib/obj/decode_mega_${NAME}.oo"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls ../lib/obj/*.o | grep -v '/decode_mega.o') "../lib/obj/decode_mega_${NAME}.oo" -o "../lib/libmv_${NAME}.so"
ls -la "../lib/libmv_${NAME}.so"
