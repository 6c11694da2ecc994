#!/bin/bash
# Builds a second copy of libmetavoice_hip.so with decode_mega.hip compiled under extra -D flags, for A/B runs on ONE box:
#   tools/build_variant.sh noearly "-DMEGA_EARLY_W1=0 -DMEGA_EARLY_W2=0"   ->  metavoice-src_amd/lib/libmv_noearly.so
#   MV_HIP_LIB=$PWD/metavoice-src_amd/lib/libmv_noearly.so python tools/fold_sweep.py
# (compile-time tunables left in decode_mega.hip: MEGA_EARLY_W1 / MEGA_EARLY_W2 / MEGA_EW2_CHUNKS / MEGA_NU_UNROLL_P0 / MEGA_NU_UNROLL_P3; the product
#  build uses their defaults.  Round 4's measured-slower switches: git apply tools/experiments/decode_mega_r04_switches.patch first)
#   tools/build_variant.sh spread "-DT256_SPREAD=1" mbd                      (third argument: the source to recompile, default decode_mega)
set -eu
NAME=$1
FLAGS=$2
SRC=${3:-decode_mega}
cd "$(dirname "$0")/../metavoice-src_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -I../../include $FLAGS -c $SRC.hip -o "../lib/obj/${SRC}_${NAME}.oo"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls ../lib/obj/*.o | grep -v "/${SRC}.o") "../lib/obj/${SRC}_${NAME}.oo" -o "../lib/libmv_${NAME}.so"
ls -la "../lib/libmv_${NAME}.so"
