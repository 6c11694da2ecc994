#!/bin/bash
# Device-ISA fingerprint of the translation units that share mega_core.h / attn_core.h: `hipcc -S --cuda-device-only`, comments and the
# compilation-unit id stripped, md5.  A source clean-up that is meant to change nothing must print the same six lines before and after
# (round 5: the removal of decode_mega.hip's measured-slower switches was checked this way).
#   tools/isa_hash.sh [csrc dir]
d=${1:-$(dirname "$0")/../metavoice-src_amd/csrc}
inc=$(dirname "$0")/../include
t=$(mktemp -d)
one() {  # name, extra flags, source
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -I"$inc" -I"$d" $2 -S --cuda-device-only "$d/$3.hip" -o "$t/$1.s" 2>"$t/$1.err" || echo "COMPILE ERROR $1"
  echo "$1 $(grep -v "^\s*;\|^\s*\.file\|^\s*\.loc\|\.ident\|^\s*\.section\s*\.debug\|__hip_cuid" "$t/$1.s" | sed 's/;.*$//' | md5sum | cut -d' ' -f1)"
}
for f in attention decode_mega engine lstm; do one $f "" $f & done
one decode_mfma "-O2 -mllvm -amdgpu-sched-strategy=max-memory-clause" decode_mfma &   # the Makefile's MFMA4_FLAGS / MFMA8_FLAGS
one decode_mfma8 "-O2 -mllvm -amdgpu-sched-strategy=max-memory-clause -mllvm -disable-machine-licm" decode_mfma8 &
one decode_mega_instr "-DMEGA_INSTR=1" decode_mega &
wait
rm -rf "$t"
