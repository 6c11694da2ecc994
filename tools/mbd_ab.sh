# A/B of the vocoder stage on ONE box: tools/mbd_time.py (precision 3, 3 calls each) alternately with the product library and with another build
# (MV_HIP_LIB; default metavoice-src_amd/lib/libmv_base.so = the previous commit's mbd.hip) -> gpurun_out/$1/ab.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-mbdab}; mkdir -p $OUT
BASE=${2:-$ROOT/metavoice-src_amd/lib/libmv_base.so}
cd $ROOT
for round in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export MV_HIP_LIB=$BASE; else unset MV_HIP_LIB; fi
    echo "== $lib (round $round)" >> $OUT/ab.txt
    MBD_PRECISIONS=3 MBD_REPS=3 timeout 300 python tools/mbd_time.py 2>&1 | grep tokens_to_wav >> $OUT/ab.txt
  done
done
unset MV_HIP_LIB
for upb in ${MBD_AB_UPB:-}; do
  echo "== new, MV_MBD_NORM_UPB=$upb" >> $OUT/ab.txt
  MV_MBD_NORM_UPB=$upb MBD_PRECISIONS=3 MBD_REPS=3 timeout 300 python tools/mbd_time.py 2>&1 | grep tokens_to_wav >> $OUT/ab.txt
done
cat $OUT/ab.txt
