# FETCH_SIZE and L2 hit / miss of decode_mfma_kernel<8> (engine of 8 utterances, short contexts: K/V is small next to the weights) -> gpurun_out/$1/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-pmc}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  tag=$(echo $C | tr ' ' '_')
  NU_CASES=${NU_CASES:-bf16:8} timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$tag -o run -- python $ROOT/tools/nu_sweep.py ${NSTEPS:-48} > $OUT/run_$tag.log 2>&1 < /dev/null
  f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
  for c in $C; do python $ROOT/tools/summarise_pmc.py $f $c | head -4; done | tee $OUT/pmc_$tag.txt
done
