cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof5
rm -rf $OUT; mkdir -p $OUT
export EDGPU_CSR_PERSIST=0
for x in 0 1; do
export EDGPU_CSR_XCD=$x
python scripts/microbench.py 15 csr 10 2>&1 | tail -1 | sed "s/^/XCD=$x /"
for pmc in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pmc | tr ' ' '_')
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_x${x}_$tag -- python3 scripts/microbench.py 15 csr 3 > /dev/null 2>&1
done
done
python scripts/pmc_summary.py $OUT
