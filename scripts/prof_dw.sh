# HBM-side bytes and L2 hit rate of the dw pass for two tile widths (Ns=16 Lanczos iteration): bash scripts/prof_dw.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_dw
rm -rf $OUT; mkdir -p $OUT
for w in 32 64; do
  export EDGPU_DW_W=$w
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    tag=$(echo $pmc | tr ' ' '_')
    rocprofv3 --pmc $pmc --output-format csv -d $OUT/w${w}_$tag -- python3 scripts/microbench.py 15 kron 3 lanczos > $OUT/w${w}_$tag.log 2>&1
  done
done
python3 scripts/pmc_summary.py $OUT | grep -E "==|dwpass"
