# Ns=18 product on one GPU: dw-first (write-only dw pass) against the local-first order, timing + FETCH/WRITE counters.
# Usage (GPU box): bash scripts/ns18_dwfirst.sh ; output gpurun_out/ns18_dwfirst/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ns18_dwfirst
rm -rf $OUT; mkdir -p $OUT
for df in 0 1 2; do
  for w in 64 32; do
    EDGPU_DW_FIRST=$df EDGPU_DW_W=$w python3 scripts/microbench.py 17 direct 5 >> $OUT/times.txt 2>&1
  done
done
EDGPU_DW_FIRST=2 python3 scripts/microbench.py 17 kron 3 >> $OUT/times.txt 2>&1
EDGPU_DW_FIRST=0 python3 scripts/microbench.py 17 kron 3 lanczos >> $OUT/times.txt 2>&1
EDGPU_DW_FIRST=2 python3 scripts/microbench.py 17 kron 3 lanczos >> $OUT/times.txt 2>&1
EDGPU_DW_FIRST=2 python3 scripts/microbench.py 17 direct 3 lanczos >> $OUT/times.txt 2>&1
cat $OUT/times.txt
for df in 0 2; do
  EDGPU_DW_FIRST=$df rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_df$df -- python3 scripts/microbench.py 17 direct 3 > $OUT/trace_df$df.log 2>&1
  cp $(find $OUT/trace_df$df -name "*kernel_stats.csv" | head -1) $OUT/ns18_direct_df${df}_kernel_stats.csv
  for pmc in FETCH_SIZE WRITE_SIZE; do
    EDGPU_DW_FIRST=$df rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_df${df}_$pmc -- python3 scripts/microbench.py 17 direct 1 > $OUT/pmc_df${df}_$pmc.log 2>&1
  done
done
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
