# per-kernel durations (+ HBM counters) of the Lanczos iteration, fused and three-pass: bash scripts/prof_lz.sh [storage]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ST=${1:-kron}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lz_$ST
rm -rf $OUT; mkdir -p $OUT
for f in 1 0; do
  export EDGPU_LZ_FUSED=$f
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f$f -- python3 scripts/microbench.py 15 $ST 10 lanczos > $OUT/trace_f$f.log 2>&1
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_f${f}_$pmc -- python3 scripts/microbench.py 15 $ST 3 lanczos > $OUT/pmc_f${f}_$pmc.log 2>&1
  done
  echo "== fused=$f"; cat $OUT/trace_f$f.log; fcsv=$(find $OUT/trace_f$f -name "*kernel_stats.csv" | head -1); head -8 "$fcsv"
done > $OUT/summary.txt 2>&1
python3 scripts/pmc_summary.py $OUT >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
