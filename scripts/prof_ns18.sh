# per-kernel durations of the Ns=18 DIRECT product, segment vs gather form of the local pass: bash scripts/prof_ns18.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ns18
rm -rf $OUT; mkdir -p $OUT
for sg in 1 0; do
  export EDGPU_UP_SEG=$sg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_s$sg -- python3 scripts/microbench.py 17 direct 3 > $OUT/trace_s$sg.log 2>&1
  echo "== seg=$sg"; grep nbath $OUT/trace_s$sg.log; f=$(find $OUT/trace_s$sg -name "*kernel_stats.csv" | head -1); head -6 "$f"
done
for sg in 1 0; do
  export EDGPU_UP_SEG=$sg
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_s${sg}_$pmc -- python3 scripts/microbench.py 17 direct 1 > $OUT/pmc_s${sg}_$pmc.log 2>&1
  done
done
python3 scripts/pmc_summary.py $OUT
