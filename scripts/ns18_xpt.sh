# Ns=18 dw pass: XCDs per column tile (EDGPU_DW_XPT), local pass first; timing, then FETCH/WRITE of the candidates
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ns18_xpt
rm -rf $OUT; mkdir -p $OUT
for x in 1 2 4 8; do
  for w in 64 128; do
    EDGPU_DW_FIRST=0 EDGPU_DW_XPT=$x EDGPU_DW_W=$w python3 scripts/microbench.py 17 direct 5 >> $OUT/times.txt 2>&1
  done
done
EDGPU_DW_FIRST=0 EDGPU_DW_XPT=8 EDGPU_DW_W=64 EDGPU_DW_ROWS=32 python3 scripts/microbench.py 17 direct 5 >> $OUT/times.txt 2>&1
EDGPU_DW_FIRST=2 EDGPU_DW_XPT=8 EDGPU_DW_W=64 python3 scripts/microbench.py 17 direct 5 >> $OUT/times.txt 2>&1
cat $OUT/times.txt
for x in 2 8; do
  for pmc in FETCH_SIZE WRITE_SIZE; do
    EDGPU_DW_FIRST=0 EDGPU_DW_XPT=$x rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_x${x}_$pmc -- python3 scripts/microbench.py 17 direct 1 > $OUT/pmc_x${x}_$pmc.log 2>&1
  done
  EDGPU_DW_FIRST=0 EDGPU_DW_XPT=$x rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_x$x -- python3 scripts/microbench.py 17 direct 3 > $OUT/trace_x$x.log 2>&1
  cp $(find $OUT/trace_x$x -name "*kernel_stats.csv" | head -1) $OUT/ns18_direct_x${x}_kernel_stats.csv
done
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
