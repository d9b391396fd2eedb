# dw pass with all loads of a row step issued up front (EDGPU_DW_UNR) at Ns=18 and Ns=16
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ns18_unr
rm -rf $OUT; mkdir -p $OUT
for unr in 0 1; do for rows in 16 8; do for df in 0 2; do
  EDGPU_DW_UNR=$unr EDGPU_DW_ROWS=$rows EDGPU_DW_FIRST=$df python3 scripts/microbench.py 17 direct 5 >> $OUT/times.txt 2>&1
done; done; done
EDGPU_DW_UNR=1 EDGPU_DW_ROWS=8 EDGPU_DW_FIRST=2 EDGPU_DW_XPT=8 python3 scripts/microbench.py 17 direct 5 >> $OUT/times.txt 2>&1
EDGPU_DW_UNR=1 EDGPU_DW_ROWS=8 EDGPU_DW_FIRST=2 EDGPU_DW_W=32 python3 scripts/microbench.py 17 direct 5 >> $OUT/times.txt 2>&1
EDGPU_DW_UNR=1 EDGPU_DW_ROWS=8 EDGPU_DW_FIRST=1 python3 scripts/microbench.py 17 direct 5 >> $OUT/times.txt 2>&1
for unr in 0 1; do for rows in 16 8; do
  EDGPU_DW_UNR=$unr EDGPU_DW_ROWS=$rows python3 scripts/microbench.py 15 kron 20 lanczos >> $OUT/times.txt 2>&1
  EDGPU_DW_UNR=$unr EDGPU_DW_ROWS=$rows python3 scripts/microbench.py 15 kron 20 >> $OUT/times.txt 2>&1
done; done
EDGPU_DW_UNR=1 EDGPU_DW_ROWS=8 EDGPU_DW_FIRST=0 python3 scripts/microbench.py 17 kron 3 lanczos >> $OUT/times.txt 2>&1
EDGPU_DW_UNR=1 EDGPU_DW_ROWS=8 EDGPU_DW_FIRST=2 python3 scripts/microbench.py 17 kron 3 lanczos >> $OUT/times.txt 2>&1
cat $OUT/times.txt
for cfg in "1 8 0" "1 8 2"; do set -- $cfg
  EDGPU_DW_UNR=$1 EDGPU_DW_ROWS=$2 EDGPU_DW_FIRST=$3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$1_$2_$3 -- python3 scripts/microbench.py 17 direct 3 > $OUT/trace_$1_$2_$3.log 2>&1
  cp $(find $OUT/trace_$1_$2_$3 -name "*kernel_stats.csv" | head -1) $OUT/ns18_direct_unr$1_rows$2_df$3_kernel_stats.csv
  for pmc in FETCH_SIZE WRITE_SIZE; do
    EDGPU_DW_UNR=$1 EDGPU_DW_ROWS=$2 EDGPU_DW_FIRST=$3 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$1_$2_$3_$pmc -- python3 scripts/microbench.py 17 direct 1 > $OUT/pmc_$1_$2_$3_$pmc.log 2>&1
  done
done
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
