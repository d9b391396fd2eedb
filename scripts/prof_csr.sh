# kernel durations + counters of the CSR product kernels, old and new forms in one call (same box): bash scripts/prof_csr.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_csr
rm -rf $OUT; mkdir -p $OUT
for cfg in "A EDGPU_CSR_LDS=0 EDGPU_CSRD_PIPE=0" "B EDGPU_CSR_LDS=1 EDGPU_CSRD_PIPE=1 EDGPU_CSR_LDS_DEPTH=2"; do
  set -- $cfg; tag=$1; shift
  export EDGPU_CSR_LDS=0 EDGPU_CSRD_PIPE=0 EDGPU_CSR_LDS_DEPTH=2
  for kv in "$@"; do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 scripts/microbench.py 15 csr 5 > $OUT/trace_$tag.log 2>&1
  i=0
  for pmc in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES SQ_INSTS_SMEM" "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_${tag}_$i -- python3 scripts/microbench.py 15 csr 3 > $OUT/pmc_${tag}_$i.log 2>&1
  done
done
python3 scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
for t in A B; do echo "== trace $t"; f=$(find $OUT/trace_$t -name "*kernel_stats.csv" | head -1); head -8 "$f"; done >> $OUT/summary.txt
cat $OUT/summary.txt
