# Local pass for rows beyond LDS (k_uppass_halo, Ns=18): timing ablations of the hop loop and SQ / LDS counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_halo
rm -rf $OUT; mkdir -p $OUT
for d in 0 16 32 48; do
  EDGPU_DEBUG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$d -- python3 scripts/microbench.py 17 direct 3 > $OUT/trace_$d.log 2>&1
  echo "EDGPU_DEBUG=$d" >> $OUT/ablation.txt
  grep "k_uppass_halo\|k_dwpass2" $(find $OUT/trace_$d -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 >> $OUT/ablation.txt
done
cat $OUT/ablation.txt
i=0
for pmc in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES SQ_INSTS_SMEM" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$i -- python3 scripts/microbench.py 17 direct 1 > $OUT/pmc_$i.log 2>&1
done
python3 scripts/pmc_summary.py $OUT | grep -A9 "k_uppass_halo" > $OUT/halo_counters.txt
cat $OUT/halo_counters.txt
