cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof2
rm -rf $OUT; mkdir -p $OUT
export EDGPU_KRON_SPLIT=1
for w in 32 128; do
export EDGPU_DW_W=$w
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_w$w -- python3 scripts/microbench.py 15 kron 5 > $OUT/trace_w$w.log 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  tag=$(echo $pmc | tr ' ' '_')
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_w${w}_$tag -- python3 scripts/microbench.py 15 kron 3 > $OUT/pmc_w${w}_$tag.log 2>&1
done
done
