# rocprofv3 passes for the c3 kernels (development aid). Outputs under gpurun_out/prof1/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof1
mkdir -p $OUT
EDGPU_CSR_TILE=16 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_csr -- python3 scripts/microbench.py 15 csr 5 > $OUT/trace_csr.log 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  tag=$(echo $pmc | tr ' ' '_')
  EDGPU_CSR_TILE=16 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_csr_$tag -- python3 scripts/microbench.py 15 csr 3 > $OUT/pmc_csr_$tag.log 2>&1
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_kron_$tag -- python3 scripts/microbench.py 15 kron 3 > $OUT/pmc_kron_$tag.log 2>&1
done
find $OUT -name "*.csv" | head -50
