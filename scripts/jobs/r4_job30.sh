#!/bin/bash
# experiment: tile width x row order of the dw pass at c3 (Ns=16), stored factors and on the fly, fused iteration and product
L=gpurun_out/r4z/dw_order_w.txt; mkdir -p gpurun_out/r4z; : > $L
for rep in 1 2; do
for w in 64 32; do for o in 0 1; do
  for st in kron direct; do
    EDGPU_DW_W=$w EDGPU_DW_ORDER=$o python scripts/microbench.py 15 $st 30 lanczos 2>/dev/null >> $L
  done
done; done; done
cat $L
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 64 32; do for o in 0 1; do
  EDGPU_DW_W=$w EDGPU_DW_ORDER=$o rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r4z/pmc_w${w}_o${o}_FETCH_SIZE -- python3 scripts/microbench.py 15 kron 3 lanczos > /dev/null 2>&1
  EDGPU_DW_W=$w EDGPU_DW_ORDER=$o rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4z/kt_w${w}_o${o} -- python3 scripts/microbench.py 15 kron 5 lanczos > /dev/null 2>&1
  echo "w=$w order=$o"; grep -E "k_dwpass2|k_uppass2_ax" gpurun_out/r4z/kt_w${w}_o${o}/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done; done
python3 scripts/pmc_summary.py gpurun_out/r4z 2>/dev/null | grep -A3 "pmc_w" | grep -E "pmc_w|k_dwpass2"
