# dw-pass tile sweep on the fused Lanczos iteration at Ns=16 and the Ns=18 product: bash scripts/sweep_dw.sh
cd $GRAFT_REPO_ROOT
for dv in 2 3; do for w in 32 64; do for r in 16 32 64; do for kb in 0 32; do
  EDGPU_DW_VARIANT=$dv EDGPU_DW_W=$w EDGPU_DW_ROWS=$r EDGPU_DW_LDS_KB=$kb python3 scripts/microbench.py 15 kron 10 lanczos
done; done; done; done
for dv in 2 3; do for w in 32 64; do for r in 16 64; do
  EDGPU_DW_VARIANT=$dv EDGPU_DW_W=$w EDGPU_DW_ROWS=$r python3 scripts/microbench.py 17 direct 2
done; done; done
