# dw-pass tile sweep of the Ns=18 DIRECT product: bash scripts/sweep_dw18.sh
cd $GRAFT_REPO_ROOT
for w in 32 64 128; do for r in 16 32 64; do for kb in 0 16 32 48; do
  EDGPU_DW_W=$w EDGPU_DW_ROWS=$r EDGPU_DW_LDS_KB=$kb python3 scripts/microbench.py 17 direct 2
done; done; done
