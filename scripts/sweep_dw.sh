# dw-pass tile sweep on the fused Lanczos iteration at Ns=16: bash scripts/sweep_dw.sh [storage]
cd $GRAFT_REPO_ROOT
ST=${1:-kron}
for w in 16 32 64 128; do for r in 16 32 64; do for kb in 0 32 48 80; do
  EDGPU_DW_W=$w EDGPU_DW_ROWS=$r EDGPU_DW_LDS_KB=$kb python3 scripts/microbench.py 15 $ST 10 lanczos
done; done; done
