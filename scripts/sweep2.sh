cd $GRAFT_REPO_ROOT
for v in 3 4 5 6 2 7; do EDGPU_CSR_VARIANT=$v python -m pytest tests -m gpu -q -x -k "csr or c2_size or known" 2>&1 | tail -1; done
for v in 3 4 5 6 2 7; do for t in 1 4 16 64; do EDGPU_CSR_VARIANT=$v EDGPU_CSR_TILE=$t python scripts/microbench.py 15 csr 5 2>&1 | tail -1; done; done
