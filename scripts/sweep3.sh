cd $GRAFT_REPO_ROOT
for x in 0 1; do EDGPU_CSR_XCD=$x EDGPU_CSR_VARIANT=6 EDGPU_CSR_TILE=3 python -m pytest tests -m gpu -q -x -k "csr or c2_size or known" 2>&1 | tail -1; done
for x in 1 0; do for v in 6 5 4; do for t in 1 2 4 8 16 32 1000; do EDGPU_CSR_XCD=$x EDGPU_CSR_VARIANT=$v EDGPU_CSR_TILE=$t python scripts/microbench.py 15 csr 5 2>&1 | tail -1 | sed "s/^/XCD=$x /"; done; done; done
