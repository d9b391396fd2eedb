set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "hxv or csr" 2>&1 | tail -3
for v in 1 2 3; do for t in 1 4 16; do EDGPU_CSR_VARIANT=$v EDGPU_CSR_TILE=$t python scripts/microbench.py 15 csr 5 2>&1 | tail -1; done; done
for d in 0 1 2 3 4 5; do EDGPU_DEBUG=$d python scripts/microbench.py 15 kron 5 2>&1 | tail -1; done
for t in 256 512; do EDGPU_KRON_THREADS=$t python scripts/microbench.py 15 kron 5 2>&1 | tail -1; done
EDGPU_DEBUG=0 python scripts/microbench.py 15 direct 5 2>&1 | tail -1
