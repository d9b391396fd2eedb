cd $GRAFT_REPO_ROOT
EDGPU_KRON_SPLIT=1 python -m pytest tests -m gpu -q -x -k "hxv or lanczos or known or c2_size or multirank or rccl" 2>&1 | tail -2
for w in 8 16 32 64 128; do EDGPU_KRON_SPLIT=1 EDGPU_DW_W=$w python scripts/microbench.py 15 kron 10 2>&1 | tail -1 | sed "s/^/SPLIT W=$w /"; done
EDGPU_DEBUG=2 python scripts/microbench.py 15 kron 10 2>&1 | tail -1
python scripts/microbench.py 15 kron 10 2>&1 | tail -1
