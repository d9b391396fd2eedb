cd $GRAFT_REPO_ROOT
for dv in 2 3; do EDGPU_DW_VARIANT=$dv EDGPU_KRON_SPLIT=1 python -m pytest tests -m gpu -q -x -k "hxv or lanczos or known or c2_size or multirank or rccl" 2>&1 | tail -1; done
for dv in 2 3; do for w in 16 32 64; do for r in 16 64; do EDGPU_DW_VARIANT=$dv EDGPU_DW_ROWS=$r EDGPU_KRON_SPLIT=1 EDGPU_DW_W=$w python scripts/microbench.py 15 kron 10 2>&1 | tail -1 | sed "s/^/SPLIT dv=$dv W=$w R=$r /"; done; done; done
