cd $GRAFT_REPO_ROOT
EDGPU_KRON_SPLIT=1 python -m pytest tests -m gpu -q -x -k "hxv or lanczos or known or c2_size or multirank or rccl" 2>&1 | tail -2
for t in 512 1024; do for g in 256 512; do for st in kron direct; do EDGPU_UP_THREADS=$t EDGPU_UP_GRID=$g EDGPU_KRON_SPLIT=1 EDGPU_DW_W=64 python scripts/microbench.py 15 $st 10 2>&1 | tail -1 | sed "s/^/SPLIT thr=$t grid=$g /"; done; done; done
EDGPU_DEBUG=2 EDGPU_KRON_SPLIT=1 EDGPU_DW_W=64 python scripts/microbench.py 15 kron 10 2>&1 | tail -1 | sed "s/^/passA+emptyB /"
