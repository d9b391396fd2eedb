cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "csr or known or c2_size or hxv_matches or multirank or rccl or lanczos" 2>&1 | tail -2
EDGPU_CSR_SPLIT=1 python -m pytest tests -m gpu -q -x -k "csr or known or c2_size or hxv_matches" 2>&1 | tail -1
EDGPU_CSR_SPLIT=0 python -m pytest tests -m gpu -q -x -k "csr or known or c2_size or hxv_matches" 2>&1 | tail -1
for g in 2048 4096 8192; do for nt in 1 0; do EDGPU_CSR_PGRID=$g EDGPU_CSRD_NT=$nt python scripts/microbench.py 15 csr 10 2>&1 | tail -1 | sed "s/^/PERSIST split grid=$g nt=$nt /"; done; done
for g in 2048 4096; do EDGPU_CSR_SPLIT=0 EDGPU_CSR_PGRID=$g python scripts/microbench.py 15 csr 10 2>&1 | tail -1 | sed "s/^/PERSIST unified grid=$g /"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -- python3 scripts/microbench.py 15 csr 5 > /dev/null 2>&1
cat gpurun_out/prof4/*/*kernel_stats.csv | cut -c1-150 | head -5
