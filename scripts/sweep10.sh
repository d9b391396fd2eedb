cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
EDGPU_CSR_SPLIT=1 python -m pytest tests -m gpu -q -x -k "csr or known or c2_size" 2>&1 | tail -1
for nt in 1 0; do EDGPU_CSRD_NT=$nt python scripts/microbench.py 15 csr 10 2>&1 | tail -1 | sed "s/^/SPLITCSR small nt=$nt /"; done
EDGPU_CSR_SMALL=0 python scripts/microbench.py 15 csr 10 2>&1 | tail -1 | sed "s/^/SPLITCSR big /"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -- python3 scripts/microbench.py 15 csr 5 > /dev/null 2>&1
cat gpurun_out/prof3/*/*kernel_stats.csv | cut -c1-150 | head -5
