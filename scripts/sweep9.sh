cd $GRAFT_REPO_ROOT
EDGPU_CSR_SPLIT=1 python -m pytest tests -m gpu -q -x -k "csr or hxv or lanczos or known or c2_size or multirank or rccl" 2>&1 | tail -2
for nt in 1 0; do for r in 64 32 16; do EDGPU_CSRD_NT=$nt EDGPU_CSRD_ROWS=$r python scripts/microbench.py 15 csr 10 2>&1 | tail -1 | sed "s/^/SPLITCSR nt=$nt rows=$r /"; done; done
EDGPU_CSR_SPLIT=0 python scripts/microbench.py 15 csr 10 2>&1 | tail -1 | sed "s/^/UNIFIED /"
python bench.py --no-extra --no-cpu > gpurun_out/bench_d.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_d.json')); print('bench split csr', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['roofline']['frac'])"
