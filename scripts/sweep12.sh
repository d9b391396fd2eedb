cd $GRAFT_REPO_ROOT
EDGPU_KRON_SPLIT=1 python -m pytest tests -m gpu -q -x -k "hxv or lanczos or known or c2_size or large_sector" 2>&1 | tail -2
for pr in 1 0; do EDGPU_UP_PAIRS=$pr python scripts/microbench.py 15 kron 20 2>&1 | tail -1 | sed "s/^/pairs=$pr /"; done
for g in 256 512; do EDGPU_UP_GRID=$g python scripts/microbench.py 15 kron 20 2>&1 | tail -1 | sed "s/^/pairs grid=$g /"; done
python bench.py --storage kron --no-extra --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('kron bench', d['ms_per_step'], d['roofline']['kernel_ms'], d['iters_per_sec'])"
