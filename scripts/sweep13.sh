cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for st in csr kron direct; do python bench.py --storage $st --no-extra --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$st bench', d['ms_per_step'], d['roofline']['kernel_ms'], d['iters_per_sec'], d['value'])"; done
