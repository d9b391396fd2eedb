cd $GRAFT_REPO_ROOT
python bench.py --no-extra > gpurun_out/bench_a.json 2>gpurun_out/bench_a.err; python -c "
import json; d=json.load(open('gpurun_out/bench_a.json')); print('csr v6', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['cpu_baseline'])"
for v in 5 4 3; do EDGPU_CSR_VARIANT=$v python bench.py --no-extra --no-cpu > gpurun_out/bench_b.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_b.json')); print('csr variant $v', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'])"; done
for st in kron direct; do python bench.py --storage $st --no-extra --no-cpu > gpurun_out/bench_c.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_c.json')); print('$st', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['iters_per_sec'])"; done
