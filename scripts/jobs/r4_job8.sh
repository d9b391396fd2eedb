#!/bin/bash
# round 4: full GPU suite (current build), multirank file twice more (flake hunt), default bench line, sweeps, small-sector bench, Davidson with a smaller space
mkdir -p gpurun_out/r4v
python -m pytest tests -x -q -m gpu > gpurun_out/r4v/t8.log 2>&1; echo "exit $?" >> gpurun_out/r4v/t8.log; tail -4 gpurun_out/r4v/t8.log
for i in 1 2; do timeout -k 10 500 python -m pytest tests/test_multirank.py -x -q -m gpu > gpurun_out/r4v/t8_mr$i.log 2>&1; echo "multirank run $i exit $?" | tee -a gpurun_out/r4v/t8_mr$i.log; done
python3 bench.py > gpurun_out/r4v/bench_default.json 2> gpurun_out/r4v/bench_default.err; python3 -c "
import json; d=json.load(open('gpurun_out/r4v/bench_default.json')); print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'],'frac_prod_only',d['roofline']['frac_counting_product_bytes_only'],'sustained',d.get('sustained'),'c4',d.get('c4_direct_product_ms'),'c5',d.get('c5_kron_iteration_ms'),'eigsh',d.get('eigsh_c3_s'))"
python3 scripts/sector_sweep.py 11 kron 1 > gpurun_out/r4v/sweep_ns12_kron.txt 2>&1; head -3 gpurun_out/r4v/sweep_ns12_kron.txt
python3 scripts/sector_sweep.py 7 kron 1 > gpurun_out/r4v/sweep_ns8_kron.txt 2>&1; head -3 gpurun_out/r4v/sweep_ns8_kron.txt
python3 scripts/small_sector_bench.py > gpurun_out/r4v/small_sectors.txt 2>&1; cat gpurun_out/r4v/small_sectors.txt | grep -v "amdgpu.ids\|warning: environment"
