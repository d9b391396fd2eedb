#!/bin/bash
mkdir -p gpurun_out/r4y
python -m pytest tests/test_small_sector_lanczos.py tests/test_gf_leg.py -x -q -m gpu > gpurun_out/r4y/t11.log 2>&1; echo "exit $?" >> gpurun_out/r4y/t11.log; tail -3 gpurun_out/r4y/t11.log
python3 scripts/small_sector_bench.py > gpurun_out/r4y/small_sectors.txt 2>&1; grep -v "amdgpu.ids\|warning: environment" gpurun_out/r4y/small_sectors.txt | head -12
python3 scripts/sector_sweep.py 11 kron 1 > gpurun_out/r4y/sweep_ns12_kron.txt 2>&1; head -3 gpurun_out/r4y/sweep_ns12_kron.txt
