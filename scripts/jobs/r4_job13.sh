#!/bin/bash
mkdir -p gpurun_out/r4z
python -m pytest tests/test_small_sector_lanczos.py tests/test_gf_leg.py tests/test_gpu_parity.py -x -q -m gpu -k "workgroup or gf_leg or lanczos or fused or sweep" > gpurun_out/r4z/t13.log 2>&1; echo "exit $?" >> gpurun_out/r4z/t13.log; tail -3 gpurun_out/r4z/t13.log
python3 scripts/small_sector_bench.py > gpurun_out/r4z/small_sectors.txt 2>&1; grep -v "amdgpu.ids\|warning: environment" gpurun_out/r4z/small_sectors.txt | head -8
python3 scripts/sector_sweep.py 11 kron 1 > gpurun_out/r4z/sweep_ns12_kron.txt 2>&1; head -3 gpurun_out/r4z/sweep_ns12_kron.txt
python3 scripts/sector_sweep.py 7 kron 1 > gpurun_out/r4z/sweep_ns8_kron.txt 2>&1; head -3 gpurun_out/r4z/sweep_ns8_kron.txt
