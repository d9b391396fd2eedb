#!/bin/bash
mkdir -p gpurun_out/r4g
python -m pytest tests/test_small_sector_lanczos.py tests/test_gf_leg.py tests/test_gpu_parity.py tests/test_eigsh.py -x -q -m gpu > gpurun_out/r4g/t18.log 2>&1; echo "exit $?" >> gpurun_out/r4g/t18.log; tail -4 gpurun_out/r4g/t18.log
for sr in 1 0; do EDGPU_LZ_SR=$sr python3 scripts/small_sector_bench.py > gpurun_out/r4g/small_sr$sr.txt 2>&1; echo "EDGPU_LZ_SR=$sr"; grep -v "amdgpu.ids\|warning: environment" gpurun_out/r4g/small_sr$sr.txt | head -16 | cut -c1-200; done
python3 scripts/sector_sweep.py 11 kron 1 > gpurun_out/r4g/sweep_ns12_kron.txt 2>&1; head -3 gpurun_out/r4g/sweep_ns12_kron.txt
