#!/bin/bash
# round 4: blocked dw walk (parity, Ns=18 and Ns=16 timings on/off), Davidson with GD+1 restarts and the fused step
mkdir -p gpurun_out/r4u
python -m pytest tests/test_gpu_parity.py tests/test_eigsh.py tests/test_analytic_gpu.py -x -q -m gpu -k "large_sector_code_paths or c4 or davidson or free_fermion_e0_c4 or fused_iteration_equals" > gpurun_out/r4u/t7.log 2>&1; echo "exit $?" >> gpurun_out/r4u/t7.log; tail -4 gpurun_out/r4u/t7.log
grep -q "exit 0" gpurun_out/r4u/t7.log || exit 1
for b in -1 0; do
  echo "EDGPU_DW_BLOCKED=$b" >> gpurun_out/r4u/ns18.txt
  EDGPU_DW_BLOCKED=$b python3 scripts/microbench.py 17 direct 5 >> gpurun_out/r4u/ns18.txt 2>&1
  EDGPU_DW_BLOCKED=$b python3 scripts/microbench.py 17 direct 5 lanczos >> gpurun_out/r4u/ns18.txt 2>&1
done
for b in 1 0; do
  echo "EDGPU_DW_BLOCKED=$b" >> gpurun_out/r4u/ns16.txt
  EDGPU_DW_BLOCKED=$b python3 scripts/microbench.py 15 direct 10 >> gpurun_out/r4u/ns16.txt 2>&1
  EDGPU_DW_BLOCKED=$b python3 scripts/microbench.py 15 direct 10 lanczos >> gpurun_out/r4u/ns16.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r4u/ns18.txt gpurun_out/r4u/ns16.txt
python3 scripts/eig_time.py 15 kron 2 > gpurun_out/r4u/eig_time_c3.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4u/eig_time_c3.txt | tail -4
EDGPU_DVD_FUSED=0 python3 scripts/eig_time.py 15 kron 2 > gpurun_out/r4u/eig_time_c3_unfused.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4u/eig_time_c3_unfused.txt | tail -2
