#!/bin/bash
# round 4, first GPU job: parity of the rewritten long-row local pass + Ns=18 timings
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_sector_code_paths or c4 or hxv_matches or free or fused" > gpurun_out/r4_t1.log 2>&1; echo "exit $?" >> gpurun_out/r4_t1.log; tail -5 gpurun_out/r4_t1.log
for st in direct kron; do python scripts/microbench.py 17 $st 5 >> gpurun_out/r4_mb1.log 2>&1; done
python scripts/microbench.py 17 direct 5 lanczos >> gpurun_out/r4_mb1.log 2>&1
cat gpurun_out/r4_mb1.log
