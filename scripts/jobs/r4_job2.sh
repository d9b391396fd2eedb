#!/bin/bash
# round 4: full GPU suite on the rewritten long-row local pass + knob table, then Ns=18 / Ns=16 timings
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r4_t2.log 2>&1; echo "exit $?" >> gpurun_out/r4_t2.log; tail -4 gpurun_out/r4_t2.log
for st in direct kron; do python scripts/microbench.py 17 $st 5 >> gpurun_out/r4_mb2.log 2>&1; done
python scripts/microbench.py 17 direct 5 lanczos >> gpurun_out/r4_mb2.log 2>&1
python scripts/microbench.py 16 direct 5 >> gpurun_out/r4_mb2.log 2>&1
python scripts/microbench.py 16 kron 5 >> gpurun_out/r4_mb2.log 2>&1
grep nbath gpurun_out/r4_mb2.log
