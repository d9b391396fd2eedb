#!/bin/bash
# where an edgpu_eigsh call on launch-bound sectors spends its wall time (EDGPU_DEBUG=8)
OUT=gpurun_out/r4z; mkdir -p $OUT
for sec in "11 4 4" "11 6 6" "11 3 2"; do EDGPU_DEBUG=8 timeout -k 10 100 python3 scripts/eigsh_trace_one.py $sec kron 3 2>&1 | grep -v "amdgpu.ids" | tail -8; echo; done > $OUT/eigsh_where.txt; cat $OUT/eigsh_where.txt
