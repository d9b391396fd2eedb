#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT
EDGPU_DEBUG=8 python3 scripts/gs_trace_one.py 11 5 2 2>&1 | grep -v amdgpu.ids | tail -2
EDGPU_DEBUG=8 python3 scripts/gs_trace_one.py 9 5 5 2>&1 | grep -v amdgpu.ids | tail -2
python3 scripts/small_sector_bench.py 2>&1 | grep "lanc_eigh" | cut -c1-140 > $OUT/small_stoprule.txt; cat $OUT/small_stoprule.txt
