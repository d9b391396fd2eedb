#!/bin/bash
# ground-state search with the next chunk issued ahead: full GPU suite, then the small-sector table and one traced search
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/suite_final2.log 2>&1; rc=$?; tail -6 $OUT/suite_final2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 scripts/small_sector_bench.py > $OUT/small_sectors2.txt 2>&1; grep "lanc_eigh" $OUT/small_sectors2.txt | cut -c1-150
EDGPU_DEBUG=8 timeout -k 10 100 python3 scripts/gs_trace_one.py 11 5 2 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/gs_trace2.txt
