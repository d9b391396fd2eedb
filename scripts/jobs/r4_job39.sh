#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT
EDGPU_DEBUG=8 python3 scripts/gs_trace_one.py 11 5 2 > $OUT/gs_dbg.log 2>&1; grep -v amdgpu.ids $OUT/gs_dbg.log
EDGPU_DEBUG=8 python3 scripts/gs_trace_one.py 11 6 6 > $OUT/gs_dbg2.log 2>&1; grep -v amdgpu.ids $OUT/gs_dbg2.log
