#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT
python3 scripts/small_sector_bench.py 2>&1 | grep "lanc_eigh" | cut -c1-140 > $OUT/small_stoprule.txt; cat $OUT/small_stoprule.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/gs_trace_one.py 11 5 2 > $OUT/gs_trace.log 2>&1
cat $OUT/gs_trace.log | grep -v "^W2026\|^E2026"
