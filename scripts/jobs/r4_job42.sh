#!/bin/bash
# Davidson at c3 under rocprofv3 --kernel-trace: which of its passes run below the memory rate
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4z; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dvd -- python3 scripts/dvd_time.py 15 kron 2 > $OUT/dvd_traced.txt 2>&1
cp $(find $OUT/trace_dvd -name "*kernel_stats.csv" | head -1) $OUT/dvd_kernel_stats.csv
head -20 $OUT/dvd_kernel_stats.csv | cut -c1-200
