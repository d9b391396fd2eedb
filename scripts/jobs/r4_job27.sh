#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/sector_sweep.py 11 kron 1 > $OUT/sweep_pool.log 2>&1; grep "^#" $OUT/sweep_pool.log | head -2
for sec in "6 4" "3 2"; do
  tag=$(echo $sec | tr ' ' '_')
  rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $OUT/eigtr_$tag -- python3 scripts/eigsh_trace_one.py 11 $sec kron 3 > $OUT/eigtr_$tag.log 2>&1
  grep "^rep" $OUT/eigtr_$tag.log
  python3 scripts/eigsh_trace_report.py $OUT/eigtr_$tag
  rm -f $OUT/eigtr_$tag/*/*hip_api_trace.csv $OUT/eigtr_$tag/*/*kernel_trace.csv
done
