#!/bin/bash
# bench part of the profile round on the final sources (traffic stamp), the solvers at c3, Davidson under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh bench > gpurun_out/prof_round_bench.log 2>&1; tail -12 gpurun_out/prof_round_bench.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
python3 scripts/eig_time.py 15 kron 2 > $OUT/eig_time_c3.txt 2>&1; grep "^Ns=" $OUT/eig_time_c3.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dvd -- python3 scripts/dvd_time.py 15 kron 2 > $OUT/dvd_traced.txt 2>&1
cp $(find $OUT/trace_dvd -name "*kernel_stats.csv" | head -1) $OUT/dvd_kernel_stats.csv; head -12 $OUT/dvd_kernel_stats.csv | cut -c1-160
