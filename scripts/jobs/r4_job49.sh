#!/bin/bash
# final sources of the round: full GPU suite, then where an edgpu_eigsh call spends its time, the Ns=12 / Ns=8 sector sweeps, the small-sector table, the default bench line
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/suite_final.log 2>&1; rc=$?; tail -6 $OUT/suite_final.log
[ $rc -eq 0 ] || exit $rc
for sec in "11 4 4" "11 6 6" "11 3 2" "9 5 5"; do echo "== sector $sec"; EDGPU_DEBUG=8 timeout -k 10 100 python3 scripts/eigsh_trace_one.py $sec kron 3 2>&1 | grep -v "amdgpu.ids" | grep "loop\|rep" | tail -3; done > $OUT/eigsh_where3.txt; cut -c1-330 $OUT/eigsh_where3.txt
timeout -k 10 200 python3 scripts/sector_sweep.py 11 kron 1 > $OUT/sector_sweep_ns12.txt 2>&1; head -3 $OUT/sector_sweep_ns12.txt
timeout -k 10 100 python3 scripts/sector_sweep.py 7 kron 1 > $OUT/sector_sweep_ns8.txt 2>&1; head -3 $OUT/sector_sweep_ns8.txt
timeout -k 10 200 python3 scripts/small_sector_bench.py > $OUT/small_sectors.txt 2>&1; grep -v "amdgpu.ids" $OUT/small_sectors.txt | cut -c1-200 | tail -20
timeout -k 10 300 python3 bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err; rc=$?; cut -c1-400 $OUT/bench_final.json
exit $rc
