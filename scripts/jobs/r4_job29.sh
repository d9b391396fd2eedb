#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT
python3 scripts/eigsh_outliers.py 11 > $OUT/eigsh_sr.txt 2>&1; cut -c1-60 $OUT/eigsh_sr.txt
python3 scripts/sector_sweep.py 11 kron 1 > $OUT/sweep_sr.log 2>&1; grep "^#" $OUT/sweep_sr.log | head -2
python -m pytest tests/test_eigsh.py tests/test_small_sector_lanczos.py tests/test_observables.py -x -q -m gpu > $OUT/t29.log 2>&1; echo "exit $?" >> $OUT/t29.log; tail -4 $OUT/t29.log
