#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/sector_sweep.py 11 kron 1 > $OUT/sweep_coef.log 2>&1; grep "^#" $OUT/sweep_coef.log | head -2
python3 scripts/eigsh_outliers.py 11 > $OUT/eigsh_coef.txt 2>&1; cut -c1-60 $OUT/eigsh_coef.txt
python -m pytest tests -x -q -m gpu > $OUT/t28.log 2>&1; echo "exit $?" >> $OUT/t28.log; tail -4 $OUT/t28.log
