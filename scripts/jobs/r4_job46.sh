#!/bin/bash
# Davidson: tests + three timed calls of the product library (kernel variants are compared across builds by this one number)
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_eigsh.py -m gpu -x -q -k "davidson or dvdson" > $OUT/dvd_tests.log 2>&1; rc=$?; tail -3 $OUT/dvd_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do timeout -k 10 120 python3 scripts/dvd_time.py 15 kron 2 2>&1 | grep "^Ns="; done | tee $OUT/dvd_variant.txt
