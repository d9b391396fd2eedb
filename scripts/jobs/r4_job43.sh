#!/bin/bash
# full GPU suite on the sources of this session, then the default bench line
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; rc=$?; tail -6 $OUT/suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err; rc=$?; cut -c1-600 $OUT/bench_plain.json
exit $rc
