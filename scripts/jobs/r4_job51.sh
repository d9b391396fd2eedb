#!/bin/bash
# bench line with the Davidson timing among the extras: contract tests, then the default run
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_bench_contract.py -m gpu -x -q > $OUT/bench_contract.log 2>&1; rc=$?; tail -3 $OUT/bench_contract.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py > $OUT/bench_final2.json 2> $OUT/bench_final2.err; rc=$?; python3 -c "
import json; d=json.load(open('$OUT/bench_final2.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d.get('eigsh_c3_s'), d.get('dvdson_c3_s'), d.get('dvdson_path'))"
exit $rc
