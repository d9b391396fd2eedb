#!/bin/bash
# Davidson: the two-pass step (column of the projected matrix from W.t) -- parity tests, then the sweep through the experiments build
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 420 python3 -m pytest tests/test_eigsh.py -m gpu -x -q -k "davidson or dvdson" > $OUT/dvd_tests.log 2>&1; rc=$?; tail -5 $OUT/dvd_tests.log
[ $rc -eq 0 ] || exit $rc
EDGPU_DEBUG=8 timeout -k 10 200 python3 scripts/dvd_time.py 15 kron 2 2>&1 | grep -v "amdgpu.ids\|^W2026\|^E2026" > $OUT/dvd_product.txt; cat $OUT/dvd_product.txt
EDGPU_LIBRARY=$GRAFT_REPO_ROOT/dmft-lanc-ed_amd/edgpu/libedgpu_exp.so EDGPU_DEBUG=8 timeout -k 10 300 python3 scripts/dvd_time.py 15 kron 2 2>&1 | grep -v "amdgpu.ids\|^W2026\|^E2026" > $OUT/dvd_sweep.txt; cat $OUT/dvd_sweep.txt
