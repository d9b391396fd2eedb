#!/bin/bash
# edgpu_eigsh with the small results through pinned memory: parity tests, then where the wall time goes (product build, then EDGPU_EIGSH_PINNED=0 through the experiments build)
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_eigsh.py tests/test_small_sector_lanczos.py -m gpu -x -q > $OUT/eigsh_tests.log 2>&1; rc=$?; tail -3 $OUT/eigsh_tests.log
[ $rc -eq 0 ] || exit $rc
L=$GRAFT_REPO_ROOT/dmft-lanc-ed_amd/edgpu/libedgpu_exp.so
for sec in "11 4 4" "11 6 6" "11 3 2" "9 5 5"; do
  for pinned in 1 0; do echo "== sector $sec EDGPU_EIGSH_PINNED=$pinned"; EDGPU_LIBRARY=$L EDGPU_EIGSH_PINNED=$pinned EDGPU_DEBUG=8 timeout -k 10 100 python3 scripts/eigsh_trace_one.py $sec kron 3 2>&1 | grep -v "amdgpu.ids" | grep "loop\|rep" | tail -3; done
done > $OUT/eigsh_where2.txt; cat $OUT/eigsh_where2.txt | cut -c1-330
