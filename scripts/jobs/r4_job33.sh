#!/bin/bash
mkdir -p gpurun_out/r4z
EDGPU_LIBRARY=$GRAFT_REPO_ROOT/dmft-lanc-ed_amd/edgpu/libedgpu_exp.so python3 scripts/kron_threads_sweep.py > gpurun_out/r4z/kron_threads.txt 2>&1; cat gpurun_out/r4z/kron_threads.txt
