#!/bin/bash
mkdir -p gpurun_out/r4z
python3 scripts/kron_split_sweep.py > gpurun_out/r4z/kron_split.txt 2>&1; cat gpurun_out/r4z/kron_split.txt
