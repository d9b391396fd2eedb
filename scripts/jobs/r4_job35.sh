#!/bin/bash
mkdir -p gpurun_out/r4z
python3 scripts/csr_split_sweep.py > gpurun_out/r4z/csr_split.txt 2>&1; cat gpurun_out/r4z/csr_split.txt
