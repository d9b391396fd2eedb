#!/bin/bash
mkdir -p gpurun_out/r4z
python scripts/eigsh_outliers.py 11 > gpurun_out/r4z/eigsh_outliers.txt 2>&1; cat gpurun_out/r4z/eigsh_outliers.txt
