#!/bin/bash
mkdir -p gpurun_out/r4z
python -m pytest tests/test_small_sector_lanczos.py tests/test_eigsh.py -x -q -m gpu > gpurun_out/r4z/t32.log 2>&1; echo "exit $?" >> gpurun_out/r4z/t32.log; tail -25 gpurun_out/r4z/t32.log
