#!/bin/bash
# round 4: full GPU suite on the frozen sources, then smoke()
mkdir -p gpurun_out/r4f
python -m pytest tests -x -q -m gpu > gpurun_out/r4f/t16.log 2>&1; echo "exit $?" >> gpurun_out/r4f/t16.log; tail -4 gpurun_out/r4f/t16.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
