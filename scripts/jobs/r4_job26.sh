#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT
python3 scripts/sector_sweep.py 11 kron 1 > $OUT/sweep_a.log 2>&1
SWEEP_NODENSE=1 python3 scripts/sector_sweep.py 11 kron 1 > $OUT/sweep_nodense.log 2>&1
OPENBLAS_NUM_THREADS=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 python3 scripts/sector_sweep.py 11 kron 1 > $OUT/sweep_1thread.log 2>&1
for f in a nodense 1thread; do echo "== $f"; grep "^#" $OUT/sweep_$f.log | head -2; awk '$7>25 && $4=="eigsh"' $OUT/sweep_$f.log | cut -c1-160; done
nproc; python3 -c "import numpy; numpy.show_config()" 2>&1 | grep -i -E "openblas|mkl|blas" | head -5
