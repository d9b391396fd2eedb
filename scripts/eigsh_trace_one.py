#!/usr/bin/env python3
"""One small sector through edgpu_eigsh (nev=2, ncv=20) a few times: the run to put under rocprofv3 --kernel-trace --hip-trace.
usage: python scripts/eigsh_trace_one.py NBATH NUP NDW [STORAGE=kron] [REPS=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
nbath, nu, nd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
st = sys.argv[4] if len(sys.argv) > 4 else "kron"; reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
m, _, _ = model.half_filled_c(nbath)
G = edgpu.HvSector(m, [nu], [nd], storage=st)
for r in range(reps):
    t0 = time.perf_counter(); ev, X, info = G.sp_eigh(2, nblock=20, nitermax=512, tol=1e-10); dt = (time.perf_counter() - t0) * 1e3
    print("rep %d: %.2f ms  %s" % (r, dt, info), flush=True)
G.close()
