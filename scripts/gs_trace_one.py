#!/usr/bin/env python3
"""One small sector through sp_lanc_eigh a few times with wall-clock splits: python scripts/gs_trace_one.py NBATH NUP NDW"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
nbath, nu, nd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m, _, _ = model.half_filled_c(nbath)
G = edgpu.HvSector(m, [nu], [nd], storage="kron")
v0 = np.random.RandomState(1).uniform(0, 1, G.Dim)
for r in range(4):
    t0 = time.perf_counter(); e0, x, nl = G.sp_lanc_eigh(v0, nitermax=300, threshold=1e-12); dt = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); a, b, used = G.sp_lanc_tridiag(v0, nl); dt2 = (time.perf_counter() - t0) * 1e3
    print("rep %d: sp_lanc_eigh %.2f ms (nlanc %d)   sp_lanc_tridiag of the same length %.2f ms" % (r, dt, nl, dt2), flush=True)
G.close()
