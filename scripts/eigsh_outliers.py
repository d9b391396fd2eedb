#!/usr/bin/env python3
"""Why do some sectors of the Ns=12 sweep take 85 ms in sp_eigh where their neighbours take 10?  Repeats the call per sector and prints
time + solver info.  usage: python scripts/eigsh_outliers.py [NBATH=11]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
nbath = int(sys.argv[1]) if len(sys.argv) > 1 else 11
m, _, _ = model.half_filled_c(nbath)
g = edgpu.HvSector(m, [6], [6], storage="kron"); g.sp_eigh(2, nblock=20, nitermax=300, tol=1e-10, vectors=False); g.close()
for (nu, nd) in ((2, 2), (3, 1), (3, 2), (4, 1), (8, 1), (4, 2), (5, 1), (7, 1), (6, 1), (8, 2), (9, 4), (4, 3), (5, 4), (8, 5), (6, 6)):
    G = edgpu.HvSector(m, [nu], [nd], storage="kron")
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        ev, X, info = G.sp_eigh(2, nblock=20, nitermax=512, tol=1e-10)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("(%d,%d) dim %7d  ms %s  info %s  ev %s" % (nu, nd, G.Dim, " ".join("%7.2f" % t for t in ts), {k: v for k, v in info.items() if not hasattr(v, "shape")}, ev[:2]), flush=True)
    G.close()
