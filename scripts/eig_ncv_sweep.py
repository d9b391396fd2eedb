#!/usr/bin/env python3
"""edgpu_eigsh at full size as a function of the basis size: eig_ncv_sweep.py NBATH STORAGE NEV NCV [NCV ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
nbath = int(sys.argv[1]); st = sys.argv[2]; nev = int(sys.argv[3])
m, nu, nd = model.half_filled_c(nbath)
G = edgpu.HvSector(m, nu, nd, storage=st)
for ncv in [int(x) for x in sys.argv[4:]]:
    for rep in range(2):
        t0 = time.perf_counter(); ev, _, info = G.sp_eigh(nev, nblock=ncv, nitermax=300, tol=1e-10, vectors=False); t1 = time.perf_counter() - t0
    print("Ns=%d N=%d %s nev=%d ncv=%d: %.3f s %s E=%s" % (nbath + 1, G.Dim, st, nev, ncv, t1, info, np.round(ev, 10)), flush=True)
G.close()
