#!/usr/bin/env python3
"""Multi-eigenpair solvers at full size: eig_time.py NBATH STORAGE NEIGEN [NCV]   (EDGPU_EIGSH_PRO=0: full re-orthogonalisation)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
nbath = int(sys.argv[1]); st = sys.argv[2]; nev = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ncv = int(sys.argv[4]) if len(sys.argv) > 4 else max(20, 10 * nev)      # Nblock = 10*max(Neigen,lanc_nstates_sector), ED_DIAG.f90:95-98
m, nu, nd = model.half_filled_c(nbath)
G = edgpu.HvSector(m, nu, nd, storage=st)
for rep in range(2):
    t0 = time.perf_counter(); ev, _, info = G.sp_eigh(nev, nblock=ncv, nitermax=300, tol=1e-10, vectors=False); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); ed, _, infd = G.sp_dvdson_eigh(nev, nitermax=600, tol=1e-10, vectors=False); t2 = time.perf_counter() - t0
print("Ns=%d N=%d %s nev=%d ncv=%d: thick-restart Lanczos %.3f s (%s) E=%s | Davidson %.3f s (%s) E=%s" % (nbath + 1, G.Dim, st, nev, ncv, t1, info, np.round(ev, 10), t2, infd, np.round(ed, 10)), flush=True)
G.close()
