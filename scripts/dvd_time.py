#!/usr/bin/env python3
"""edgpu_dvdson at full size through the experiments build: dvd_time.py NBATH STORAGE [NEIGEN]
Sweeps the size of the search space (EDGPU_DVD_MMAX), the Ritz vectors kept at a restart (EDGPU_DVD_KEEP) and the two forms of the
step (EDGPU_DVD_GCOL=1: new column of the projected matrix from the W_i.t summed inside the step kernel, new basis vector written
normalised by the orthogonalisation pass; 0: the round-4 form).  Run with EDGPU_LIBRARY=.../libedgpu_exp.so; EDGPU_DEBUG=8 prints how
many steps took the short form.  EDGPU_DVD_ROTFUSED=0: restarts by the rotation kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
nbath = int(sys.argv[1]); st = sys.argv[2]; nev = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m, nu, nd = model.half_filled_c(nbath)
G = edgpu.HvSector(m, nu, nd, storage=st)
G.sp_dvdson_eigh(nev, nitermax=600, tol=1e-10, vectors=False)      # warm-up: allocations, code objects
exp = "exp" in os.path.basename(os.environ.get("EDGPU_LIBRARY", ""))
sweep = [(0, 1, 1, 1)] if not exp else [(0, 1, 0, 0), (0, 1, 1, 0), (0, 1, 1, 1), (0, 2, 1, 1), (7, 1, 1, 1), (10, 1, 1, 1), (10, 2, 1, 1), (12, 2, 1, 1)]
for mmax, keep, gcol, rotf in sweep:
    for k, v in (("EDGPU_DVD_MMAX", mmax), ("EDGPU_DVD_KEEP", keep), ("EDGPU_DVD_GCOL", gcol), ("EDGPU_DVD_ROTFUSED", rotf)):
        os.environ[k] = str(v)
    t0 = time.perf_counter(); ed, _, infd = G.sp_dvdson_eigh(nev, nitermax=900, tol=1e-10, vectors=False, allow_unconverged=True); dt = time.perf_counter() - t0
    print("Ns=%d N=%d %s nev=%d mmax=%s keep=nev+%d gcol=%d rotfused=%d: %.3f s  %s  E=%s" % (nbath + 1, G.Dim, st, nev, mmax or "default(2*nev+4)", keep, gcol, rotf, dt, infd, np.round(ed, 10)), flush=True)
G.close()
