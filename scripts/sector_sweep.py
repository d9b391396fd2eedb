#!/usr/bin/env python3
"""The sector sweep the host drives in one DMFT iteration (ED_DIAG.f90:83-276, then ED_GF_NORMAL.f90:224-239), timed per sector:
   create -> sp_eigh(Neigen=2, Nblock=20) [dense form for dim <= lanc_dim_threshold = 1024, ED_DIAG.f90:105-107]
          -> one Green's-function leg (c^+_up on the sector's lowest state, 200 Lanczos steps in the target sector) -> destroy
for every (Nup, Ndw) sector of a 1-orbital model.  usage: python scripts/sector_sweep.py [NBATH=11] [STORAGE=kron] [twin=1]
twin=1 (the reference's ed_twin=T default for Nspin=1): only Nup >= Ndw sectors are diagonalised, the others follow by symmetry.
Prints a per-sector table and the totals; the reference's own recorded time for ONE product of the largest Ns=12 sector is 0.272 s
(BASELINE.md)."""
import os, sys, time
# the dense sectors (dim <= 1024) go through numpy's LAPACK here, as the host code does them on the CPU; OpenBLAS starts one thread per
# VISIBLE core (256 on a GPU box whose share is 16) and their spinning starved the solver's host thread: 85-90 ms for an edgpu_eigsh call
# that takes 9 ms, and 4.1 s for a sweep that takes 1.8 s (profiles/r04/sector_sweep_hip_api.txt)
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model

nbath = int(sys.argv[1]) if len(sys.argv) > 1 else 11
storage = sys.argv[2] if len(sys.argv) > 2 else "kron"
twin = (sys.argv[3] if len(sys.argv) > 3 else "1") != "0"
m, _, _ = model.half_filled_c(nbath)
Ns = m["Ns"]
THRESH = 1024
rows = []
# warm-up: runtime initialisation, kernel loading, the work-vector cache
g = edgpu.HvSector(m, [Ns // 2], [Ns // 2], storage=storage); g.sp_eigh(2, nblock=20, nitermax=300, tol=1e-10, vectors=False); g.close()
t_all = time.perf_counter()
tot = dict(create=0.0, solve=0.0, gf=0.0, destroy=0.0)
nlanc_sectors = 0
for nup in range(Ns + 1):
    for ndw in range(Ns + 1):
        if twin and nup < ndw:
            continue
        t0 = time.perf_counter()
        G = edgpu.HvSector(m, [nup], [ndw], storage=storage)
        t1 = time.perf_counter(); c1 = time.process_time()
        dim = int(G.Dim)
        if dim <= THRESH and os.environ.get("SWEEP_NODENSE"):
            e0, x, how, prods = 0.0, np.ones(dim) / np.sqrt(dim), "skip", 0
        elif dim <= THRESH:
            w, V = np.linalg.eigh(G.hmat())
            e0, x, how, prods = float(w[0]), np.ascontiguousarray(V[:, 0]), "dense", 0
        else:
            ev, X, info = G.sp_eigh(min(2, dim), nblock=min(dim, 20), nitermax=512, tol=1e-10)
            e0, x, how, prods = float(ev[0]), np.ascontiguousarray(X[:, 0]), "eigsh", info["matvecs"]
            nlanc_sectors += 1
        t2 = time.perf_counter(); c2 = time.process_time()
        tg = 0.0; nused = 0
        if nup < Ns:                       # c^+_up leg: target sector (nup+1, ndw)
            T = edgpu.HvSector(m, [nup + 1], [ndw], storage=storage)
            tg0 = time.perf_counter()
            n2, a, b, nused = edgpu.gf_lanczos(G, T, 1, 1, True, x, min(200, int(T.Dim)))
            tg = time.perf_counter() - tg0
            T.close()
        t3 = time.perf_counter()
        G.close()
        t4 = time.perf_counter()
        tot["create"] += t1 - t0; tot["solve"] += t2 - t1; tot["gf"] += tg; tot["destroy"] += (t4 - t3) + ((t3 - t2) - tg)
        rows.append((nup, ndw, dim, how, prods, (t1 - t0) * 1e3, (t2 - t1) * 1e3, nused, tg * 1e3, e0, (c2 - c1) * 1e3))
t_all = time.perf_counter() - t_all
print("# Ns=%d storage=%s twin=%s: %d sectors (%d through the Lanczos solver), wall %.3f s" % (Ns, storage, twin, len(rows), nlanc_sectors, t_all))
print("# totals: create %.3f s  solve %.3f s  GF legs %.3f s  target-sector create + destroy %.3f s" % (tot["create"], tot["solve"], tot["gf"], tot["destroy"]))
print("# nup ndw      dim  solver products  create_ms  solve_ms  gf_steps  gf_ms  (us per GF step)   E0   solve_cpu_ms (all threads)")
for r in rows:
    print("%4d %3d %8d  %6s %8d  %9.2f %9.2f  %8d %7.2f  %8.1f   %.10f  %8.2f" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], (r[8] * 1e3 / r[7]) if r[7] else 0.0, r[9], r[10]))
