#!/usr/bin/env python3
"""Callers either side of the product at BASELINE config 3 size (Ns=16): one Green's-function leg of 200 Lanczos steps in the
(9,8) sector (ED_GF_NORMAL.f90:184-239: c^+_up|gs>, normalise, sp_lanc_tridiag), one in (8,7) (c_dw), and lanc_local_energy /
the impurity histogram on the ground state (ED_OBSERVABLES.f90:150-215, 372-616).  gf_leg_time.py [STORAGE of the target sector]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
st = sys.argv[1] if len(sys.argv) > 1 else "kron"
m, nu, nd = model.half_filled_c(15)
G0 = edgpu.HvSector(m, nu, nd, storage="kron")
gs = edgpu.DeviceVector(G0.vecDim).zero()
t0 = time.perf_counter(); e0, nl = G0.sp_lanc_eigh_device(gs, nitermax=300, threshold=1e-12); tg = time.perf_counter() - t0
print("ground state (8,8) N=%d: E0=%.12f, %d iterations, %.3f s" % (G0.Dim, e0, nl, tg), flush=True)
edgpu.cache_release()
for ispin, dagger, nu2, nd2 in ((1, True, [9], [8]), (2, False, [8], [7])):
    Gt = edgpu.HvSector(m, nu2, nd2, storage=st)
    for rep in range(2):
        t0 = time.perf_counter(); n2, a, b, used = edgpu.gf_lanczos(G0, Gt, 1, ispin, dagger, gs, 200); tl = time.perf_counter() - t0
    print("GF leg %s_%s -> sector (%d,%d) N=%d storage=%s: norm2=%.12f, %d steps, %.3f s (%.2f ms per step incl. O|gs>)"
          % ("c^+" if dagger else "c", "up" if ispin == 1 else "dw", nu2[0], nd2[0], Gt.Dim, st, n2, used, tl, tl / max(used, 1) * 1e3), flush=True)
    Gt.close()
for rep in range(2):
    t0 = time.perf_counter(); le = G0.local_energy(gs); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); ob = G0.observables(gs); t2 = time.perf_counter() - t0
print("lanc_local_energy on the resident ground state: %.2f ms %s" % (t1 * 1e3, {k: round(v, 10) for k, v in le.items()}))
print("impurity histogram + lanc_observables sums: %.2f ms  dens=%.10f docc=%.10f" % (t2 * 1e3, ob["dens"][0], ob["docc"][0]))
G0.close()
