#!/usr/bin/env python3
"""Wall time of the whole ground-state search (edgpu_lanczos_gs_device, vector resident) with the Krylov vectors
stored in HBM (default) and with the second Lanczos pass (EDGPU_LZ_STORE=0).  gs_time.py NBATH STORAGE"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
nbath = int(sys.argv[1]); st = sys.argv[2]
m, nu, nd = model.half_filled_c(nbath)
for mode in ("1", "0"):
    os.environ["EDGPU_LZ_STORE"] = mode
    G = edgpu.HvSector(m, nu, nd, storage=st)
    dv = edgpu.DeviceVector(G.Dim)
    dts = []
    for rep in range(2):
        dv.upload(np.zeros(G.Dim))
        t0 = time.perf_counter(); e0, nl = G.sp_lanc_eigh_device(dv, nitermax=300, threshold=1e-12); dts.append(time.perf_counter() - t0)
    print("Ns=%d N=%d %s EDGPU_LZ_STORE=%s: E0=%.12f nlanc=%d  first call %.3f s, second %.3f s" % (nbath + 1, G.Dim, st, mode, e0, nl, dts[0], dts[1]), flush=True)
    dv.free(); G.close()
