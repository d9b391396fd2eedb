#!/usr/bin/env python3
"""Explicit CSR: one block (row-major, EDGPU_CSR_SPLIT=0) against H_loc + H_dw (lane-major split form, EDGPU_CSR_SPLIT=1): us per step of a
200-step sp_lanc_tridiag on sectors between 0.8 M and 12 M states.  usage: python scripts/csr_split_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
SECT = [(9, 5, 5), (11, 4, 2), (11, 5, 3), (11, 6, 4), (11, 5, 5), (11, 6, 6), (13, 4, 4), (13, 5, 4), (13, 5, 5), (13, 6, 5), (13, 6, 6), (13, 7, 7)]
for nbath, nup, ndw in SECT:
    m, _, _ = model.half_filled_c(nbath)
    out = []
    for sp in ("0", "1"):
        os.environ["EDGPU_CSR_SPLIT"] = sp
        G = edgpu.HvSector(m, [nup], [ndw], storage="csr")
        v0 = np.random.RandomState(1).uniform(0, 1, G.Dim)
        G.sp_lanc_tridiag(v0, 50)
        t0 = time.perf_counter(); a, b, used = G.sp_lanc_tridiag(v0, 200); dt = time.perf_counter() - t0
        out.append("split=%s: %.1f" % (sp, dt / max(used, 1) * 1e6))
        dim, du, dd = G.Dim, G.info.DimUp, G.info.DimDw
        G.close()
    print("Ns=%d (%d,%d) N=%d (%d x %d, %.1f MB per vector) csr  us/step  %s" % (nbath + 1, nup, ndw, dim, du, dd, dim * 8 / 1e6, "  ".join(out)), flush=True)
