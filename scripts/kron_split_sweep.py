#!/usr/bin/env python3
"""Where should the single-pass product (k_kron2) hand over to the two-pass fused iteration?  us per step of a 200-step sp_lanc_tridiag
with EDGPU_KRON_SPLIT=0 / 1 on sectors between 0.8 M and 12 M states.  usage: python scripts/kron_split_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
SECT = [(11, 6, 6), (13, 3, 3), (13, 4, 3), (13, 4, 4), (13, 5, 4), (13, 5, 5), (13, 6, 5), (13, 6, 6), (13, 7, 7)]
for nbath, nup, ndw in SECT:
    m, _, _ = model.half_filled_c(nbath)
    for st in ("kron", "direct"):
        out = []
        for sp in ("0", "1"):
            os.environ["EDGPU_KRON_SPLIT"] = sp
            G = edgpu.HvSector(m, [nup], [ndw], storage=st)
            v0 = np.random.RandomState(1).uniform(0, 1, G.Dim)
            G.sp_lanc_tridiag(v0, 100)
            t0 = time.perf_counter(); a, b, used = G.sp_lanc_tridiag(v0, 200); dt = time.perf_counter() - t0
            out.append("split=%s: %.1f" % (sp, dt / max(used, 1) * 1e6))
            dim, du, dd = G.Dim, G.info.DimUp, G.info.DimDw
            G.close()
        print("Ns=%d (%d,%d) N=%d (%d x %d, %.1f MB per vector) %s  us/step  %s" % (nbath + 1, nup, ndw, dim, du, dd, dim * 8 / 1e6, st, "  ".join(out)), flush=True)
