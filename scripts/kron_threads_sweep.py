#!/usr/bin/env python3
"""Experiments build (EDGPU_LIBRARY=.../libedgpu_exp.so): threads per row workgroup of the single-pass product (EDGPU_KRON_THREADS) on
launch-bound sectors: us per step of a 200-step sp_lanc_tridiag.  usage: python scripts/kron_threads_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
SECT = [(9, 4, 4), (9, 5, 5), (11, 4, 2), (11, 5, 3), (11, 5, 5), (11, 6, 6), (11, 6, 4), (13, 5, 5), (13, 7, 7)]
for nbath, nup, ndw in SECT:
    m, _, _ = model.half_filled_c(nbath)
    for st in ("kron", "direct"):
        out = []
        for thr in ("0", "256", "512", "1024"):
            os.environ["EDGPU_KRON_THREADS"] = thr
            G = edgpu.HvSector(m, [nup], [ndw], storage=st)
            v0 = np.random.RandomState(1).uniform(0, 1, G.Dim)
            G.sp_lanc_tridiag(v0, 200)
            t0 = time.perf_counter(); a, b, used = G.sp_lanc_tridiag(v0, 200); dt = time.perf_counter() - t0
            out.append("%s: %.1f" % (thr, dt / max(used, 1) * 1e6))
            dim, du = G.Dim, G.info.DimUp
            G.close()
        print("Ns=%d (%d,%d) N=%d DimUp=%d %s  us/step by threads  %s" % (nbath + 1, nup, ndw, dim, du, st, "  ".join(out)), flush=True)
