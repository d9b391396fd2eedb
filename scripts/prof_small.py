#!/usr/bin/env python3
"""One small sector, 200 Lanczos steps (for rocprofv3 --kernel-trace --stats): prof_small.py NBATH STORAGE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
m, nu, nd = model.half_filled_c(int(sys.argv[1]))
G = edgpu.HvSector(m, nu, nd, storage=sys.argv[2])
v0 = np.random.RandomState(1).uniform(0, 1, G.Dim)
for _ in range(3):
    G.sp_lanc_tridiag(v0, 200)
G.close()
