#!/usr/bin/env python3
"""BASELINE config 5 family (2-orbital BHZ-like impurity, replica bath, H_basis={sigma_z}): product and Lanczos
iteration times on one GPU, with and without the non-local Jx/Jp terms.  bench_c5.py NBATH [kron|direct]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import edgpu
from edgpu import model
import cases
nbath = int(sys.argv[1]); st = sys.argv[2] if len(sys.argv) > 2 else "kron"
Ns = 2 * (nbath + 1)
for J in (0.0, 0.25):
    m = model.make_model(Norb=2, Nbath=nbath, bath_type="replica", H_basis=[cases.sigma_z()], lambda_imp=[1.0],
                         Uloc=(2.0, 2.0), Jh=0.25, Ust=1.5, Jx=J, Jp=J)
    t0 = time.time(); G = edgpu.HvSector(m, [Ns // 2], [Ns // 2], storage=st); tb = time.time() - t0
    N = G.Dim
    v = edgpu.DeviceVector(N); h = edgpu.DeviceVector(N)
    v.upload(np.random.RandomState(0).uniform(-0.5, 0.5, N))
    ms = G.hxv_time(v, h, 2, 10)
    v.free(); h.free()
    G.bench_begin(14); G.bench_steps(2)
    import torch; torch.cuda.synchronize(); t0 = time.perf_counter(); G.bench_steps(10); torch.cuda.synchronize(); it = (time.perf_counter() - t0) / 10
    print("c5 Nbath=%d Ns=%d N=%d %s Jx=Jp=%.2f: build %.1fs  HxV %.3f ms (24N: %.0f GB/s)  Lanczos iteration %.3f ms" % (nbath, Ns, N, st, J, tb, ms, 24 * N / ms / 1e6, it * 1e3), flush=True)
    G.close()
