#!/usr/bin/env python3
"""Launch-bound regime: wall time of sp_lanc_eigh and a 200-step sp_lanc_tridiag on small sectors, with the
chunked host checks of the stopping rule on (default, 6 iterations per round trip) and off (EDGPU_LZ_CHUNK=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model

for nbath in (7, 9, 11, 13):
    m, nu, nd = model.half_filled_c(nbath)
    for st in ("kron", "direct"):
        row = []
        for g in ("1", "6"):
            os.environ["EDGPU_LZ_CHUNK"] = g
            G = edgpu.HvSector(m, nu, nd, storage=st)
            v0 = np.random.RandomState(1).uniform(0, 1, G.Dim)
            G.sp_lanc_eigh(v0, nitermax=300, threshold=1e-12)          # warm-up (graph capture, allocations)
            t0 = time.perf_counter(); e0, x, nl = G.sp_lanc_eigh(v0, nitermax=300, threshold=1e-12); t1 = time.perf_counter() - t0
            G.sp_lanc_tridiag(v0, 200)
            t0 = time.perf_counter(); a, b, used = G.sp_lanc_tridiag(v0, 200); t2 = time.perf_counter() - t0
            row.append((e0, nl, t1, used, t2, a[:5].copy()))
            G.close()
        (e_a, nl_a, t1a, ua, t2a, aa), (e_b, nl_b, t1b, ub, t2b, ab) = row
        assert nl_a == nl_b and abs(e_a - e_b) < 1e-12 and np.array_equal(aa, ab), (nl_a, nl_b, e_a, e_b)
        print("Ns=%d N=%d %s: lanc_eigh nlanc=%d  check-every-step %.2f ms (%.1f us/it)  chunked %.2f ms (%.1f us/it) | tridiag200 %.2f / %.2f ms"
              % (nbath + 1, G.Dim, st, nl_a, t1a * 1e3, t1a / (2 * nl_a) * 1e6, t1b * 1e3, t1b / (2 * nl_b) * 1e6, t2a * 1e3, t2b * 1e3), flush=True)

# the reference's default solver path on the same sectors: sp_eigh(Neigen=2, Nblock=20) -> edgpu_eigsh, both forms
for nbath in (7, 9, 11, 13):
    m, nu, nd = model.half_filled_c(nbath)
    out = []
    for pro in ("1", "0"):
        os.environ["EDGPU_EIGSH_PRO"] = pro
        G = edgpu.HvSector(m, nu, nd, storage="kron")
        G.sp_eigh(2, nblock=20, nitermax=300, tol=1e-10, vectors=False)
        t0 = time.perf_counter(); ev, _, info = G.sp_eigh(2, nblock=20, nitermax=300, tol=1e-10, vectors=False); t1 = time.perf_counter() - t0
        out.append((t1, info["matvecs"], ev))
        G.close()
    assert np.abs(out[0][2] - out[1][2]).max() < 1e-9
    print("Ns=%d N=%d kron: sp_eigh(2 pairs, 20 vectors) partial re-orthogonalisation %.2f ms (%d products) | whole-basis form %.2f ms (%d products)"
          % (nbath + 1, G.Dim, out[0][0] * 1e3, out[0][1], out[1][0] * 1e3, out[1][1]), flush=True)
