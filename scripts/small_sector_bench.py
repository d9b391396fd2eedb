#!/usr/bin/env python3
"""Launch-bound regime: wall time of sp_lanc_eigh and a 200-step sp_lanc_tridiag on small sectors.  Round 4: sectors whose vector
fits LDS run the whole loop in one workgroup (k_lanczos_wg, default) -- compared with the launch-per-iteration loop (EDGPU_LZ_WG=0);
larger ones take the launch-per-iteration loop either way."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model

SECT = [(7, 4, 4), (9, 3, 3), (9, 4, 4), (9, 5, 5), (11, 4, 2), (11, 5, 2), (11, 6, 6), (13, 7, 7)]      # (Nbath, Nup, Ndw)
for nbath, nup, ndw in SECT:
    m, _, _ = model.half_filled_c(nbath)
    nu, nd = [nup], [ndw]
    for st in ("kron", "direct"):
        row = []
        for g in ("0", "1"):
            os.environ["EDGPU_LZ_WG"] = g
            G = edgpu.HvSector(m, nu, nd, storage=st)
            v0 = np.random.RandomState(1).uniform(0, 1, G.Dim)
            G.sp_lanc_eigh(v0, nitermax=300, threshold=1e-12)          # warm-up (graph capture, allocations)
            t0 = time.perf_counter(); e0, x, nl = G.sp_lanc_eigh(v0, nitermax=300, threshold=1e-12); t1 = time.perf_counter() - t0
            G.sp_lanc_tridiag(v0, 200)
            t0 = time.perf_counter(); a, b, used = G.sp_lanc_tridiag(v0, 200); t2 = time.perf_counter() - t0
            row.append((e0, nl, t1, used, t2, a[:5].copy()))
            G.close()
        (e_a, nl_a, t1a, ua, t2a, aa), (e_b, nl_b, t1b, ub, t2b, ab) = row
        assert abs(nl_a - nl_b) <= 1 and abs(e_a - e_b) < 1e-10 and np.allclose(aa, ab, rtol=1e-9, atol=1e-9), (nl_a, nl_b, e_a, e_b)
        print("Ns=%d (%d,%d) N=%d %s: lanc_eigh nlanc=%d  launch per iteration %.2f ms  one workgroup %.2f ms | tridiag200: %.2f ms (%.1f us/step) -> %.2f ms (%.1f us/step)%s"
              % (nbath + 1, nup, ndw, G.Dim, st, nl_a, t1a * 1e3, t1b * 1e3, t2a * 1e3, t2a / max(ua, 1) * 1e6, t2b * 1e3, t2b / max(ub, 1) * 1e6,
                 "" if G.Dim * 8 <= 150 * 1024 else "   [vector beyond LDS: same loop in both columns]"), flush=True)
os.environ.pop("EDGPU_LZ_WG", None)

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
