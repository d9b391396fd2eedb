"""The algorithm behind edgpu_eigsh (thick-restart Lanczos, partial re-orthogonalisation) as a numpy model on the oracle's
product: lowest eigenvalues against dense diagonalisation for sectors that need many restarts, the cost in basis-vector
reads."""
import numpy as np
import pytest

import cases
import eigsh_model
from oracle import oracle

SEL = [z for z in cases.zoo() if z[0] in ("c_nb5_half", "nb5_offhalf_2_4", "norb3_nb1_JxJp", "replica_sz_sx_JxJp", "orbs_norb2_nb3", "nb6_even")]


@pytest.mark.parametrize("name,m,nu,nd", SEL, ids=[z[0] for z in SEL])
def test_partial_reorthogonalisation_model_matches_dense(name, m, nu, nd):
    S = oracle.Sector(m, nu, nd)
    w = np.linalg.eigvalsh(S.hmat())
    for nev, ncv in ((2, 20), (4, 40)):
        th, nmv, nrestart, reads, nfull, nsel = eigsh_model.eigsh_pro(S.spmatvec, S.Dim, nev, min(S.Dim, ncv))
        # single-vector Krylov: one copy of an exactly degenerate level
        for x in th:
            assert np.abs(w - x).min() <= 1e-10, (nev, x)
        assert abs(th[0] - w[0]) <= 1e-10
        full_cost = 2.0 * (ncv + nev + max(1, (ncv - nev) // 2)) / 2 + 4      # reads per product of two-pass Gram-Schmidt against the whole basis
        assert reads / nmv < full_cost, (reads / nmv, full_cost)
