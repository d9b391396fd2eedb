"""Exact sum rules of the interacting impurity model on the CPU oracle: M0 = <{c,c^+}> = 1 and
M1 = <{[c,H],c^+}> = eps_d - mu + U(<n_-s> - 1/2) from the two Green's-function legs of an eigenstate (ED_GF_NORMAL.f90:184-311)."""
import numpy as np
import pytest

from edgpu import model
from oracle import oracle


@pytest.mark.parametrize("nu,nd,U,xmu", [(3, 3, 2.0, 0.25), (4, 2, 2.0, 0.25), (2, 3, 5.0, -0.4)])
def test_oracle_spectral_sum_rules(nu, nd, U, xmu):
    m, _, _ = model.half_filled_c(5)
    m = dict(m, xmu=xmu, Uloc=(U, 0, 0, 0, 0))
    S = oracle.Sector(m, [nu], [nd])
    w, V = np.linalg.eigh(S.hmat()); gs = V[:, 0]; e0 = w[0]
    ob = oracle.lanc_observables(S, gs)
    for ispin, nother in ((1, ob["dens_dw"][0]), (2, ob["dens_up"][0])):
        m0 = m1 = 0.0
        for dagger in (True, False):
            nu2, nd2 = [nu], [nd]
            (nu2 if ispin == 1 else nd2)[0] += 1 if dagger else -1
            St = oracle.Sector(m, nu2, nd2)
            v, n2 = oracle.apply_op_vec(S, St, 1, ispin, dagger, gs)
            m0 += n2; m1 += (1.0 if dagger else -1.0) * n2 * (v @ St.spmatvec(v) - e0)
        assert abs(m0 - 1.0) < 1e-13
        assert abs(m1 - (-xmu + U * (nother - 0.5))) < 1e-12
