"""Anchors of the CPU oracle that do not come from the oracle's own family of restatements:

* the interacting sector Hamiltonian written as operator algebra on the Fock space (tests/analytic.py), compared
  ELEMENT BY ELEMENT with the oracle's dense ``Hmat`` for every variant of the zoo (phonons, ed_total_ud=F, hybrid and
  replica baths, Jx/Jp);
* the reference-derived known answers of SURVEY 8c re-derived from that operator algebra (so the recorded sums, the
  operator algebra and the oracle agree pairwise);
* free-fermion spectra (all interactions off), the closed-form dimer, the U=0 impurity Green's function.
"""
import numpy as np
import pytest

import analytic
import cases
from oracle import oracle

ZOO = cases.zoo()
FOCK = [z for z in ZOO if 2 * z[1]["Ns"] <= 16]


@pytest.mark.parametrize("name,m,nu,nd", FOCK, ids=[z[0] for z in FOCK])
def test_oracle_hmat_equals_operator_algebra(name, m, nu, nd):
    S = oracle.Sector(m, nu, nd)
    H = S.hmat()
    F = analytic.fock_hamiltonian(m, nu, nd)
    assert F.shape == H.shape
    assert np.abs(H - F).max() <= 1e-14 * max(1.0, np.abs(F).max())
    v = cases.sinvec(S.Dim)
    ref = F @ v
    for hv in (S.spmatvec(v), S.directmatvec(v), S.rows(np.arange(S.Dim), v)):
        assert np.abs(hv - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("g", [g for g in cases.golden_cases() if not g.get("unpinned") and g["N"] <= 400], ids=lambda g: g["name"])
def test_reference_known_answers_from_operator_algebra(g):
    """the sums SURVEY 8c recorded from the reference's own product, recomputed without the oracle"""
    m = cases.model_from_golden(g)
    F = analytic.fock_hamiltonian(m, g["nups"], g["ndws"])
    assert F.shape[0] == g["N"]
    v = cases.sinvec(g["N"])
    hv = F @ v
    scale = np.abs(hv).sum()
    assert abs(hv.sum() - g["sum_Hv"]) <= 1e-12 * scale
    if "v_Hv" in g:
        assert abs(v @ hv - g["v_Hv"]) <= 1e-12 * scale
    if "Hv_1_3" in g:
        np.testing.assert_allclose(hv[:3], g["Hv_1_3"], rtol=1e-13)
    w = np.linalg.eigvalsh(F)
    assert abs(w[0] - g["E0"]) < 1e-11
    if "E1" in g:
        assert abs(w[1] - g["E1"]) < 1e-11


FREE = [z for z in ZOO if oracle.Sector(analytic.non_interacting(z[1]), z[2], z[3]).Dim <= 1300 and z[1].get("Nph", 0) == 0]


@pytest.mark.parametrize("name,m,nu,nd", FREE, ids=[z[0] for z in FREE])
def test_oracle_free_fermion_spectrum(name, m, nu, nd):
    """all interactions off: EVERY sector eigenvalue is a sum of one-body levels"""
    z = analytic.non_interacting(m, xmu=0.37)
    S = oracle.Sector(z, nu, nd)
    w = np.linalg.eigvalsh(S.hmat())
    ref = analytic.free_spectrum(z, nu, nd)
    assert w.size == ref.size
    np.testing.assert_allclose(w, ref, rtol=0, atol=1e-12 * max(1.0, np.abs(ref).max()) * 10)
    assert abs(w[0] - analytic.free_e0(z, nu, nd)[0]) <= 1e-12


def test_oracle_free_phonons_decouple():
    """g=0: the spectrum is the electronic free spectrum plus w0*n"""
    name, m, nu, nd = [z for z in ZOO if z[0] == "phonons_nb3_nph3"][0]
    z = analytic.non_interacting(m)
    w = np.linalg.eigvalsh(oracle.Sector(z, nu, nd).hmat())
    el = analytic.free_spectrum(dict(z, Nph=0), nu, nd)
    ref = np.sort((el[:, None] + z["w0_ph"] * np.arange(z["Nph"] + 1)[None, :]).ravel())
    np.testing.assert_allclose(w, ref, rtol=0, atol=1e-12)


@pytest.mark.parametrize("U,V", [(0.0, 0.5), (2.0, 0.5), (4.0, 1.0), (8.0, 0.3)])
def test_oracle_dimer_closed_form(U, V):
    from edgpu import model
    m = model.make_model(Norb=1, Nbath=1, Uloc=(U,), bath_e=np.zeros((1, 1, 1)), bath_v=np.full((1, 1, 1), V))
    w = np.linalg.eigvalsh(oracle.Sector(m, [1], [1]).hmat())
    np.testing.assert_allclose(w, analytic.dimer_spectrum(U, V), rtol=0, atol=1e-13)


@pytest.mark.parametrize("case,iorb,ispin", [("c_nb5_half", 1, 1), ("c_nb5_half", 1, 2), ("nb5_offhalf_2_4", 1, 2), ("norb2_normal_JxJp", 2, 1),
                                             ("norb2_normal_JxJp", 2, 2), ("replica_sz_sx_JxJp", 1, 2)])
def test_oracle_apply_op_equals_operator_algebra(case, iorb, ispin):
    """c / c^+ on a sector state (ED_GF_NORMAL.f90:190-213, 264-283): oracle loop form, its vectorised form (used at
    full size) and the Jordan-Wigner matrices"""
    name, m, nu, nd = [z for z in ZOO if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    st = np.random.RandomState(4).uniform(-1, 1, S.Dim)
    for dagger in (True, False):
        nu2, nd2 = list(nu), list(nd)
        (nu2 if ispin == 1 else nd2)[0] += 1 if dagger else -1
        St = oracle.Sector(m, nu2, nd2)
        v, n2 = oracle.apply_op(S, St, iorb, ispin, dagger, st)
        v2, n22 = oracle.apply_op_vec(S, St, iorb, ispin, dagger, st)
        np.testing.assert_array_equal(v, v2)
        assert n2 == n22
        ref = analytic.apply_c_fock(m, nu, nd, iorb, ispin, dagger, st)
        np.testing.assert_allclose(v * np.sqrt(n2), ref, rtol=0, atol=1e-15)


@pytest.mark.parametrize("case", ["orbs_norb2_nb3", "phonons_nb3_nph3", "phonons_orbs"])
def test_oracle_apply_op_vectorised_form_on_other_layouts(case):
    name, m, nu, nd = [z for z in ZOO if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    st = np.random.RandomState(5).uniform(-1, 1, S.Dim)
    total = m.get("ed_total_ud", True)
    for iorb in range(1, (1 if total else m["Norb"]) + 1):
        for ispin in (1, 2):
            for dagger in (True, False):
                nu2, nd2 = list(nu), list(nd)
                (nu2 if ispin == 1 else nd2)[0 if total else iorb - 1] += 1 if dagger else -1
                if min(nu2 + nd2) < 0 or max(nu2 + nd2) > S.Ns_Orb:
                    continue
                St = oracle.Sector(m, nu2, nd2)
                v, n2 = oracle.apply_op(S, St, iorb, ispin, dagger, st)
                v2, n22 = oracle.apply_op_vec(S, St, iorb, ispin, dagger, st)
                np.testing.assert_array_equal(v, v2)


def _cf(z, norm2, a, b, n, e0, sign):
    g = 0.0
    for k in range(n - 1, -1, -1):
        bk2 = b[k + 1] ** 2 if k + 1 < n else 0.0
        g = 1.0 / (z - sign * (a[k] - e0) - bk2 * g)
    return norm2 * g


@pytest.mark.parametrize("nb,ispin", [(5, 1), (5, 2), (7, 1)])
def test_oracle_free_green_function(nb, ispin):
    """U=0: particle + hole continued fractions of the oracle's GF leg (apply_op, sp_lanc_tridiag restatement) add up to
    [(z-h)^-1]_dd (ED_GF_NORMAL.f90:599-654)"""
    from edgpu import model
    m, nu, nd = model.half_filled_c(nb)
    z0 = analytic.non_interacting(m, xmu=0.2)
    S = oracle.Sector(z0, nu, nd)
    w, U = np.linalg.eigh(S.hmat())
    e0, gs = w[0], U[:, 0]
    assert abs(e0 - analytic.free_e0(z0, nu, nd)[0]) < 1e-12
    zs = 1j * np.array([0.1, 0.7, 2.0]) + np.array([0.0, 0.3, -1.0])
    tot = np.zeros(3, complex)
    for dagger in (True, False):
        nu2, nd2 = list(nu), list(nd)
        (nu2 if ispin == 1 else nd2)[0] += 1 if dagger else -1
        St = oracle.Sector(z0, nu2, nd2)
        v, n2 = oracle.apply_op_vec(S, St, 1, ispin, dagger, gs)
        a, b, used = oracle.sp_lanc_tridiag(St.spmatvec, v, min(St.Dim, 40), threshold=1e-9)
        assert used <= nb + 2          # the Krylov space of c^+|gs> is spanned by the empty one-body levels
        tot += np.array([_cf(zz, n2, a, b, used, e0, 1.0 if dagger else -1.0) for zz in zs])
    np.testing.assert_allclose(tot, analytic.free_gf(z0, 1, ispin, zs), rtol=0, atol=1e-9)


def test_oracle_c2_free_ground_state_by_lanczos():
    """BASELINE config 2 (Ns=12, N=853,776), all interactions off: the oracle's stored product inside its Lanczos
    restatement reaches the sum of the six lowest one-body levels per spin"""
    from edgpu import model
    m, nu, nd = model.half_filled_c(11)
    z = analytic.non_interacting(m)
    S = oracle.Sector(z, nu, nd)
    v0 = np.random.RandomState(1).uniform(-1, 1, S.Dim)
    e0, vec, nl = oracle.sp_lanc_eigh(S.spmatvec, v0, 300, threshold=1e-13)
    ref, gap = analytic.free_e0(z, nu, nd)
    assert abs(e0 - ref) <= 1e-10, (e0, ref, nl)


@pytest.mark.parametrize("nb,nu,nd", [(5, 4, 2), (5, 3, 3), (7, 5, 2)])
def test_oracle_free_observables(nb, nu, nd):
    """lanc_observables of the oracle on the non-interacting ground state against one-body occupations"""
    from edgpu import model
    m, _, _ = model.half_filled_c(nb)
    z0 = analytic.non_interacting(m, xmu=-0.15)
    S = oracle.Sector(z0, [nu], [nd])
    w, U = np.linalg.eigh(S.hmat())
    ob = oracle.lanc_observables(S, U[:, 0])
    nup, ndw = analytic.free_imp_occupations(z0, [nu], [nd])
    assert abs(ob["dens_up"][0] - nup[0]) <= 1e-12 and abs(ob["dens_dw"][0] - ndw[0]) <= 1e-12
    assert abs(ob["docc"][0] - nup[0] * ndw[0]) <= 1e-12
