"""Green's-function Lanczos leg on the device (O|gs>, normalise,
sp_lanc_tridiag) against the oracle restatement and against dense resolvents."""
import numpy as np
import pytest

import cases
from oracle import oracle

pytestmark = pytest.mark.gpu


def _gf_from_lanczos(z, norm2, a, b, n, e0, sign):
    """continued fraction  norm2 / (z -+ (a1-e0) - b2^2/(z -+ (a2-e0) - ...))  (ED_GF_NORMAL.f90:599-654 uses
    the equivalent pole expansion of the same tridiagonal matrix)"""
    g = 0.0
    for k in range(n - 1, -1, -1):
        bk2 = b[k + 1] ** 2 if k + 1 < n else 0.0
        g = 1.0 / (z - sign * (a[k] - e0) - bk2 * g)
    return norm2 * g


CASES = [("c_nb5_half", 1, 1), ("c_nb5_half", 1, 2), ("norb2_normal_JxJp", 2, 1), ("orbs_norb2_nb3", 2, 2), ("phonons_nb3_nph3", 1, 1)]


@pytest.mark.parametrize("case,iorb,ispin,tstore", [c + ("kron",) for c in CASES] + [("c_nb5_half", 1, 1, "csr"), ("c_nb5_half", 1, 2, "csr"), ("phonons_nb3_nph3", 1, 1, "csr")])
def test_gf_leg_matches_oracle_and_dense(edgpu_mod, monkeypatch, case, iorb, ispin, tstore):
    """tstore: storage of the target sector the tridiagonalisation runs in; "csr" = the split explicit CSR (lane-major blocks,
    pitched columns on one rank), forced on these small sectors."""
    if tstore == "csr":
        monkeypatch.setenv("EDGPU_CSR_SPLIT", "1")
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    H = S.hmat(); w, U = np.linalg.eigh(H); e0, gs = w[0], U[:, 0]
    G0 = edgpu_mod.HvSector(m, nu, nd, storage="direct")
    total_ud = m.get("ed_total_ud", True)
    ia = 0 if total_ud else iorb - 1
    for dagger in (True, False):
        nu2, nd2 = list(nu), list(nd)
        (nu2 if ispin == 1 else nd2)[ia] += 1 if dagger else -1
        St = oracle.Sector(m, nu2, nd2)
        vref, n2ref = oracle.apply_op(S, St, iorb, ispin, dagger, gs)
        Gt = edgpu_mod.HvSector(m, nu2, nd2, storage=tstore)
        din, dout = edgpu_mod.DeviceVector(S.Dim).upload(gs), edgpu_mod.DeviceVector(St.Dim)
        n2 = edgpu_mod.apply_op(G0, Gt, iorb, ispin, dagger, din, dout)
        assert abs(n2 - n2ref) <= 1e-13
        np.testing.assert_allclose(dout.download(), vref, rtol=0, atol=1e-14)   # a signed copy: exact up to the 1/sqrt scale
        nl = min(St.Dim, 60)
        n2b, a, b, used = edgpu_mod.gf_lanczos(G0, Gt, iorb, ispin, dagger, gs, nl)
        a_ref, b_ref, used_ref = oracle.sp_lanc_tridiag(St.spmatvec, vref, nl)
        assert used == used_ref and abs(n2b - n2ref) <= 1e-13
        k = min(used, 25)
        np.testing.assert_allclose(a[:k], a_ref[:k], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(b[1:k], b_ref[1:k], rtol=1e-9, atol=1e-9)
        # G(i w_n) from the GPU coefficients vs the dense resolvent <v|(z -+ (H_t - e0))^-1|v>*norm2: 1e-8
        Ht = St.hmat(); sign = 1.0 if dagger else -1.0
        for wn in (0.3, 1.0, 3.0):
            z = 1j * wn
            dense = n2ref * (vref @ np.linalg.solve(z * np.eye(St.Dim) - sign * (Ht - e0 * np.eye(St.Dim)), vref))
            assert abs(_gf_from_lanczos(z, n2b, a, b, used, e0, sign) - dense) <= 1e-8 * max(1.0, abs(dense))
        Gt.close()
    G0.close()


CHI = [("c_nb5_half", 1, None), ("norb2_normal_JxJp", 2, 1), ("orbs_norb2_nb3", 1, 2), ("phonons_nb3_nph3", 1, None)]


@pytest.mark.parametrize("case,iorb,jorb", CHI)
def test_susceptibility_leg(edgpu_mod, case, iorb, jorb):
    """Sz_a, n_a and the mixed (a,b) start vectors of the chi builders (ED_GF_CHISPIN.f90:169-171,
    393-394; ED_GF_CHIDENS.f90:166-168, 389-390): norm2 and Lanczos coefficients against a direct
    numpy evaluation on the oracle's basis + the oracle's sp_lanc_tridiag."""
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    w, U = np.linalg.eigh(S.hmat()); gs = U[:, 0]
    O = oracle.lanc_observables          # occupations via the same basis walk: weights of single basis states
    Norb = m["Norb"]
    # impurity occupations of every basis state, from the oracle's restatement
    nup = np.zeros((S.Dim, Norb)); ndw = np.zeros((S.Dim, Norb))
    for i in range(S.Dim):
        e = np.zeros(S.Dim); e[i] = 1.0
        o = O(S, e)
        nup[i], ndw[i] = o["dens_up"], o["dens_dw"]
    G = edgpu_mod.HvSector(m, nu, nd)
    combos = [("spin", {iorb: (0.5, -0.5)}), ("dens", {iorb: (1.0, 1.0)})]
    if jorb:
        combos.append(("spin_mix", {iorb: (0.5, -0.5), jorb: (0.5, -0.5)}))
        combos.append(("dens_mix", {iorb: (1.0, 1.0), jorb: (1.0, 1.0)}))
    for label, coef in combos:
        cu = np.zeros(Norb); cd = np.zeros(Norb)
        for a, (x, y) in coef.items():
            cu[a - 1] += x; cd[a - 1] += y
        v = (nup @ cu + ndw @ cd) * gs
        n2ref = float(v @ v)
        nl = min(S.Dim, 40)
        n2, a, b, used = G.chi_lanczos(cu, cd, gs, nl)
        assert abs(n2 - n2ref) <= 1e-13, label
        if n2ref < 1e-20:
            assert used == 0
            continue
        a_ref, b_ref, used_ref = oracle.sp_lanc_tridiag(S.spmatvec, v / np.sqrt(n2ref), nl)
        assert used == used_ref, label
        k = min(used, 20)
        np.testing.assert_allclose(a[:k], a_ref[:k], rtol=1e-9, atol=1e-9, err_msg=label)
        np.testing.assert_allclose(b[1:k], b_ref[1:k], rtol=1e-9, atol=1e-9, err_msg=label)
    G.close()


@pytest.mark.parametrize("case", ["phonons_nb3_nph3", "phonons_norb2_JxJp", "phonons_orbs"])
def test_phonon_propagator_leg(edgpu_mod, case):
    """(b+b^+)|gs> restated from ED_GF_PHONON.f90:76-90 (scatter over source states) vs the device gather."""
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    w, U = np.linalg.eigh(S.hmat()); gs = U[:, 0]
    nel = S.DimUp * S.DimDw
    v = np.zeros(S.Dim)
    for i in range(1, S.Dim + 1):
        iph = (i - 1) // nel + 1; i_el = (i - 1) % nel + 1
        if iph > 1:
            v[i_el + (iph - 2) * nel - 1] += np.sqrt(float(iph - 1)) * gs[i - 1]
        if iph < S.DimPh:
            v[i_el + iph * nel - 1] += np.sqrt(float(iph)) * gs[i - 1]
    n2ref = float(v @ v)
    G = edgpu_mod.HvSector(m, nu, nd)
    nl = min(S.Dim, 40)
    n2, a, b, used = G.phonon_lanczos(gs, nl)
    assert abs(n2 - n2ref) <= 1e-12 * max(1.0, n2ref)
    a_ref, b_ref, used_ref = oracle.sp_lanc_tridiag(S.spmatvec, v / np.sqrt(n2ref), nl)
    assert used == used_ref
    k = min(used, 20)
    np.testing.assert_allclose(a[:k], a_ref[:k], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(b[1:k], b_ref[1:k], rtol=1e-9, atol=1e-9)
    G.close()


def test_legs_take_the_ground_state_straight_from_hbm(edgpu_mod):
    """sp_lanc_eigh_device leaves the eigenvector in HBM; the GF / chi legs accept that device pointer in place of a
    host vector (no PCIe round trip of the N-vector between diagonalisation and Green's functions)."""
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == "c_nb7_half"][0]
    G0 = edgpu_mod.HvSector(m, nu, nd)
    dv = edgpu_mod.DeviceVector(G0.Dim).upload(np.zeros(G0.Dim))
    e0, nl = G0.sp_lanc_eigh_device(dv, nitermax=300, threshold=1e-14)
    gs = dv.download()
    Gt = edgpu_mod.HvSector(m, [nu[0] + 1], nd)
    r_dev = edgpu_mod.gf_lanczos(G0, Gt, 1, 1, True, dv, 40)
    r_host = edgpu_mod.gf_lanczos(G0, Gt, 1, 1, True, gs, 40)
    assert r_dev[0] == r_host[0] and r_dev[3] == r_host[3]
    np.testing.assert_array_equal(r_dev[1], r_host[1]); np.testing.assert_array_equal(r_dev[2], r_host[2])
    c_dev = G0.chi_lanczos([0.5], [-0.5], dv, 30); c_host = G0.chi_lanczos([0.5], [-0.5], gs, 30)
    assert c_dev[0] == c_host[0]
    np.testing.assert_array_equal(c_dev[1], c_host[1])
    np.testing.assert_array_equal(dv.download(), gs)          # the resident state is left untouched
    Gt.close(); G0.close()
