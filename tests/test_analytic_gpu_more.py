"""More oracle-free anchors of the HIP path (tests/analytic.py): the spin / charge susceptibility legs of the non-interacting
model at every BASELINE size, and free-fermion ground states of ed_total_ud=F and hybrid-bath sectors beyond dense reach."""
import numpy as np
import pytest

import analytic
import cases

pytestmark = pytest.mark.gpu


def _cf(z, norm2, a, b, n, e0):
    g = 0.0
    for k in range(n - 1, -1, -1):
        bk2 = b[k + 1] ** 2 if k + 1 < n else 0.0
        g = 1.0 / (z - (a[k] - e0) - bk2 * g)
    return norm2 * g


def _free_chi(m, nu, nd, cu, cd, zs):
    """<gs| O (z - (H - E0))^-1 O |gs> for O = cu n_d,up + cd n_d,dw on the Slater-determinant ground state:
    O|gs> = <O>|gs> + sum_s c_s sum_{n filled, m empty} U_dm U_dn |ph_s(m,n)>, particle-hole states being eigenstates at e_m - e_n"""
    tot = np.zeros(len(zs), complex); mean = 0.0
    for spin, n, c in ((0, nu, cu), (1, nd, cd)):
        w, U = np.linalg.eigh(analytic.one_body(m, spin))
        mean += c * (U[0, :n] ** 2).sum()
        for i in range(n):
            for j in range(n, w.size):
                tot += c * c * (U[0, i] * U[0, j]) ** 2 / (zs - (w[j] - w[i]))
    return tot + mean * mean / zs


@pytest.mark.parametrize("nb,nu,nd", [(5, 3, 3), (5, 4, 2), (11, 6, 6), (15, 8, 8), (15, 9, 7)])
def test_free_susceptibility_legs(edgpu_mod, nb, nu, nd):
    """edgpu_chi_lanczos (Sz and n start vectors of ED_GF_CHISPIN.f90:169-171 / ED_GF_CHIDENS.f90:166-168) on the U=0 ground
    state against the particle-hole sum, to 1e-8, up to the c3 sectors (N = 165,636,900 and 130,873,600)."""
    from edgpu import model
    m, _, _ = model.half_filled_c(nb)
    z0 = analytic.non_interacting(m, xmu=0.1)
    G = edgpu_mod.HvSector(z0, [nu], [nd], storage="kron")
    ev, X, info = G.sp_eigh(1, nblock=20, nitermax=300, tol=1e-12)
    e0 = float(ev[0])
    assert abs(e0 - analytic.free_e0(z0, [nu], [nd])[0]) <= 1e-10
    gs = edgpu_mod.DeviceVector(G.vecDim).upload(X[:, 0]); del X
    zs = 1j * np.array([0.2, 1.0, 3.0]) + np.array([0.0, 0.4, -0.7])
    for cu, cd in ((0.5, -0.5), (1.0, 1.0)):
        n2, a, b, used = G.chi_lanczos([cu], [cd], gs, 80, threshold=1e-9)
        got = np.array([_cf(z, n2, a, b, used, e0) for z in zs])
        np.testing.assert_allclose(got, _free_chi(z0, nu, nd, cu, cd, zs), rtol=0, atol=1e-8)
    gs.free(); G.close()


def test_free_fermion_e0_orbs_and_hybrid_large(edgpu_mod):
    """ed_total_ud=F with two orbitals and 7 bath levels each (N = 24,010,000: composite two-factor blocks) and a hybrid bath
    (Norb=2, Nbath=10, N = 853,776), interactions off: Lanczos E0 of every storage = sum of one-body levels."""
    from edgpu import model
    imp = np.diag([0.3, -0.3]).reshape(1, 1, 2, 2)
    mo = model.make_model(Norb=2, Nbath=7, ed_total_ud=False, Uloc=(2.0, 2.0), Ust=1.5, Jh=0.25, impHloc=imp,
                          bath_v=np.random.RandomState(8).uniform(0.2, 0.6, (1, 2, 7)), bath_e=np.random.RandomState(9).uniform(-1, 1, (1, 2, 7)))
    mh = model.make_model(Norb=2, Nbath=10, bath_type="hybrid", Uloc=(2.0, 2.0), Ust=1.0, Jh=0.3, Jx=0.3, Jp=0.3,
                          bath_v=np.random.RandomState(3).uniform(0.2, 0.6, (1, 2, 10)))
    for mm, nu, nd in ((mo, [4, 4], [4, 3]), (mh, [6], [6])):
        z = analytic.non_interacting(mm, xmu=0.15)
        ref, gap = analytic.free_e0(z, nu, nd)
        for st in ("kron", "direct", "csr"):
            G = edgpu_mod.HvSector(z, nu, nd, storage=st)
            dv = edgpu_mod.DeviceVector(G.vecDim).zero()
            e0, nl = G.sp_lanc_eigh_device(dv, nitermax=800, threshold=1e-13)
            assert abs(e0 - ref) <= 1e-10, (st, e0, ref, nl)
            dv.free(); G.close()
            edgpu_mod.cache_release()


@pytest.mark.parametrize("nb,nu,nd", [(5, 3, 3), (5, 4, 2), (11, 6, 6), (15, 8, 8), (15, 9, 7)])
def test_spectral_sum_rules_of_the_interacting_model(edgpu_mod, nb, nu, nd):
    """Exact identities of the INTERACTING impurity model (U=2, away from particle-hole symmetry) that need no reference
    solution: for an eigenstate, the zeroth and first moments of the impurity spectral function built from the two
    Green's-function legs are  M0 = <{c,c^+}> = 1  and  M1 = <{[c,H],c^+}> = eps_d - mu + U(<n_-s> - 1/2)  (the hybridisation
    drops out of the anticommutator).  M0 = norm2(c^+ leg) + norm2(c leg); M1 = norm2_p (a_1p - E) - norm2_h (a_1h - E) from
    the first Lanczos coefficient of each leg; <n_-s> from edgpu_imp_histogram.  Up to the c3 sectors."""
    from edgpu import model
    m, _, _ = model.half_filled_c(nb)
    m = dict(m, xmu=0.25)
    G0 = edgpu_mod.HvSector(m, [nu], [nd], storage="kron")
    ev, X, info = G0.sp_eigh(1, nblock=20, nitermax=400, tol=1e-12)
    e0 = float(ev[0])
    gs = edgpu_mod.DeviceVector(G0.vecDim).upload(X[:, 0]); del X
    ob = G0.observables(gs)
    U = m["Uloc"][0]
    for ispin, nother in ((1, ob["dens_dw"][0]), (2, ob["dens_up"][0])):
        m0 = 0.0; m1 = 0.0
        for dagger in (True, False):
            nu2, nd2 = [nu], [nd]
            (nu2 if ispin == 1 else nd2)[0] += 1 if dagger else -1
            Gt = edgpu_mod.HvSector(m, nu2, nd2, storage="direct")
            n2, a, b, used = edgpu_mod.gf_lanczos(G0, Gt, 1, ispin, dagger, gs, 3)
            m0 += n2
            m1 += (1.0 if dagger else -1.0) * n2 * (a[0] - e0)
            Gt.close()
        assert abs(m0 - 1.0) <= 1e-9, (ispin, m0)
        assert abs(m1 - (-m["xmu"] + U * (nother - 0.5))) <= 1e-7, (ispin, m1, -m["xmu"] + U * (nother - 0.5))
    gs.free(); G0.close()


@pytest.mark.parametrize("nb", [5, 11, 15])
def test_symmetries_and_hellmann_feynman(edgpu_mod, nb):
    """More reference-free identities of the interacting model, up to c3 size.  (i) spin symmetry: E0(Nup,Ndw) = E0(Ndw,Nup) for
    sectors off half filling, whose up configurations have different numbers of hops (9 / 7 at c3).  (ii) particle-hole symmetry
    of the half-filled sector (symmetric bath, mu = 0, Hartree-shifted U): the particle and the hole leg of the Green's function have
    the same weight and the same Lanczos coefficients relative to E0.  (iii) Hellmann-Feynman: dE0/dU = <(n_up-1/2)(n_dw-1/2)>,
    central difference of three ground-state searches against the impurity histogram."""
    from edgpu import model
    m, nu, nd = model.half_filled_c(nb)
    h = nu[0]

    def e0_of(mm, a, b, vec=False):
        G = edgpu_mod.HvSector(mm, [a], [b], storage="kron")
        ev, X, info = G.sp_eigh(1, nblock=20, nitermax=400, tol=1e-12, vectors=vec)
        return float(ev[0]), X, G
    ea, _, Ga = e0_of(m, h + 1, h - 1); Ga.close()
    eb, _, Gb = e0_of(m, h - 1, h + 1); Gb.close()
    assert abs(ea - eb) <= 1e-9, (ea, eb)
    e0, X, G0 = e0_of(m, h, h, vec=True)
    gs = edgpu_mod.DeviceVector(G0.vecDim).upload(X[:, 0]); del X
    legs = []; NL = 80
    for dagger in (True, False):
        Gt = edgpu_mod.HvSector(m, [h + (1 if dagger else -1)], [h], storage="kron")
        n2, a, b, used = edgpu_mod.gf_lanczos(G0, Gt, 1, 1, dagger, gs, NL)
        legs.append((n2, a - e0, b, used)); Gt.close()
    assert abs(legs[0][0] - 0.5) <= 1e-7 and abs(legs[0][0] - legs[1][0]) <= 1e-8
    # the first coefficients agree directly; later ones amplify the 1e-8 difference of the two start vectors exponentially (a
    # property of the Lanczos recurrence, not of the kernels), so the legs are compared through what they are for: with enough
    # steps for the continued fractions to have converged, G(i w) = G_particle + G_hole has no real part at particle-hole symmetry
    np.testing.assert_allclose(legs[0][1][:6], legs[1][1][:6], rtol=0, atol=1e-6)
    np.testing.assert_allclose(legs[0][2][1:6], legs[1][2][1:6], rtol=0, atol=1e-6)
    for wn in (0.5, 1.0, 3.0):
        z = 1j * wn
        gp = _cf(z, legs[0][0], legs[0][1] + e0, legs[0][2], legs[0][3], e0)
        gh = 0.0; nh = legs[1][3]
        for k in range(nh - 1, -1, -1):      # hole leg: 1 / (z + (a_k - E0) - b_{k+1}^2 ...)
            bk2 = legs[1][2][k + 1] ** 2 if k + 1 < nh else 0.0
            gh = 1.0 / (z + legs[1][1][k] - bk2 * gh)
        gh *= legs[1][0]
        assert abs((gp + gh).real) <= 1e-6 and (gp + gh).imag < 0, (wn, gp + gh)
    ob = G0.observables(gs)
    rhs = ob["docc"][0] - 0.5 * ob["dens"][0] + 0.25
    gs.free(); G0.close()
    dU = 2e-3
    ep, _, Gp = e0_of(dict(m, Uloc=(m["Uloc"][0] + dU, 0, 0, 0, 0)), h, h); Gp.close()
    em, _, Gm = e0_of(dict(m, Uloc=(m["Uloc"][0] - dU, 0, 0, 0, 0)), h, h); Gm.close()
    assert abs((ep - em) / (2 * dU) - rhs) <= 2e-6, ((ep - em) / (2 * dU), rhs)
