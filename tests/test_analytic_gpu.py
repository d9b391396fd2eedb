"""The HIP path against known answers that do not pass through oracle/ (tests/analytic.py): operator algebra on the Fock
space for small sectors, free-fermion energies and Green's functions and the closed-form dimer -- at every BASELINE size,
including c3 / c4 / c5 where nothing else but the sampled-row oracle reaches.  One test at the end does use the oracle:
the full-size Green's-function target sectors of c3 (rows of different lengths, k_apply_op with 64-bit rows)."""
import numpy as np
import pytest

import analytic
import cases
from oracle import oracle

pytestmark = pytest.mark.gpu
ZOO = cases.zoo()
FOCK = [z for z in ZOO if 2 * z[1]["Ns"] <= 16]
E0_TOL = 1e-10          # |E0(Lanczos on the GPU) - sum of one-body levels|


def _cf(z, norm2, a, b, n, e0, sign):
    """continued fraction of one Lanczos leg (equivalent to the pole expansion of ED_GF_NORMAL.f90:599-654)"""
    g = 0.0
    for k in range(n - 1, -1, -1):
        bk2 = b[k + 1] ** 2 if k + 1 < n else 0.0
        g = 1.0 / (z - sign * (a[k] - e0) - bk2 * g)
    return norm2 * g


@pytest.mark.parametrize("name,m,nu,nd", FOCK, ids=[z[0] for z in FOCK])
def test_gpu_operator_equals_operator_algebra(edgpu_mod, name, m, nu, nd):
    """dense Hmat (edgpu_sector_hmat) element by element, and the product of every storage, against the Jordan-Wigner
    construction -- no oracle in between"""
    F = analytic.fock_hamiltonian(m, nu, nd)
    v = cases.sinvec(F.shape[0]); u = np.random.RandomState(3).uniform(-0.5, 0.5, F.shape[0])
    for st in ("kron", "direct", "csr"):
        G = edgpu_mod.HvSector(m, nu, nd, storage=st)
        assert G.Dim == F.shape[0]
        if st != "csr" and G.Dim <= 1500:
            H = G.hmat()
            assert np.abs(H - F).max() <= 1e-14 * max(1.0, np.abs(F).max()), (st, np.abs(H - F).max())
        for x in (v, u):
            ref = F @ x
            hv = G.spHtimesV(x)
            assert np.abs(hv - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300), (st, np.abs(hv - ref).max())
        G.close()


@pytest.mark.parametrize("U,V", [(0.0, 0.5), (2.0, 0.5), (8.0, 0.3)])
def test_dimer_closed_form(edgpu_mod, U, V):
    """Nbath=1: the whole (1,1) spectrum through edgpu_eigsh and through the dense form"""
    from edgpu import model
    m = model.make_model(Norb=1, Nbath=1, Uloc=(U,), bath_e=np.zeros((1, 1, 1)), bath_v=np.full((1, 1, 1), V))
    ref = analytic.dimer_spectrum(U, V)
    for st in ("kron", "direct", "csr"):
        G = edgpu_mod.HvSector(m, [1], [1], storage=st)
        if st != "csr":
            np.testing.assert_allclose(np.linalg.eigvalsh(G.hmat()), ref, rtol=0, atol=1e-13)
        ev, vec, info = G.sp_eigh(2, nblock=4, tol=1e-13)
        np.testing.assert_allclose(ev, ref[:2], rtol=0, atol=1e-11)
        e0, x, nl = G.sp_lanc_eigh(nitermax=50, threshold=1e-14)
        assert abs(e0 - ref[0]) <= 1e-12
        G.close()


FREE = [z for z in ZOO if z[1].get("Nph", 0) == 0]


@pytest.mark.parametrize("name,m,nu,nd", FREE, ids=[z[0] for z in FREE])
def test_free_fermion_low_spectrum_small_sectors(edgpu_mod, name, m, nu, nd):
    """interactions off: the lowest eigenvalues of edgpu_eigsh are sums of one-body levels (every operator variant)"""
    z = analytic.non_interacting(m, xmu=0.37)
    G = edgpu_mod.HvSector(z, nu, nd, storage="kron")
    if G.Dim <= 1300:
        ref = analytic.free_spectrum(z, nu, nd)
    else:
        ref = np.array([analytic.free_e0(z, nu, nd)[0]])
    nev = min(3, ref.size, G.Dim - 1)
    if G.Dim > 1300:
        ev, vec, info = G.sp_eigh(1, nblock=24, tol=1e-13, nitermax=2000)
        assert abs(ev[0] - ref[0]) <= 1e-10
    elif G.Dim >= 8:
        # interactions off, up and dw excitations cost the same: exactly degenerate levels.  A single-vector Krylov method
        # (ARPACK, the reference's sp_eigh, included) resolves ONE copy of such a level from a generic start vector, so the
        # returned values are checked as members of the spectrum, ascending from E0, none more often than it occurs
        ev, vec, info = G.sp_eigh(nev, nblock=min(G.Dim, 24), tol=1e-13, nitermax=2000)
        assert abs(ev[0] - ref[0]) <= 1e-10 and np.all(np.diff(ev) >= -1e-10)
        levels = ref[np.concatenate(([True], np.diff(ref) > 1e-8))]          # distinct levels, ascending
        assert ev[-1] <= levels[min(nev, levels.size) - 1] + 1e-10          # nothing skipped below the nev-th distinct level
        for x in ev:
            mult = int(np.sum(np.abs(ref - x) <= 1e-9))
            assert mult >= 1, (x, ref[:8])
            assert int(np.sum(np.abs(ev - x) <= 1e-9)) <= mult
    else:
        w = np.linalg.eigvalsh(G.hmat())
        np.testing.assert_allclose(w, ref, rtol=0, atol=1e-12)
    G.close()


def _free_e0_check(edgpu_mod, z, nu, nd, storages, nitermax=600):
    ref, gap = analytic.free_e0(z, nu, nd)
    out = {}
    for st in storages:
        G = edgpu_mod.HvSector(z, nu, nd, storage=st)
        dv = edgpu_mod.DeviceVector(G.vecDim).zero()
        e0, nl = G.sp_lanc_eigh_device(dv, nitermax=nitermax, threshold=1e-13)
        assert abs(e0 - ref) <= E0_TOL, (st, e0, ref, nl)
        out[st] = (e0, nl)
        dv.free(); G.close()
        edgpu_mod.cache_release()
    return ref, out


def test_free_fermion_e0_c1_c2(edgpu_mod):
    from edgpu import model
    for nb in (5, 11):
        m, nu, nd = model.half_filled_c(nb)
        _free_e0_check(edgpu_mod, analytic.non_interacting(m), nu, nd, ("kron", "direct", "csr"))
        _free_e0_check(edgpu_mod, analytic.non_interacting(m, xmu=0.3), [nu[0] - 1], [nd[0] + 1], ("kron", "direct", "csr"))


def test_free_fermion_e0_c3_full_size(edgpu_mod):
    """BASELINE config 3 (Ns=16, N=165,636,900), interactions off: the Lanczos ground state of every storage equals the sum
    of the 8 lowest levels of the 16x16 one-body matrix per spin, -17.223146712820"""
    from edgpu import model
    m, nu, nd = model.half_filled_c(15)
    ref, out = _free_e0_check(edgpu_mod, analytic.non_interacting(m), nu, nd, ("kron", "direct", "csr"))
    assert abs(ref - (-17.223146712820377)) < 1e-12


def test_free_fermion_e0_odd_rows_and_off_half_filling_full_size(edgpu_mod):
    """Analytic anchors for the kernel forms of round 3 at full size, interactions off: Ns=15 (every sector has an odd DimUp: the
    column-pair kernels on pitched vectors), sector (7,8), N=41,409,225; and the Green's-function target sectors of config 3,
    (9,8) and (7,8) at Ns=16 (9 up hops per configuration: the pair form with 10 hop slots), N=147,232,800."""
    from edgpu import model
    m, nu, nd = model.half_filled_c(14)
    assert (nu, nd) == ([7], [8])
    _free_e0_check(edgpu_mod, analytic.non_interacting(m), nu, nd, ("kron", "direct"))
    _free_e0_check(edgpu_mod, analytic.non_interacting(m, xmu=0.2), [6], [8], ("direct",))
    m, _, _ = model.half_filled_c(15)
    _free_e0_check(edgpu_mod, analytic.non_interacting(m), [9], [8], ("kron", "direct"))
    _free_e0_check(edgpu_mod, analytic.non_interacting(m, xmu=-0.1), [7], [8], ("kron",))


def test_free_fermion_e0_c4_full_size(edgpu_mod):
    """BASELINE config 4 (Ns=18, N=2,363,904,400) on one GPU: on-the-fly and stored-factor storage, E0 = -19.218415152809"""
    from edgpu import model
    m, nu, nd = model.half_filled_c(17)
    ref, out = _free_e0_check(edgpu_mod, analytic.non_interacting(m), nu, nd, ("direct", "kron"))
    assert abs(ref - (-19.218415152809488)) < 1e-12


def test_free_fermion_e0_c5_full_size(edgpu_mod):
    """BASELINE config 5 model (2 orbitals, replica bath on sigma_z, Ns=16), interactions off, E0 = -18.381414807149"""
    from edgpu import model
    m = model.make_model(Norb=2, Nbath=7, bath_type="replica", H_basis=[cases.sigma_z()], lambda_imp=[1.0],
                         Uloc=(2.0, 2.0), Jh=0.25, Ust=1.5, Jx=0.25, Jp=0.25)
    ref, out = _free_e0_check(edgpu_mod, analytic.non_interacting(m), [8], [8], ("kron", "direct", "csr"))
    assert abs(ref - (-18.381414807149397)) < 1e-12


@pytest.mark.parametrize("nb,ispin,tstore", [(5, 1, "kron"), (5, 2, "csr"), (11, 1, "kron"), (11, 2, "direct"), (15, 1, "kron"), (15, 2, "csr")])
def test_free_green_function(edgpu_mod, nb, ispin, tstore):
    """U=0 impurity Green's function from the device legs (edgpu_eigsh ground state -> edgpu_gf_lanczos for c^+ and c) against
    1/(z + mu - eps_d - sum_k V_k^2/(z - eps_k)) = [(z-h)^-1]_dd to 1e-8, at c1, c2 and c3 (target sectors (9,8), (7,8), (8,9),
    (8,7) with N = 147,232,800: rows of different lengths in the explicit CSR, 64-bit rows in k_apply_op)"""
    from edgpu import model
    m, nu, nd = model.half_filled_c(nb)
    z0 = analytic.non_interacting(m, xmu=0.2)
    ref_e0, gap = analytic.free_e0(z0, nu, nd)
    G0 = edgpu_mod.HvSector(z0, nu, nd, storage="kron")
    # G is a quadratic form of the ground state: its error is first order in the error of the VECTOR.  The energy criterion of
    # sp_lanc_eigh (|dE| <= 1e-13) leaves ~sqrt(dE/gap) ~ 6e-7 in the vector; the residual criterion of sp_eigh (1e-12 |E|) leaves
    # ~1e-11, which is what a 1e-8 comparison needs
    ev, X, info = G0.sp_eigh(1, nblock=20, nitermax=300, tol=1e-12)
    e0 = float(ev[0])
    assert abs(e0 - ref_e0) <= E0_TOL
    gs = edgpu_mod.DeviceVector(G0.vecDim).upload(X[:, 0])
    del X
    edgpu_mod.cache_release()
    zs = 1j * np.array([0.1, 0.7, 2.0]) + np.array([0.0, 0.3, -1.0])
    tot = np.zeros(3, complex); w = 0.0
    for dagger in (True, False):
        nu2, nd2 = list(nu), list(nd)
        (nu2 if ispin == 1 else nd2)[0] += 1 if dagger else -1
        Gt = edgpu_mod.HvSector(z0, nu2, nd2, storage=tstore)
        n2, a, b, used = edgpu_mod.gf_lanczos(G0, Gt, 1, ispin, dagger, gs, 60, threshold=1e-9)
        # (the Krylov space of c^+|gs> is spanned by the nb+1 one-body levels; the Lanczos ground state carries an error of
        #  ~1e-7 in the vector, so beta does not drop below the threshold there -- the surplus steps carry weights ~1e-14)
        tot += np.array([_cf(zz, n2, a, b, used, e0, 1.0 if dagger else -1.0) for zz in zs])
        w += n2
        Gt.close()
    assert abs(w - 1.0) <= 1e-9        # <n_d> + <1-n_d>
    np.testing.assert_allclose(tot, analytic.free_gf(z0, 1, ispin, zs), rtol=0, atol=1e-8)
    gs.free(); G0.close()


def test_c3_gf_target_sectors_full_size(edgpu_mod):
    """The sectors the Green's-function legs of c3 run in (ED_GF_NORMAL.f90:184-239): (9,8) for c^+_up and (8,7) for c_dw,
    N = 147,232,800, DimUp != DimDw, up configurations with different numbers of hops (level-packed 64-row blocks of the
    explicit CSR).  k_apply_op against the oracle's vectorised statement of the same loop (whole vector), and the product
    of every storage against orc_rows on > 1e4 rows."""
    from edgpu import model
    import importlib
    tp = importlib.import_module("test_gpu_parity")
    m, nu, nd = model.half_filled_c(15)
    S0 = oracle.Sector(m, nu, nd)
    rng = np.random.default_rng(21)
    gs = rng.random(S0.Dim) - 0.5
    G0 = edgpu_mod.HvSector(m, nu, nd, storage="direct")
    dgs = edgpu_mod.DeviceVector(S0.Dim).upload(gs)
    for (ispin, dagger, nu2, nd2) in ((1, True, [9], [8]), (2, False, [8], [7])):
        St = oracle.Sector(m, nu2, nd2)
        assert St.Dim == 11440 * 12870
        vref, n2ref = oracle.apply_op_vec(S0, St, 1, ispin, dagger, gs)
        idx = tp._sample_rows(St.DimUp, St.DimDw, np.random.default_rng(57 + ispin), 12000)
        assert idx.size > 10000
        orc = St.rows(idx, vref)
        dout, dh = edgpu_mod.DeviceVector(St.Dim), edgpu_mod.DeviceVector(St.Dim)
        for st in ("kron", "direct", "csr"):
            Gt = edgpu_mod.HvSector(m, nu2, nd2, storage=st)
            assert Gt.Dim == St.Dim and Gt.DimUp == St.DimUp and Gt.DimDw == St.DimDw
            n2 = edgpu_mod.apply_op(G0, Gt, 1, ispin, dagger, dgs, dout)
            assert abs(n2 - n2ref) <= 1e-12 * n2ref
            np.testing.assert_allclose(dout.download(), vref, rtol=0, atol=1e-15)
            Gt.spHtimesV_device(dout, dh)
            hv = dh.download()
            assert np.abs(hv[idx] - orc).max() <= 1e-12 * np.abs(orc).max(), (st, ispin, np.abs(hv[idx] - orc).max())
            del hv
            Gt.close()
        dout.free(); dh.free()
    dgs.free(); G0.close()


@pytest.mark.parametrize("nb,nu,nd", [(5, 4, 2), (11, 7, 5), (15, 9, 7)])
def test_free_observables(edgpu_mod, nb, nu, nd):
    """Local observables of the non-interacting ground state from the device (edgpu_imp_histogram -> lanc_observables sums,
    ED_OBSERVABLES.f90:150-215) against one-body occupations: <n_s> = sum over filled levels of |<d|level>|^2,
    <n_up n_dw> = <n_up><n_dw>, at c1, c2 and c3 size (sector (9,7): N = 130,873,600, magnetised)."""
    from edgpu import model
    m, _, _ = model.half_filled_c(nb)
    z0 = analytic.non_interacting(m, xmu=-0.15)
    G0 = edgpu_mod.HvSector(z0, [nu], [nd], storage="direct")
    ev, X, info = G0.sp_eigh(1, nblock=20, nitermax=300, tol=1e-12)
    assert abs(ev[0] - analytic.free_e0(z0, [nu], [nd])[0]) <= E0_TOL
    gs = edgpu_mod.DeviceVector(G0.vecDim).upload(X[:, 0]); del X
    nup, ndw = analytic.free_imp_occupations(z0, [nu], [nd])
    ob = G0.observables(gs)
    assert abs(ob["dens_up"][0] - nup[0]) <= 1e-9 and abs(ob["dens_dw"][0] - ndw[0]) <= 1e-9
    assert abs(ob["dens"][0] - (nup[0] + ndw[0])) <= 1e-9 and abs(ob["magz"][0] - (nup[0] - ndw[0])) <= 1e-9
    assert abs(ob["docc"][0] - nup[0] * ndw[0]) <= 1e-9
    le = G0.local_energy(gs)
    assert abs(le["Epot"]) <= 1e-12 and abs(le["Dust"]) <= 1e-12          # no interaction, one orbital
    gs.free(); G0.close()
