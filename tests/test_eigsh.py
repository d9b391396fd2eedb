"""a16 / SURVEY 8f rank 2: the multi-eigenpair solver behind the reference's default
lanc_method=arpack (sp_eigh, ED_DIAG.f90:153-170) against dense diagonalisation of the oracle's Hmat."""
import numpy as np
import pytest

import cases
from oracle import oracle

pytestmark = pytest.mark.gpu
SEL = [z for z in cases.zoo() if z[0] in ("c_nb5_half", "c_nb7_half", "nb5_offhalf_2_4", "norb2_normal_JxJp", "replica_sz_sx_JxJp",
                                         "orbs_norb2_nb3", "phonons_nb3_nph3", "nspin2_random_bath", "norb3_nb1_JxJp", "norb2_hybrid_JxJp")]


@pytest.mark.parametrize("name,m,nu,nd", SEL, ids=[z[0] for z in SEL])
def test_lowest_eigenpairs_match_dense(edgpu_mod, name, m, nu, nd):
    S = oracle.Sector(m, nu, nd)
    w = np.linalg.eigvalsh(S.hmat())
    for neigen, st in ((2, "kron"), (4, "direct")):
        G = edgpu_mod.HvSector(m, nu, nd, storage=st)
        ev, X, info = G.sp_eigh(neigen, nblock=min(S.Dim, 10 * neigen), nitermax=300, tol=1e-12)
        assert info["nconv"] == neigen, info
        np.testing.assert_allclose(ev, w[:neigen], rtol=0, atol=1e-10)
        for i in range(neigen):       # eigenvectors: unit norm, small residual, mutually orthogonal (degenerate levels included)
            x = X[:, i]
            assert abs(np.linalg.norm(x) - 1.0) < 1e-10
            assert np.linalg.norm(S.spmatvec(x) - ev[i] * x) < 1e-8
        assert np.abs(X.T @ X - np.eye(neigen)).max() < 1e-10
        G.close()


@pytest.mark.parametrize("case", ["c_nb7_half", "norb3_nb1_JxJp", "replica_sz_sx_JxJp", "orbs_norb2_nb3"])
def test_eigsh_survives_many_restarts(edgpu_mod, case):
    """Tiny bases (8 vectors for 3 pairs, 14 for 6, 4 for 1): 30-300 thick restarts per search.  The partial re-orthogonalisation
    must keep the basis semi-orthogonal through all of them (the relation-defect term of DESIGN.md section 5; the numpy model
    tests/eigsh_model.py takes 854 products and 283 restarts on norb3_nb1_JxJp with 8 vectors): every returned value is an
    eigenvalue of the dense matrix to 1e-10, the lowest one is E0, and the vectors are orthonormal eigenvectors."""
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    w = np.linalg.eigvalsh(S.hmat())
    G = edgpu_mod.HvSector(m, nu, nd, storage="kron")
    for neigen, nblock in ((3, 8), (6, 14), (1, 4)):
        ev, X, info = G.sp_eigh(neigen, nblock=nblock, nitermax=3000, tol=1e-12)
        assert info["nconv"] == neigen and info["ncv_used"] == nblock, info
        assert abs(ev[0] - w[0]) <= 1e-10
        for i in range(neigen):
            assert np.abs(w - ev[i]).min() <= 1e-10, (neigen, i, ev[i])
            assert np.linalg.norm(S.spmatvec(X[:, i]) - ev[i] * X[:, i]) < 1e-7
        assert np.abs(X.T @ X - np.eye(neigen)).max() < 1e-10
    G.close()


@pytest.mark.parametrize("env", [{"EDGPU_EIGSH_PRO": "0"}, {"EDGPU_EIGSH_FUSED": "0"}, {"EDGPU_ROT2": "0", "EDGPU_ROT_MFMA": "0", "EDGPU_EIGSH_OMEGA_PCT": "1"},
                                 {"EDGPU_ROT_MFMA": "0"}, {"EDGPU_ROT_MFMA": "0", "EDGPU_EIGSH_PRO": "0"}, {"EDGPU_EIGSH_OMEGA_PCT": "2000000000"}])      # never re-orthogonalise: the true-residual check must catch what comes out
def test_eigsh_alternative_forms(edgpu_mod, monkeypatch, env):
    """The forms behind knobs stay correct: every vector orthogonalised against the whole basis (round-2 solver), the
    three-pass iteration inside the partial-reorthogonalisation solver, the one-element rotation kernel with a trigger level
    of 1 % of sqrt(eps) (many more passes over the basis, same answer), the register-resident rotation kernels instead of the
    matrix-core one.  Last case: a trigger level that never fires lets the
    basis lose its orthogonality; the solver's safety net (true residuals of the returned pairs, then the whole-basis form)
    must still hand back correct pairs."""
    from edgpu import model
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m, nu, nd = model.half_filled_c(9)          # N = 63504, split two-pass product with the fused iteration available
    monkeypatch.setenv("EDGPU_KRON_SPLIT", "1")
    S = oracle.Sector(m, nu, nd)
    G = edgpu_mod.HvSector(m, nu, nd, storage="kron")
    e0, x, nl = G.sp_lanc_eigh(nitermax=300, threshold=1e-13)
    ev, X, info = G.sp_eigh(3, nblock=12, nitermax=500, tol=1e-12)
    assert info["nconv"] == 3 and abs(ev[0] - e0) <= 1e-10 and np.all(np.diff(ev) > -1e-10)
    for i in range(3):
        assert np.linalg.norm(S.spmatvec(X[:, i]) - ev[i] * X[:, i]) < 1e-7
    assert np.abs(X.T @ X - np.eye(3)).max() < 1e-10
    G.close()


def test_eigsh_agrees_with_plain_lanczos_on_c2(edgpu_mod):
    from edgpu import model
    m, nu, nd = model.half_filled_c(11)          # N = 853776
    G = edgpu_mod.HvSector(m, nu, nd)
    e0, x, nl = G.sp_lanc_eigh(nitermax=300, threshold=1e-13)
    ev, X, info = G.sp_eigh(2, nblock=20, nitermax=100, tol=1e-11)
    assert info["nconv"] == 2 and abs(ev[0] - e0) < 1e-9 and ev[1] > ev[0]
    r = G.spHtimesV(X[:, 1]) - ev[1] * X[:, 1]
    assert np.linalg.norm(r) < 1e-7
    G.close()


def test_eigsh_large_basis_and_unconverged_is_an_error(edgpu_mod):
    """Nblock = 10*max(Neigen,lanc_nstates_sector)+lanc_ncv_add (ED_DIAG.f90:95-98) exceeds 48 as soon as lanc_nstates_sector
    does 5 (finite temperature): bases up to 96 vectors run as asked, larger requests run with 96 and say so; and a run that
    ends before all requested pairs converged raises instead of handing back Ritz values as eigenvalues."""
    from edgpu import model, capi
    m, nu, nd = model.half_filled_c(7)           # N = 4900
    S = oracle.Sector(m, nu, nd)
    w = np.linalg.eigvalsh(S.hmat())
    G = edgpu_mod.HvSector(m, nu, nd)
    assert capi.lib().edgpu_eigsh_ncv_cap() == 96
    for nblock, used in ((70, 70), (200, 96)):
        ev, X, info = G.sp_eigh(6, nblock=nblock, nitermax=300, tol=1e-12)
        assert info["ncv_used"] == used and info["nconv"] == 6
        np.testing.assert_allclose(ev, w[:6], rtol=0, atol=1e-10)
        assert np.abs(X.T @ X - np.eye(6)).max() < 1e-10
        for i in range(6):
            assert np.linalg.norm(S.spmatvec(X[:, i]) - ev[i] * X[:, i]) < 1e-8
    with pytest.raises(capi.EdgpuError) as e:     # one restart of a 8-vector basis cannot converge 3 pairs of a 4900-state sector
        G.sp_eigh(3, nblock=8, nitermax=1, tol=1e-13)
    assert e.value.code == capi.ENOTCONVERGED
    ev, X, info = G.sp_eigh(3, nblock=8, nitermax=1, tol=1e-13, allow_unconverged=True)
    assert info["nconv"] < 3 and np.all(np.isfinite(ev))
    with pytest.raises(capi.EdgpuError) as e:
        G.sp_dvdson_eigh(3, nitermax=3, tol=1e-13)
    assert e.value.code == capi.ENOTCONVERGED
    G.close()


def test_eigsh_runs_with_the_basis_that_fits(edgpu_mod, monkeypatch):
    """Nblock is a workspace size: when the requested basis does not fit the free device memory (Ns=18 on one GPU: 14 vectors of
    18.9 GB) the search runs with the vectors that do and reports it; below nev + 2 vectors it returns EDGPU_ENOMEM and leaves
    no basis buffers behind.  EDGPU_EIGSH_MEM_MB pretends the device is that full."""
    from edgpu import model, capi
    m, nu, nd = model.half_filled_c(9)           # N = 63504: 0.5 MB per vector
    S = oracle.Sector(m, nu, nd)
    G = edgpu_mod.HvSector(m, nu, nd)
    e0, x, nl = G.sp_lanc_eigh(nitermax=300, threshold=1e-13)
    monkeypatch.setenv("EDGPU_EIGSH_MEM_MB", "5")
    ev, X, info = G.sp_eigh(2, nblock=20, nitermax=2000, tol=1e-11)
    assert info["nconv"] == 2 and 4 <= info["ncv_used"] <= 10, info      # 10 vectors fit in 5 MB, two are held already, three go to the work buffers
    assert abs(ev[0] - e0) < 1e-9 and ev[1] > ev[0]
    for i in range(2):
        assert np.linalg.norm(S.spmatvec(X[:, i]) - ev[i] * X[:, i]) < 1e-7
    G.close()
    G = edgpu_mod.HvSector(m, nu, nd)
    monkeypatch.setenv("EDGPU_EIGSH_MEM_MB", "1")
    with pytest.raises(capi.EdgpuError) as e:
        G.sp_eigh(2, nblock=20, nitermax=300, tol=1e-11)
    assert e.value.code == capi.ENOMEM and "fit" in str(e.value)
    monkeypatch.delenv("EDGPU_EIGSH_MEM_MB")
    ev2, _, info = G.sp_eigh(2, nblock=20, nitermax=300, tol=1e-11)      # and the sector is still usable
    assert info["ncv_used"] == 20 and abs(ev2[0] - e0) < 1e-9
    G.close()


def test_eigsh_argument_errors(edgpu_mod):
    from edgpu import model, capi
    m, nu, nd = model.half_filled_c(1)           # N = 4
    G = edgpu_mod.HvSector(m, nu, nd)
    with pytest.raises(capi.EdgpuError):
        G.sp_eigh(4)                             # Neigen == dim: the reference uses LAPACK there
    ev, X, info = G.sp_eigh(2, nblock=4)
    S = oracle.Sector(m, nu, nd)
    np.testing.assert_allclose(ev, np.linalg.eigvalsh(S.hmat())[:2], atol=1e-12)
    G.close()


@pytest.mark.parametrize("name,m,nu,nd", SEL, ids=[z[0] for z in SEL])
def test_davidson_matches_dense(edgpu_mod, monkeypatch, name, m, nu, nd):
    """lanc_method="dvdson" (sp_dvdson_eigh, ED_DIAG.f90:189-203): Davidson with the diagonal preconditioner.
    The routine itself lives in SciFortran (parity unpinned): checked against dense diagonalisation.  Search spaces of up to 12
    vectors (1-3 pairs) take the fused step (k_dvd_step; EDGPU_DVD_FUSED=0: separate passes), larger ones the separate passes;
    every space restarts with the previous iterate kept (GD+1)."""
    S = oracle.Sector(m, nu, nd)
    w = np.linalg.eigvalsh(S.hmat())
    for neigen, st, fused in ((1, "kron", "1"), (3, "direct", "1"), (2, "kron", "0"), (4, "kron", "1")):
        if neigen >= S.Dim:
            continue
        monkeypatch.setenv("EDGPU_DVD_FUSED", fused)
        G = edgpu_mod.HvSector(m, nu, nd, storage=st)
        ev, X, info = G.sp_dvdson_eigh(neigen, nitermax=600, tol=1e-11)
        assert info["nconv"] == neigen, info
        np.testing.assert_allclose(ev, w[:neigen], rtol=0, atol=1e-9)
        for i in range(neigen):
            x = X[:, i]
            assert abs(np.linalg.norm(x) - 1.0) < 1e-9
            assert np.linalg.norm(S.spmatvec(x) - ev[i] * x) < 1e-7
        assert np.abs(X.T @ X - np.eye(neigen)).max() < 1e-9
        G.close()


def test_davidson_on_c2_needs_fewer_products_than_lanczos(edgpu_mod):
    from edgpu import model
    m, nu, nd = model.half_filled_c(11)          # N = 853776
    G = edgpu_mod.HvSector(m, nu, nd)
    e0, x, nl = G.sp_lanc_eigh(nitermax=300, threshold=1e-13)
    ev, X, info = G.sp_dvdson_eigh(1, nitermax=300, tol=1e-10)
    assert info["nconv"] == 1 and abs(ev[0] - e0) < 1e-9
    assert info["matvecs"] < 2 * nl              # Lanczos regenerates the basis in a second pass
    G.close()


def test_davidson_two_pass_step_against_separate_passes(edgpu_mod, monkeypatch):
    """The small-space Davidson step (k_dvd_step + k_dvd_orth: new column of the projected matrix from the W_i.t summed inside the
    step kernel, new basis vector written normalised by the orthogonalisation pass, thick-restart rotation written by the step
    kernel itself) against the separate passes (EDGPU_DVD_FUSED=0: residual, multi-dot, update, scale, rotation kernels): same
    pairs, about the same number of products.  Ns=7 (3,4): N = 1225 is odd (the kernels' last-element branch, rotation included);
    c2: a space of 8 vectors restarts every fourth product over ~100 products."""
    from edgpu import model
    for nbath, neigen in ((6, 2), (6, 3), (11, 2)):
        m, nu, nd = model.half_filled_c(nbath)
        res = {}
        for fused in ("1", "0"):
            monkeypatch.setenv("EDGPU_DVD_FUSED", fused)
            G = edgpu_mod.HvSector(m, nu, nd, storage="kron")
            ev, X, info = G.sp_dvdson_eigh(neigen, nitermax=900, tol=1e-11)
            assert info["nconv"] == neigen, (nbath, fused, info)
            for i in range(neigen):
                x = np.ascontiguousarray(X[:, i])
                assert abs(np.linalg.norm(x) - 1.0) < 1e-9
                assert np.linalg.norm(G.spHtimesV(x) - ev[i] * x) < 1e-7, (nbath, fused, i)
            assert np.abs(X.T @ X - np.eye(neigen)).max() < 1e-9
            res[fused] = (ev, info["matvecs"])
            G.close()
        assert np.abs(res["1"][0] - res["0"][0]).max() < 1e-9, (nbath, res)
        assert res["1"][1] <= 1.2 * res["0"][1] + 8, (nbath, res)      # the search itself is the same method: product counts stay close
        if nbath == 6:
            S = oracle.Sector(m, nu, nd)
            assert S.Dim % 2 == 1
            np.testing.assert_allclose(res["1"][0], np.linalg.eigvalsh(S.hmat())[:neigen], rtol=0, atol=1e-9)
