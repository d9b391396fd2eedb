"""Round 4: sectors whose vector fits LDS (up to ~19 000 states) run their Lanczos loops in ONE workgroup (k_lanczos_wg):
sp_lanc_eigh, sp_lanc_tridiag and the Green's-function / susceptibility legs built on it.  Against the launch-per-iteration
loop (EDGPU_LZ_WG=0), the oracle's restatement of SciFortran's recurrence, and dense diagonalisation."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


def _sectors():
    from edgpu import model
    out = []
    m7, _, _ = model.half_filled_c(7)       # Ns = 8
    m9, _, _ = model.half_filled_c(9)       # Ns = 10
    m11, _, _ = model.half_filled_c(11)     # Ns = 12
    for m, nu, nd in ((m7, 4, 4), (m7, 3, 5), (m7, 6, 2), (m7, 1, 4), (m9, 5, 3), (m9, 3, 3), (m9, 2, 6), (m9, 9, 4), (m9, 4, 9),
                      (m11, 2, 3), (m11, 10, 2), (m11, 11, 5), (m11, 1, 6), (m11, 3, 2), (m11, 12, 12)):
        out.append((m, [nu], [nd]))
    return out


@pytest.mark.parametrize("st", ["kron", "direct"])
def test_workgroup_lanczos_equals_the_launch_per_iteration_loop(edgpu_mod, monkeypatch, st):
    for m, nu, nd in _sectors():
        S = oracle.Sector(m, nu, nd)
        v0 = np.random.RandomState(3).uniform(0, 1, S.Dim)
        n = min(40, S.Dim)
        res = {}
        for tag, env in (("wg", {}), ("plain", {"EDGPU_LZ_WG": "0"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            G = edgpu_mod.HvSector(m, nu, nd, storage=st)
            a, b, used = G.sp_lanc_tridiag(v0, n)
            e0, x, nl = G.sp_lanc_eigh(v0, nitermax=min(300, S.Dim), threshold=1e-13)
            ez, xz, nlz = G.sp_lanc_eigh(None, nitermax=min(300, S.Dim), threshold=1e-13)       # library start vector
            res[tag] = (a, b, used, e0, x, nl, ez)
            G.close()
            for k in env:
                monkeypatch.delenv(k)
        a_ref, b_ref, used_ref = oracle.sp_lanc_tridiag(S.spmatvec, v0, n)
        for tag, (a, b, used, e0, x, nl, ez) in res.items():
            assert used == used_ref and b[0] == 0.0, (tag, nu, nd, used, used_ref)
            k = min(used, 16 if S.Dim < 2000 else 30)
            np.testing.assert_allclose(a[:k], a_ref[:k], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(b[1:k], b_ref[1:k], rtol=1e-9, atol=1e-9)
            assert abs(np.linalg.norm(x) - 1.0) < 1e-12
            assert np.linalg.norm(S.spmatvec(x) - e0 * x) < 1e-6, (tag, nu, nd)
        assert abs(res["wg"][3] - res["plain"][3]) <= 1e-10 and abs(res["wg"][5] - res["plain"][5]) <= 1, (nu, nd, res["wg"][3], res["plain"][3], res["wg"][5], res["plain"][5])
        assert abs(res["wg"][6] - res["plain"][6]) <= 1e-10
        if S.Dim <= 5000:
            w = np.linalg.eigvalsh(S.hmat())
            assert abs(res["wg"][3] - w[0]) <= 1e-10 or abs(res["wg"][3] - w[np.argmin(np.abs(w - res["wg"][3]))]) <= 1e-10      # a start vector may miss the lowest level's symmetry class: an eigenvalue in any case


def test_workgroup_green_function_leg(edgpu_mod, monkeypatch):
    """c^+ and c legs from a small sector's ground state (edgpu_gf_lanczos): the workgroup form against the launch-per-iteration one."""
    import edgpu
    from edgpu import model
    m, _, _ = model.half_filled_c(9)
    out = {}
    for tag, env in (("wg", {}), ("plain", {"EDGPU_LZ_WG": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        G = edgpu_mod.HvSector(m, [5], [5], storage="kron")          # 63 504 states: launch-per-iteration loop in both runs
        e0, x, nl = G.sp_lanc_eigh(None, nitermax=300, threshold=1e-13)
        legs = []
        for (tu, td, dag, spin) in ((6, 5, True, 1), (4, 5, False, 1)):
            T = edgpu_mod.HvSector(m, [tu], [td], storage="kron")
            legs.append(edgpu.gf_lanczos(G, T, 1, spin, dag, x, 60))
            T.close()
        G2 = edgpu_mod.HvSector(m, [8], [2], storage="direct")       # 2 025 states: workgroup loop
        e2, x2, _ = G2.sp_lanc_eigh(None, nitermax=300, threshold=1e-13)
        T2 = edgpu_mod.HvSector(m, [9], [2], storage="direct")       # 450 states
        legs.append(edgpu.gf_lanczos(G2, T2, 1, 1, True, x2, 60))
        T2.close(); G2.close(); G.close()
        out[tag] = (e0, e2, legs)
        for k in env:
            monkeypatch.delenv(k)
    assert abs(out["wg"][0] - out["plain"][0]) < 1e-10 and abs(out["wg"][1] - out["plain"][1]) < 1e-10
    for (n2a, aa, ba, ua), (n2b, ab, bb, ub) in zip(out["wg"][2], out["plain"][2]):
        assert abs(n2a - n2b) < 1e-10 and ua == ub
        k = min(ua, 6)       # two different recurrences (normalised / scale-carrying): in a target sector of a few hundred states the Krylov space
        # closes on near-invariant subspaces within ~10 steps (small beta) and rounding differences of the two grow by 1/beta per step
        np.testing.assert_allclose(aa[:k], ab[:k], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(ba[1:k], bb[1:k], rtol=1e-9, atol=1e-9)
        # ... what the host builds from them agrees far beyond that: the continued fraction of the leg at a few frequencies
        for z in (0.3j, 1.0j, -1.5 + 0.2j):
            def cf(a, b, n):
                g = 0.0
                for i in range(n - 1, -1, -1):
                    g = 1.0 / (z - a[i] - (b[i + 1] ** 2 * g if i + 1 < n else 0.0))
                return g
            assert abs(cf(aa, ba, ua) - cf(ab, bb, ub)) < 1e-8 * max(1.0, abs(cf(ab, bb, ub)))


@pytest.mark.parametrize("st", ["kron", "direct"])
def test_self_reducing_iteration_equals_the_four_launch_loop(edgpu_mod, monkeypatch, st):
    """Launch-bound sectors beyond the one-workgroup form (whole product in one kernel, up to 4 M states) run two launches per Lanczos
    iteration: every workgroup of the consuming kernel sums the producing kernel's partial sums itself (k_kron2<KF_SR>, k_axpy_norm_sr).
    Against the four-launch loop (EDGPU_LZ_SR=0) and the oracle: 1-orbital sectors of 14 400 - 853 776 states, a phonon sector and a
    Jx/Jp sector (the row-wise kernel with those terms), sp_lanc_tridiag and sp_lanc_eigh (chunked host checks, stored and regenerated vectors)."""
    from edgpu import model
    import cases
    zoo = {z[0]: z[1:] for z in cases.zoo()}
    m9, _, _ = model.half_filled_c(9)
    m11, nu11, nd11 = model.half_filled_c(11)
    todo = [(m9, [3], [3]), (m9, [5], [5]), (m11, [5], [2]), (m11, nu11, nd11), zoo["phonons_nb3_nph3"], zoo["norb2_normal_JxJp"], zoo["replica_sz_sx_JxJp"]]
    for m, nu, nd in todo:
        S = oracle.Sector(m, nu, nd)
        v0 = np.random.RandomState(7).uniform(0, 1, S.Dim)
        n = min(40, S.Dim)
        res = {}
        for tag, env in (("sr", {}), ("sr_twopass", {"EDGPU_LZ_STORE": "0"}), ("plain", {"EDGPU_LZ_SR": "0"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            G = edgpu_mod.HvSector(m, nu, nd, storage=st)
            a, b, used = G.sp_lanc_tridiag(v0, n)
            e0, x, nl = G.sp_lanc_eigh(v0, nitermax=min(300, S.Dim), threshold=1e-13)
            res[tag] = (a, b, used, e0, x, nl)
            G.close()
            for k in env:
                monkeypatch.delenv(k)
        for tag, (a, b, used, e0, x, nl) in res.items():
            assert used == res["plain"][2] and b[0] == 0.0
            k = min(used, 25)
            np.testing.assert_allclose(a[:k], res["plain"][0][:k], rtol=1e-11, atol=1e-11)      # the same arithmetic up to the order in which the partial sums are added
            np.testing.assert_allclose(b[1:k], res["plain"][1][1:k], rtol=1e-11, atol=1e-11)
            assert abs(e0 - res["plain"][3]) <= 1e-11 and abs(nl - res["plain"][5]) <= 1, (tag, e0, res["plain"][3], nl, res["plain"][5])      # the partial sums are added in another order: the 1e-13 stopping rule may fire one step apart
            assert abs(np.linalg.norm(x) - 1.0) < 1e-12
            assert min(np.abs(x - res["plain"][4]).max(), np.abs(x + res["plain"][4]).max()) < 1e-6, tag      # one Lanczos step more or less: the vector is converged to ~sqrt of the energy threshold
            assert np.linalg.norm(S.spmatvec(x) - e0 * x) < 1e-5, tag
        if S.Dim <= 70000:
            a_ref, b_ref, used_ref = oracle.sp_lanc_tridiag(S.spmatvec, v0, n)
            k = min(used_ref, 16)
            np.testing.assert_allclose(res["sr"][0][:k], a_ref[:k], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(res["sr"][1][1:k], b_ref[1:k], rtol=1e-9, atol=1e-9)


def test_eigsh_two_launch_iteration_equals_the_four_launch_one(edgpu_mod, monkeypatch):
    """edgpu_eigsh on launch-bound sectors runs the self-reducing iteration too; beta_k reaches the host as the partial sums of |w|^2 in pinned
    memory (summed there), alpha_k through the pinned coefficient array.  Same pairs, same number of products and restarts as with
    EDGPU_LZ_SR=0 (four launches, the beta step taken by a reduction kernel), eigenvalues against the dense spectrum; a phonon sector and a
    Jx/Jp sector ride along (re-orthogonalisations and restarts commit the pending step first)."""
    from edgpu import model
    import cases
    zoo = {z[0]: z[1:] for z in cases.zoo()}
    m9, _, _ = model.half_filled_c(9)
    todo = [(m9, [3], [3]), (m9, [5], [4]), zoo["phonons_nb3_nph3"], zoo["norb2_normal_JxJp"]]
    for m, nu, nd in todo:
        S = oracle.Sector(m, nu, nd)
        res = {}
        for tag, env in (("sr", {}), ("plain", {"EDGPU_LZ_SR": "0"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            G = edgpu_mod.HvSector(m, nu, nd, storage="kron")
            nev = min(3, S.Dim - 1)
            ev, X, info = G.sp_eigh(nev, nblock=min(S.Dim, 12), nitermax=400, tol=1e-11)
            res[tag] = (ev, X, info)
            G.close()
            for k in env:
                monkeypatch.delenv(k)
        ev, X, info = res["sr"]
        np.testing.assert_allclose(ev, res["plain"][0], rtol=0, atol=1e-10)
        assert abs(info["matvecs"] - res["plain"][2]["matvecs"]) <= 12 and info["nconv"] == res["plain"][2]["nconv"], (info, res["plain"][2])
        for i in range(len(ev)):
            assert np.linalg.norm(S.spmatvec(X[:, i]) - ev[i] * X[:, i]) < 1e-7 * max(1.0, abs(ev[i]))
        if S.Dim <= 5000:
            w = np.linalg.eigvalsh(S.hmat())
            assert abs(ev[0] - w[0]) <= 1e-9
            for e in ev:      # a single-vector Krylov space holds one copy of a degenerate level
                assert np.abs(w - e).min() <= 1e-9


def test_sector_churn_recycles_streams_and_blocks(edgpu_mod):
    """The host creates and deletes O(Ns^2) sectors per DMFT iteration: streams, small device blocks and the pinned coefficient blocks are
    recycled through process-wide pools (runtime.hpp).  Many create / solve / leg / delete cycles, edgpu_cache_release in between, and the
    answers stay the ones of the first cycle."""
    import edgpu
    from edgpu import model
    m, _, _ = model.half_filled_c(7)
    first = None
    for cycle in range(40):
        G = edgpu_mod.HvSector(m, [4], [3], storage=("kron", "direct", "csr")[cycle % 3])
        T = edgpu_mod.HvSector(m, [5], [3], storage="kron")
        ev, X, info = G.sp_eigh(2, nblock=20, nitermax=300, tol=1e-11)
        n2, a, b, used = edgpu.gf_lanczos(G, T, 1, 1, True, np.ascontiguousarray(X[:, 0]), 30)
        got = (ev.copy(), n2, a[:10].copy(), b[:10].copy())
        T.close(); G.close()
        if first is None:
            first = got
        np.testing.assert_allclose(got[0], first[0], rtol=0, atol=1e-10)
        assert abs(got[1] - first[1]) < 1e-10
        np.testing.assert_allclose(got[2], first[2], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(got[3], first[3], rtol=1e-8, atol=1e-8)
        if cycle % 13 == 12:
            edgpu_mod.cache_release()
