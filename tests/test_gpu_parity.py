"""GPU parity: libedgpu.so (through the C ABI) against the CPU oracle on the same inputs.

Integer/index work is bit-exact; the fp64 products are compared with
  max|Hv_gpu - Hv_oracle| <= 1e-12 * max|Hv_oracle|
(the reference's own stored-vs-direct spread is 3e-15 at |Hv|~1e3; the GPU sums
the same terms in a different order)."""
import numpy as np
import pytest

import cases
from oracle import oracle

pytestmark = pytest.mark.gpu
HXV_RTOL = 1e-12
ZOO = cases.zoo()
IDS = [z[0] for z in ZOO]


def _sample_rows(DimUp, DimDw, rng, nrandom, parts=8, tile=64, above=None):
    """0-based rows for the sampled-row oracle check of a full-size sector: `nrandom` uniform rows, `nrandom//4` rows above
    `above` (Ns=18: flat index > 2^31), the first and last row of every dw block of a `parts`-rank split
    (ED_HAMILTONIAN.f90:96-110) and of its neighbours, and both ends of some `tile`-column tiles."""
    N = DimUp * DimDw
    idx = [rng.integers(0, N, nrandom)]
    if above is not None and above < N:
        idx.append(rng.integers(above, N, max(nrandom // 4, 1)))
        idx.append(np.array([above - 1, above, above + 1, N - 1]))
    q, r = divmod(DimDw, parts)
    edges = []
    for k in range(parts):
        d0 = k * (q + 1) if k < r else k * q + r
        qd = q + 1 if k < r else q
        edges += [d0, d0 + qd - 1]
    edges = np.unique(np.clip(np.array(edges + [e + 1 for e in edges] + [e - 1 for e in edges]), 0, DimDw - 1))
    ups = np.array([0, 1, DimUp // 2, DimUp - 2, DimUp - 1])
    idx.append((edges[:, None] * DimUp + ups[None, :]).ravel())
    tiles = rng.integers(0, max(DimUp // tile, 1), 24)
    tcols = np.unique(np.clip(np.concatenate([tiles * tile, tiles * tile + tile - 1, tiles * tile + tile]), 0, DimUp - 1))
    trows = rng.integers(0, DimDw, 12)
    idx.append((trows[:, None] * DimUp + tcols[None, :]).ravel())
    return np.unique(np.concatenate(idx).astype(np.int64))


def _storages(m):
    # explicit CSR for every sector: composite blocks for ed_total_ud=F, compressed-row non-local block for Jx/Jp, stored
    # electronic factor of the e-ph coupling for phonons
    return ["kron", "direct", "csr"]


@pytest.mark.parametrize("name,m,nu,nd", ZOO, ids=IDS)
def test_sector_maps_and_factors_bit_exact(edgpu_mod, name, m, nu, nd):
    S = oracle.Sector(m, nu, nd).buildh()
    G = edgpu_mod.HvSector(m, nu, nd)
    assert G.isector == S.isector and G.Dim == S.Dim and G.vecDim_Hv_sector() == S.Dim
    for k in range(2 * S.Ns_Ud):
        np.testing.assert_array_equal(G.map(k), S.map(k))
    for iud in range(S.Ns_Ud):
        for which in ("up", "dw"):
            rp, c, v = S.sp(which, iud); rp2, c2, v2 = G.sp(which, iud)
            np.testing.assert_array_equal(rp2, rp); np.testing.assert_array_equal(c2, c)
            np.testing.assert_array_equal(v2, v)   # values are +-t: bit-exact incl. sign
    if m.get("ed_total_ud", True) and m["Norb"] > 1 and (m.get("Jx", 0) or m.get("Jp", 0)):
        rp, c, v = S.sp("nd"); rp2, c2, v2 = G.sp("nd")
        np.testing.assert_array_equal(rp2, rp); np.testing.assert_array_equal(c2, c); np.testing.assert_array_equal(v2, v)
    # spH0d: the device evaluates the reference's statement order with unfused IEEE ops
    d_ref = S.sp("d")[2]
    np.testing.assert_array_equal(G.diag(), d_ref)
    G.close()


@pytest.mark.parametrize("name,m,nu,nd", ZOO, ids=IDS)
def test_hxv_matches_oracle(edgpu_mod, name, m, nu, nd):
    S = oracle.Sector(m, nu, nd)
    for v in (cases.sinvec(S.Dim), np.random.RandomState(11).uniform(-0.5, 0.5, S.Dim)):
        ref = S.spmatvec(v)
        tol = HXV_RTOL * max(np.abs(ref).max(), 1e-300)
        for st in _storages(m):
            G = edgpu_mod.HvSector(m, nu, nd, storage=st)
            hv = G.spHtimesV(v)
            assert np.abs(hv - ref).max() <= tol, (st, np.abs(hv - ref).max(), tol)
            if S.Dim < 2 ** 31:
                np.testing.assert_array_equal(G.spHtimesV_i32(v), hv)
            G.close()


@pytest.mark.parametrize("g", cases.golden_cases(), ids=lambda g: g["name"])
def test_hxv_against_reference_known_answers(edgpu_mod, g):
    m = cases.model_from_golden(g)
    for st in _storages(m):
        G = edgpu_mod.HvSector(m, g["nups"], g["ndws"], storage=st)
        assert G.isector == g["isector"] and G.Dim == g["N"]
        v = cases.sinvec(G.Dim)
        hv = G.spHtimesV(v)
        scale = np.abs(hv).sum()
        if g.get("unpinned"):      # fitted record (see its _note): only isector and N are known answers
            G.close()
            continue
        assert abs(hv.sum() - g["sum_Hv"]) <= 1e-12 * scale
        if "v_Hv" in g:
            assert abs(v @ hv - g["v_Hv"]) <= 1e-12 * scale
        if "Hv_1_3" in g:
            np.testing.assert_allclose(hv[:3], g["Hv_1_3"], rtol=1e-13)
        G.close()


@pytest.mark.parametrize("name,m,nu,nd", [z for z in ZOO if z[0] not in ("c_nb7_half",)], ids=[z[0] for z in ZOO if z[0] not in ("c_nb7_half",)])
def test_dense_hmat_matches_oracle(edgpu_mod, name, m, nu, nd):
    """build_Hv_sector(isector,Hmat): the dense form every sector with dim <= lanc_dim_threshold takes
    (ED_DIAG.f90:105-107) against the oracle's restatement of ed_buildh_main/orbs with Hmat -- element by element."""
    S = oracle.Sector(m, nu, nd)
    Href = S.hmat()
    for st in ("kron", "direct"):
        G = edgpu_mod.HvSector(m, nu, nd, storage=st)
        H = G.hmat()
        assert np.abs(H - Href).max() <= 1e-15 * max(1.0, np.abs(Href).max()), (st, np.abs(H - Href).max())
        G.close()


def test_explicit_csr_rows_equal_dense(edgpu_mod):
    from edgpu import model
    m, nu, nd = model.half_filled_c(5)
    S = oracle.Sector(m, nu, nd); H = S.hmat()
    G = edgpu_mod.HvSector(m, nu, nd, storage="csr")
    rp, cols, vals = G.csr_rows(0, S.Dim)
    assert rp[0] == 0 and rp[-1] == G.info.nnz_total == np.count_nonzero(H) + np.count_nonzero(np.diag(H) == 0)
    D = np.zeros_like(H)
    for i in range(S.Dim):
        c = cols[rp[i]:rp[i + 1]]
        assert np.all(np.diff(c) > 0)          # canonical CSR: ascending columns
        D[i, c] = vals[rp[i]:rp[i + 1]]
    np.testing.assert_array_equal(D, H)
    G.close()


@pytest.mark.gpu
@pytest.mark.parametrize("il,tm", [("3", "1"), ("3", "0"), ("1", "1"), ("2", "1"), ("0", "1")])
def test_explicit_csr_rows_of_the_split_lane_major_blocks(edgpu_mod, monkeypatch, il, tm):
    """Split explicit CSR (local + dw block) with the 64-row blocks stored lane-major (k_csr_il): the row export undoes the
    interleave -- whole matrix and windows that start / end inside a block (DimUp = 70: blocks of 64 and 6 rows)."""
    from edgpu import model
    m, nu, nd = model.half_filled_c(7)
    S = oracle.Sector(m, nu, nd); H = S.hmat()
    monkeypatch.setenv("EDGPU_CSR_SPLIT", "1"); monkeypatch.setenv("EDGPU_CSR_IL", il); monkeypatch.setenv("EDGPU_CSR_TM", tm)
    G = edgpu_mod.HvSector(m, nu, nd, storage="csr")
    for r0, n in ((0, S.Dim), (37, 200), (64, 6), (69, 2), (S.Dim - 5, 5)):
        rp, cols, vals = G.csr_rows(r0, n)
        for i in range(n):
            c = cols[rp[i]:rp[i + 1]]
            assert np.all(np.diff(c) > 0)
            row = np.zeros(S.Dim); row[c] = vals[rp[i]:rp[i + 1]]
            np.testing.assert_array_equal(row, H[r0 + i])
    v = np.random.RandomState(5).uniform(-0.5, 0.5, S.Dim)
    ref = S.spmatvec(v)
    assert np.abs(G.spHtimesV(v) - ref).max() <= HXV_RTOL * np.abs(ref).max()
    G.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nu,nd,jag", [(4, 5, "1"), (4, 5, "0"), (6, 3, "1"), (3, 3, "1")])
def test_split_csr_with_rows_of_different_lengths(edgpu_mod, monkeypatch, nu, nd, jag):
    """Away from half filling the up configurations differ in their number of hops (impurity level occupied or not), so the
    rows of a 64-row block of the local block differ in length: they are packed level by level (k_csr_il, jagged path;
    EDGPU_CSR_JAG=0: row-major block + LDS-staged kernel).  Ns=10: DimUp = 210 (3 blocks of 64 and one of 18) / 120."""
    from edgpu import model
    m = model.make_model(Norb=1, Nbath=9, Uloc=(2.0,), xmu=0.1, hfmode=False)
    S = oracle.Sector(m, [nu], [nd])
    v = np.random.RandomState(11).uniform(-0.5, 0.5, S.Dim)
    ref = S.spmatvec(v)
    monkeypatch.setenv("EDGPU_CSR_SPLIT", "1"); monkeypatch.setenv("EDGPU_CSR_JAG", jag)
    G = edgpu_mod.HvSector(m, [nu], [nd], storage="csr")
    assert ("lane-major-blocks=local" in G.describe()) == (jag == "1")
    assert np.abs(G.spHtimesV(v) - ref).max() <= HXV_RTOL * np.abs(ref).max()
    for r0, n in ((0, 700), (61, 9), (S.Dim - 300, 300), (12345 % S.Dim, 257)):
        rp, cols, vals = G.csr_rows(r0, n)
        y = np.array([vals[rp[i]:rp[i + 1]] @ v[cols[rp[i]:rp[i + 1]]] for i in range(n)])
        assert all(np.all(np.diff(cols[rp[i]:rp[i + 1]]) > 0) for i in range(n))
        assert np.abs(y - ref[r0:r0 + n]).max() <= 1e-12 * np.abs(ref).max()
    G.close()


LANC = [z for z in ZOO if z[0] in ("c_nb5_half", "c_nb7_half", "norb2_normal_JxJp", "replica_sz_sx_JxJp", "orbs_norb2_nb3", "phonons_nb3_nph3", "nb6_even")]


@pytest.mark.parametrize("name,m,nu,nd", LANC, ids=[z[0] for z in LANC])
def test_lanczos_ground_state(edgpu_mod, name, m, nu, nd):
    """E0 against dense diagonalisation of the oracle's Hmat: |dE0| <= 1e-10;
    eigenvector: residual |H x - E0 x| small, unit norm."""
    S = oracle.Sector(m, nu, nd)
    w = np.linalg.eigvalsh(S.hmat())
    for st in ("kron", "direct"):
        G = edgpu_mod.HvSector(m, nu, nd, storage=st)
        e0, x, nl = G.sp_lanc_eigh(nitermax=min(512, S.Dim), threshold=1e-14)
        assert abs(e0 - w[0]) <= 1e-10, (st, e0, w[0], nl)
        assert abs(np.linalg.norm(x) - 1.0) < 1e-12
        assert np.linalg.norm(S.spmatvec(x) - e0 * x) < 1e-6
        G.close()


@pytest.mark.parametrize("name,m,nu,nd", LANC[:4], ids=[z[0] for z in LANC[:4]])
def test_lanczos_same_start_same_coefficients(edgpu_mod, name, m, nu, nd):
    """Same start vector => same alpha/beta as the oracle's restatement (k<=30: rel 1e-9)
    and same E0 from sp_lanc_eigh."""
    S = oracle.Sector(m, nu, nd)
    v0 = np.random.RandomState(5).uniform(0, 1, S.Dim)
    n = 30
    a_ref, b_ref, used_ref = oracle.sp_lanc_tridiag(S.spmatvec, v0, n)
    G = edgpu_mod.HvSector(m, nu, nd)
    a, b, used = G.sp_lanc_tridiag(v0, n)
    assert used == used_ref
    np.testing.assert_allclose(a[:used], a_ref[:used], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(b[1:used], b_ref[1:used], rtol=1e-9, atol=1e-9)
    assert b[0] == 0.0
    e_ref, x_ref, _ = oracle.sp_lanc_eigh(S.spmatvec, v0, min(300, S.Dim), threshold=1e-14)
    e, x, _ = G.sp_lanc_eigh(v0, nitermax=min(300, S.Dim), threshold=1e-14)
    assert abs(e - e_ref) <= 1e-10
    assert min(np.linalg.norm(x - x_ref), np.linalg.norm(x + x_ref)) < 1e-6
    G.close()


def test_c2_size_properties(edgpu_mod):
    """BASELINE config 2 (Ns=12, N=853776): oracle parity on the full vector, plus
    size-independent properties: symmetry <u,Hv>=<Hu,v> and linearity."""
    from edgpu import model
    m, nu, nd = model.half_filled_c(11)
    S = oracle.Sector(m, nu, nd)
    assert S.Dim == 853776
    rng = np.random.RandomState(3)
    u, v = rng.uniform(-0.5, 0.5, S.Dim), rng.uniform(-0.5, 0.5, S.Dim)
    ref = S.spmatvec(v)
    for st in ("kron", "direct", "csr"):
        G = edgpu_mod.HvSector(m, nu, nd, storage=st)
        hv, hu = G.spHtimesV(v), G.spHtimesV(u)
        assert np.abs(hv - ref).max() <= HXV_RTOL * np.abs(ref).max()
        assert abs(u @ hv - hu @ v) <= 1e-11 * (np.abs(u) @ np.abs(hv))
        hl = G.spHtimesV(2.0 * u - 3.0 * v)
        assert np.abs(hl - (2.0 * hu - 3.0 * hv)).max() <= 1e-12 * np.abs(hl).max()
        G.close()


def test_device_pointer_path_and_errors(edgpu_mod):
    from edgpu import model, capi
    m, nu, nd = model.half_filled_c(5)
    G = edgpu_mod.HvSector(m, nu, nd)
    v = cases.sinvec(G.Dim)
    dv, dh = edgpu_mod.DeviceVector(G.Dim).upload(v), edgpu_mod.DeviceVector(G.Dim)
    G.spHtimesV_device(dv, dh)
    np.testing.assert_array_equal(dh.download(), G.spHtimesV(v))
    with pytest.raises(capi.EdgpuError):          # wrong Nloc, reference: stop "Nloc != dim(isector)"
        G.spHtimesV(v[:-1])
    with pytest.raises(capi.EdgpuError):
        edgpu_mod.HvSector(dict(m, Norb=6), nu, nd)
    ms = G.hxv_time(dv, dh, 1, 3)
    assert ms > 0
    G.close()


FORCED = [z for z in ZOO if z[0] in ("orbs_norb2_nb3", "phonons_orbs", "c_nb7_half", "nb5_offhalf_2_4", "nb6_even", "nspin2_random_bath", "norb2_normal_offdiag_hloc", "c_nb5_half",
                                    "norb2_normal_JxJp", "norb2_hybrid_JxJp", "replica_sz_sx_JxJp", "norb3_nb1_JxJp")]   # Jx/Jp: the non-local pass


@pytest.mark.parametrize("name,m,nu,nd", FORCED, ids=[z[0] for z in FORCED])
def test_large_sector_code_paths_on_small_sectors(edgpu_mod, monkeypatch, name, m, nu, nd):
    """The two-pass products (persistent local pass + column-tiled dw pass; split explicit CSR)
    switch on automatically only for sectors that outgrow the caches.  Force them on small sectors
    and compare with the oracle; also the non-default variants and odd tile widths."""
    S = oracle.Sector(m, nu, nd)
    v = np.random.RandomState(21).uniform(-0.5, 0.5, S.Dim)
    ref = S.spmatvec(v)
    tol = HXV_RTOL * np.abs(ref).max()
    w = np.linalg.eigvalsh(S.hmat())
    for env in ({"EDGPU_KRON_SPLIT": "1", "EDGPU_CSR_SPLIT": "1"},
                {"EDGPU_COMPOSITE": "0"},      # ed_total_ud=F through the generic mixed-radix kernel instead of the composite two-factor path (no CSR there)
                {"EDGPU_KRON_SPLIT": "1", "EDGPU_CSR_SPLIT": "1", "EDGPU_DW_W": "8", "EDGPU_DW_ROWS": "16", "EDGPU_CSRD_ROWS": "16", "EDGPU_UP_GRID": "3", "EDGPU_CSR_IL": "0"},      # explicit CSR through the LDS-staged row-major kernels
                {"EDGPU_CSR_SPLIT": "1", "EDGPU_CSR_IL": "2", "EDGPU_CSR_TM": "0"},      # lane-major dw block only, blocks in row order
                {"EDGPU_KRON_SPLIT": "1", "EDGPU_DW_VARIANT": "1", "EDGPU_UP_PERSISTENT": "0", "EDGPU_CSR_SPLIT": "1", "EDGPU_CSRD_NT": "0", "EDGPU_CSR_UNI": "0", "EDGPU_CSR_IL": "1"},
                {"EDGPU_CSR_SPLIT": "0", "EDGPU_KRON_SPLIT": "1", "EDGPU_UP_THREADS": "512", "EDGPU_UP_PAIRS": "0"},
                {"EDGPU_CSR_SPLIT": "1", "EDGPU_CSR_PITCH": "0", "EDGPU_KRON_SPLIT": "1"},      # explicit CSR with columns in the dense vector space
                {"EDGPU_KRON_SPLIT": "1", "EDGPU_DEBUG": "4"},                                  # long-row path: one workgroup per row segment, halo in LDS (k_uppass_halo)
                {"EDGPU_KRON_SPLIT": "1", "EDGPU_DEBUG": "4", "EDGPU_UP_GRID": "24", "EDGPU_LZ_FUSED": "0"},     # ... with idle workgroups (3 per XCD, 2 segments)
                {"EDGPU_KRON_SPLIT": "1", "EDGPU_DEBUG": "4", "EDGPU_UP_SEGMAX": "64"},      # ... with many short segments (up to 64 per row)
                {"EDGPU_KRON_SPLIT": "1", "EDGPU_DEBUG": "4", "EDGPU_UP_HALO": "0", "EDGPU_ROW_SPLIT": "3"},       # long-row fallback: no LDS staging, packed 8-dword image (k_uppass_big)
                {"EDGPU_KRON_SPLIT": "1", "EDGPU_DEBUG": "4", "EDGPU_UP_BIG": "0", "EDGPU_ROW_SPLIT": "1"}):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        for st in _storages(m):
            if st == "csr" and env.get("EDGPU_COMPOSITE") == "0" and not m.get("ed_total_ud", True):
                continue
            G = edgpu_mod.HvSector(m, nu, nd, storage=st)
            hv = G.spHtimesV(v)
            assert np.abs(hv - ref).max() <= tol, (env, st, np.abs(hv - ref).max())
            e0, x, nl = G.sp_lanc_eigh(nitermax=min(400, S.Dim), threshold=1e-14)
            assert abs(e0 - w[0]) <= 1e-10, (env, st)
            if st == "csr":
                rp, cols, vals = G.csr_rows(0, min(S.Dim, 500))
                H = S.hmat()
                for i in range(min(S.Dim, 500)):
                    c = cols[rp[i]:rp[i + 1]]
                    assert np.all(np.diff(c) > 0)
                    row = np.zeros(S.Dim); row[c] = vals[rp[i]:rp[i + 1]]
                    if m.get("Nph", 0) == 0:
                        np.testing.assert_array_equal(row, H[i])
                    else:      # w0*iph and g*sqrt(iph) are formed in a different order than the oracle's dense build
                        np.testing.assert_allclose(row, H[i], rtol=0, atol=1e-14)
            G.close()
        for k in env:
            monkeypatch.delenv(k)


def test_c3_full_size_properties(edgpu_mod):
    """BASELINE config 3 (Ns=16, N=165,636,900) at full size.  The oracle's full product cannot go there (its stored
    matrix is 3.3e8 heap blocks), its sampled-row evaluator can: every storage is compared with orc_rows on > 2e4 rows.
    On top: the storages agree with each other on the whole vector, H is symmetric (<u,Hv> = <Hu,v>), and a short
    Lanczos run gives the same coefficients for every storage."""
    from edgpu import model
    m, nu, nd = model.half_filled_c(15)
    rng = np.random.default_rng(5)
    N = 165636900
    u = rng.random(N) - 0.5
    v = rng.random(N) - 0.5
    du, dv, dh = edgpu_mod.DeviceVector(N).upload(u), edgpu_mod.DeviceVector(N).upload(v), edgpu_mod.DeviceVector(N)
    # ORACLE parity at full size: > 2e4 rows (random, dw-block edges of an 8-rank split, 64-column tile ends) of every
    # storage against orc_rows, which evaluates the reference's statements for single rows from the bit patterns
    S = oracle.Sector(m, nu, nd)
    assert S.Dim == N
    idx = _sample_rows(12870, 12870, np.random.default_rng(55), 20000)
    assert idx.size > 20000
    orc = S.rows(idx, v)
    ref_hv = None; coeffs = {}
    for st in ("kron", "direct", "csr"):
        G = edgpu_mod.HvSector(m, nu, nd, storage=st)
        assert G.Dim == N and G.isector == 1 + 8 * 17 + 8
        G.spHtimesV_device(dv, dh); hv = dh.download()
        assert np.abs(hv[idx] - orc).max() <= HXV_RTOL * np.abs(orc).max(), (st, np.abs(hv[idx] - orc).max())
        G.spHtimesV_device(du, dh); hu = dh.download()
        scale = float(np.abs(u) @ np.abs(hv))
        assert abs(float(u @ hv) - float(hu @ v)) <= 1e-12 * scale, st
        if ref_hv is None:
            ref_hv = hv
        else:
            assert np.abs(hv - ref_hv).max() <= 1e-12 * np.abs(ref_hv).max(), st
        del hv, hu
        G.bench_begin(8); G.bench_steps(6)
        _, a, b = G.bench_result(0, ncoef=6)
        coeffs[st] = (a, b)
        G.close()
    for st in ("direct", "csr"):
        np.testing.assert_allclose(coeffs[st][0], coeffs["kron"][0], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(coeffs[st][1], coeffs["kron"][1], rtol=1e-10, atol=1e-10)


def test_c4_full_size_64bit_indexing(edgpu_mod):
    """BASELINE config 4 (Ns=18, N=2,363,904,400 > 2^31: the reference's int32 row index overflows here,
    direct_mpi/HxV_local.f90:2-5) on ONE GPU with on-the-fly storage: H symmetric on random vectors, and the
    first Lanczos coefficients equal to those of the stored (KRON) format, which runs different local kernels."""
    from edgpu import model
    m, nu, nd = model.half_filled_c(17)
    N = 2363904400
    G = edgpu_mod.HvSector(m, nu, nd, storage="direct")
    assert G.Dim == N and G.isector == 1 + 9 * 19 + 9 and G.vecDim == N
    rng = np.random.default_rng(7)
    du, dv, dh = edgpu_mod.DeviceVector(N), edgpu_mod.DeviceVector(N), edgpu_mod.DeviceVector(N)
    u = rng.random(N) - 0.5; du.upload(u)
    v = rng.random(N) - 0.5; dv.upload(v)
    # ORACLE parity: sampled rows incl. 5000 with flat index > 2^31, the rows either side of 2^31, the first/last rows of
    # every dw block of the 8-rank split and 64-column tile ends, against orc_rows (64-bit row numbers)
    S = oracle.Sector(m, nu, nd)
    idx = _sample_rows(48620, 48620, np.random.default_rng(77), 20000, above=2 ** 31)
    assert (idx >= 2 ** 31).sum() >= 5000
    orc = S.rows(idx, v)
    G.spHtimesV_device(dv, dh); h = dh.download()
    assert np.abs(h[idx] - orc).max() <= HXV_RTOL * np.abs(orc).max(), np.abs(h[idx] - orc).max()
    a = float(u @ h); scale = float(np.abs(u) @ np.abs(h))
    G.spHtimesV_device(du, dh); h = dh.download()
    b = float(h @ v)
    assert abs(a - b) <= 1e-12 * scale
    del u, h
    du.free()
    G.bench_begin(5); G.bench_steps(3)
    _, a_d, b_d = G.bench_result(0, ncoef=3)
    G.close()
    G = edgpu_mod.HvSector(m, nu, nd, storage="kron")
    G.spHtimesV_device(dv, dh); h = dh.download()           # the stored format against the oracle rows as well
    assert np.abs(h[idx] - orc).max() <= HXV_RTOL * np.abs(orc).max(), np.abs(h[idx] - orc).max()
    del h, v
    dv.free(); dh.free()
    G.bench_begin(5); G.bench_steps(3)
    _, a_k, b_k = G.bench_result(0, ncoef=3)
    G.close()
    np.testing.assert_allclose(a_d, a_k, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(b_d, b_k, rtol=1e-10, atol=1e-10)
    assert np.all(np.isfinite(a_d)) and np.all(b_d > 0)


def test_edge_sectors(edgpu_mod):
    """Degenerate shapes the reference can meet: one-state sectors, no hopping at all (V=0: empty H_up/H_dw),
    five orbitals, the largest phonon cut-off used in the zoo, and a sector with a single dw column."""
    from edgpu import model
    edge = []
    m = model.make_model(Norb=1, Nbath=3, Uloc=(2.0,))
    edge += [("vacuum", m, [0], [0]), ("full", m, [4], [4]), ("one_dw_column", m, [2], [0])]
    m0 = model.make_model(Norb=1, Nbath=3, Uloc=(2.0,), bath_v=np.zeros((1, 1, 3)))
    edge += [("no_hybridisation", m0, [2], [2])]
    m5 = model.make_model(Norb=5, Nbath=1, Uloc=(2.0, 1.9, 1.8, 1.7, 1.6), Ust=1.0, Jh=0.2, Jx=0.2, Jp=0.2)
    edge += [("five_orbitals", m5, [5], [5])]
    for name, m, nu, nd in edge:
        S = oracle.Sector(m, nu, nd)
        v = np.random.RandomState(3).uniform(-0.5, 0.5, S.Dim)
        ref = S.spmatvec(v)
        tol = HXV_RTOL * max(np.abs(ref).max(), 1e-300)
        for st in _storages(m):
            G = edgpu_mod.HvSector(m, nu, nd, storage=st)
            assert G.Dim == S.Dim
            assert np.abs(G.spHtimesV(v) - ref).max() <= tol, (name, st)
            if S.Dim <= 4000:
                w = np.linalg.eigvalsh(S.hmat())
                e0, x, nl = G.sp_lanc_eigh(v if np.abs(v).max() > 0 else None, nitermax=300, threshold=1e-14)
                assert abs(e0 - w[0]) <= 1e-10, (name, st, e0, w[0])
            G.close()


def test_bad_descriptors_are_rejected(edgpu_mod):
    import ctypes as C
    from edgpu import model, capi
    m, nu, nd = model.half_filled_c(3)
    L = capi.lib()
    for bad in (dict(Nspin=3), dict(Norb=0), dict(bath_type="hybrid", ed_total_ud=False)):
        mm = dict(m); mm.update(bad)
        with pytest.raises(capi.EdgpuError):
            edgpu_mod.HvSector(mm, nu, nd)
    with pytest.raises(capi.EdgpuError):
        edgpu_mod.HvSector(m, [9], [1])                       # quantum number out of range
    with pytest.raises(capi.EdgpuError):
        edgpu_mod.HvSector(m, nu, nd, rank=0, size=50)        # DimDw < ranks
    G = edgpu_mod.HvSector(m, nu, nd)
    G._desc.abi_version = 99
    h = C.c_void_p()
    assert L.edgpu_sector_create(C.byref(G._desc), C.byref(h)) != 0 and b"abi_version" in L.edgpu_last_error()
    assert L.edgpu_hxv_device(None, None, None) != 0
    G.close()


def test_ground_state_with_stored_krylov_vectors(edgpu_mod, monkeypatch):
    """edgpu_lanczos_gs keeps the Krylov vectors in HBM when they fit and forms the eigenvector as one weighted sum
    (no second Lanczos pass).  Same nlanc, E0 and vector as the two-pass form, also when memory 'runs out' mid-way."""
    from edgpu import model
    for nb, st in ((7, "kron"), (9, "direct"), (11, "csr")):
        m, nu, nd = model.half_filled_c(nb)
        res = {}
        for tag, env in (("stored", {}), ("twopass", {"EDGPU_LZ_STORE": "0"}), ("runs_out", {"EDGPU_LZ_STORE_MAX": "17"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            G = edgpu_mod.HvSector(m, nu, nd, storage=st)
            v0 = np.random.RandomState(3).uniform(0, 1, G.Dim)
            e0, x, nl = G.sp_lanc_eigh(v0, nitermax=300, threshold=1e-13)
            r = np.linalg.norm(G.spHtimesV(x) - e0 * x)
            res[tag] = (e0, x, nl, r)
            G.close()
            for k in env:
                monkeypatch.delenv(k)
        e_ref, x_ref, nl_ref, r_ref = res["twopass"]
        assert nl_ref > 20
        for tag in ("stored", "runs_out"):
            e0, x, nl, r = res[tag]
            assert nl == nl_ref and e0 == e_ref, tag            # the iteration itself is untouched
            assert abs(np.linalg.norm(x) - 1.0) < 1e-12 and r < max(2 * r_ref, 1e-9), (tag, r, r_ref)
            assert np.abs(x - x_ref).max() < 1e-10, tag


def test_fused_iteration_equals_three_pass_iteration(edgpu_mod, monkeypatch):
    """The fused Lanczos iteration (axpy+norm folded into the next local pass, scale and -beta*u_prev into the dw pass)
    against the product + axpy/norm loop it replaces (EDGPU_LZ_FUSED=0) and against the oracle's restatement of
    sp_lanc_tridiag: alpha/beta to 1e-9 relative over 40 steps, E0 to 1e-10, eigenvector to 1e-8."""
    from edgpu import model
    zoo = {z[0]: z[1:] for z in ZOO}
    todo = [model.half_filled_c(nb) + (st,) for nb, st in ((7, "kron"), (9, "direct"), (11, "kron"), (11, "direct"))]
    todo += [zoo[name] + (st,) for name in ("norb2_normal_JxJp", "norb3_nb1_JxJp", "replica_sz_sx_JxJp", "norb2_normal_offdiag_hloc") for st in ("kron", "direct")]      # non-local pass between the two halves; factors with more than 8 hops (k_uppass_ax)
    for m, nu, nd, st in todo:
        nb = m["Nbath"]
        S = oracle.Sector(m, nu, nd)
        v0 = np.random.RandomState(11).uniform(0, 1, S.Dim)
        n = 40 if S.Dim > 2000 else 16      # small sectors: compare before rounding differences get amplified by the loss of orthogonality
        res = {}
        monkeypatch.setenv("EDGPU_KRON_SPLIT", "1")      # the two-kernel product large sectors use (default above 48 MB per vector)
        for tag, env in (("fused", {}), ("fused_twopass", {"EDGPU_LZ_STORE": "0"}), ("fused_dense", {"EDGPU_PAD": "0"}), ("plain_dense", {"EDGPU_LZ_FUSED": "0", "EDGPU_PAD": "0"}),
                         ("plain", {"EDGPU_LZ_FUSED": "0"})):      # dense: work vectors without the padded row pitch (what several ranks use)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            G = edgpu_mod.HvSector(m, nu, nd, storage=st)
            assert G.lanczos_fused == (not tag.startswith("plain")), (nb, st, tag)
            assert ("lanczos=fused" in G.describe()) == (not tag.startswith("plain")) and "product=two-pass" in G.describe() and "storage=" + st in G.describe()
            assert ("pitch=dense" in G.describe()) == (tag.endswith("dense") or G.DimUp % 16 == 0), (tag, G.describe())
            a, b, used = G.sp_lanc_tridiag(v0, n)
            e0, x, nl = G.sp_lanc_eigh(v0, nitermax=300, threshold=1e-13)
            res[tag] = (a, b, used, e0, x, nl, np.linalg.norm(G.spHtimesV(x) - e0 * x))
            G.close()
            for k in env:
                monkeypatch.delenv(k)
        a_ref, b_ref, used_ref = oracle.sp_lanc_tridiag(S.spmatvec, v0, n)
        for tag in ("fused", "fused_twopass", "fused_dense", "plain_dense", "plain"):
            a, b, used, e0, x, nl, r = res[tag]
            assert used == used_ref and b[0] == 0.0
            np.testing.assert_allclose(a[:used], a_ref[:used], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(b[1:used], b_ref[1:used], rtol=1e-9, atol=1e-9)
            assert abs(e0 - res["plain"][3]) <= 1e-10 and abs(nl - res["plain"][5]) <= 1 and r < 1e-6, (tag, e0, nl, r)
            assert abs(np.linalg.norm(x) - 1.0) < 1e-12
            assert min(np.abs(x - res["plain"][4]).max(), np.abs(x + res["plain"][4]).max()) < 1e-8, tag


@pytest.mark.parametrize("nup,ndw", [(3, 6), (10, 6), (9, 5), (2, 7)])
def test_pair_form_with_ten_hop_slots(edgpu_mod, monkeypatch, nup, ndw):
    """Sectors one or two electrons off half filling have 9-10 up hops per configuration (max(Nup, Ns-Nup)): the Green's-function
    target sectors of a half-filled ground state.  Their LDS-row local passes run the pair form on the 8-dword element image
    (k_uppass2 / k_uppass2_ax with 10 hop slots) instead of the one-configuration-per-lane kernels (EDGPU_UP_PAIRS10=0): product
    against the oracle, fused iteration against the three-pass one and against the old form, stored and on the fly."""
    from edgpu import model
    m, _, _ = model.half_filled_c(11)            # Ns = 12
    nu, nd = [nup], [ndw]
    S = oracle.Sector(m, nu, nd)
    assert max(nup, 12 - nup) in (9, 10) and S.DimUp % 2 == 0
    v = np.random.RandomState(3).uniform(-0.5, 0.5, S.Dim)
    ref = S.spmatvec(v)
    v0 = np.random.RandomState(11).uniform(0, 1, S.Dim)
    monkeypatch.setenv("EDGPU_KRON_SPLIT", "1")
    out = {}
    for st in ("kron", "direct"):
        for tag, env in (("pairs10", {}), ("pairs10_plain", {"EDGPU_LZ_FUSED": "0"}), ("lane", {"EDGPU_UP_PAIRS10": "0"})):
            for k, val in env.items():
                monkeypatch.setenv(k, val)
            G = edgpu_mod.HvSector(m, nu, nd, storage=st)
            hv = G.spHtimesV(v)
            assert np.abs(hv - ref).max() <= HXV_RTOL * np.abs(ref).max(), (st, tag)
            a, b, used = G.sp_lanc_tridiag(v0, 30)
            e0, x, nl = G.sp_lanc_eigh(v0, nitermax=300, threshold=1e-13)
            assert np.linalg.norm(S.spmatvec(x) - e0 * x) < 1e-6, (st, tag)
            out[(st, tag)] = (a, b, e0)
            G.close()
            for k in env:
                monkeypatch.delenv(k)
    a_ref, b_ref, used_ref = oracle.sp_lanc_tridiag(S.spmatvec, v0, 30)
    for key, (a, b, e0) in out.items():
        np.testing.assert_allclose(a[:used_ref], a_ref[:used_ref], rtol=1e-9, atol=1e-9, err_msg=str(key))
        np.testing.assert_allclose(b[1:used_ref], b_ref[1:used_ref], rtol=1e-9, atol=1e-9, err_msg=str(key))
        assert abs(e0 - out[("kron", "lane")][2]) <= 1e-10, key


@pytest.mark.parametrize("nbath,nup,ndw", [(6, 3, 4), (10, 3, 5), (10, 8, 6), (14, 7, 8)])
def test_odd_row_length_uses_the_pair_kernels_on_pitched_vectors(edgpu_mod, monkeypatch, nbath, nup, ndw):
    """Every sector of Ns = 7 (the reference's default Nbath = 6) and Ns = 15 has an odd DimUp.  The vectorised dw pass and the fused
    iteration work on column pairs; on the padded row pitch of the Lanczos loops they run over DimUp + 1 columns (the padding
    column holds zeros and keeps them).  Against the oracle (small sectors) and against the form without it (EDGPU_ODD_PAIRS=0:
    scalar dw pass, three-pass iteration), incl. the BASELINE-sized Ns = 15 sector."""
    from edgpu import model
    m, _, _ = model.half_filled_c(nbath)
    nu, nd = [nup], [ndw]
    import math
    DimUp, DimDw = math.comb(nbath + 1, nup), math.comb(nbath + 1, ndw)
    assert DimUp % 2 == 1
    small = DimUp * DimDw < 10 ** 6
    monkeypatch.setenv("EDGPU_KRON_SPLIT", "1")
    if small:
        S = oracle.Sector(m, nu, nd)
        v0 = np.random.RandomState(11).uniform(0, 1, S.Dim)
        a_ref, b_ref, used_ref = oracle.sp_lanc_tridiag(S.spmatvec, v0, 30)
    else:
        v0 = np.random.RandomState(11).uniform(0, 1, DimUp * DimDw)
    out = {}
    for st in ("kron", "direct"):
        for tag, env in (("pairs", {}), ("pairs_plain", {"EDGPU_LZ_FUSED": "0"}), ("scalar", {"EDGPU_ODD_PAIRS": "0"})):
            for k, val in env.items():
                monkeypatch.setenv(k, val)
            G = edgpu_mod.HvSector(m, nu, nd, storage=st)
            assert G.lanczos_fused == (tag == "pairs"), (st, tag, G.describe())
            assert ("pitch=dense" in G.describe()) == (tag == "scalar"), (tag, G.describe())
            a, b, used = G.sp_lanc_tridiag(v0, 30)
            e0, x, nl = G.sp_lanc_eigh(v0, nitermax=300, threshold=1e-13)
            r = np.linalg.norm(G.spHtimesV(x) - e0 * x)
            if small:
                assert np.linalg.norm(S.spmatvec(x) - e0 * x) < 1e-6, (st, tag)
            out[(st, tag)] = (a, b, e0, r)
            G.close()
            for k in env:
                monkeypatch.delenv(k)
    ref = out[("kron", "scalar")]
    for key, (a, b, e0, r) in out.items():
        if small:
            np.testing.assert_allclose(a[:used_ref], a_ref[:used_ref], rtol=1e-9, atol=1e-9, err_msg=str(key))
            np.testing.assert_allclose(b[1:used_ref], b_ref[1:used_ref], rtol=1e-9, atol=1e-9, err_msg=str(key))
        np.testing.assert_allclose(a, ref[0], rtol=1e-9, atol=1e-9, err_msg=str(key))
        np.testing.assert_allclose(b, ref[1], rtol=1e-9, atol=1e-9, err_msg=str(key))
        assert abs(e0 - ref[2]) <= 1e-10 and r < 1e-6, key


def test_sweep_over_all_sectors_of_a_small_model(edgpu_mod):
    """What a DMFT iteration does: every (Nup,Ndw) sector of the Ns=6 model in turn (dimensions 1 ... 400), ground state of
    each against dense diagonalisation.  Exercises the work-vector cache across sectors of changing size."""
    from edgpu import model
    m, _, _ = model.half_filled_c(5)
    Ns = m["Ns"]
    order = [(a, b) for a in range(Ns + 1) for b in range(Ns + 1)]
    rng = np.random.default_rng(11); rng.shuffle(order)          # sizes go up and down
    for nup, ndw in order:
        S = oracle.Sector(m, [nup], [ndw])
        w = np.linalg.eigvalsh(S.hmat())
        G = edgpu_mod.HvSector(m, [nup], [ndw], storage="kron" if (nup + ndw) % 2 else "direct")
        assert G.Dim == S.Dim and G.isector == 1 + nup * (Ns + 1) + ndw
        e0, x, nl = G.sp_lanc_eigh(nitermax=min(300, S.Dim), threshold=1e-14)
        assert abs(e0 - w[0]) < 1e-9, (nup, ndw, e0, w[0])
        assert abs(np.linalg.norm(x) - 1.0) < 1e-10
        if S.Dim > 1:
            deg = np.sum(np.abs(w - w[0]) < 1e-9)
            r = np.linalg.norm(S.spmatvec(x) - e0 * x)
            assert r < 1e-5 or deg > 1, (nup, ndw, r)
        G.close()


def test_c5_full_size_properties(edgpu_mod):
    """BASELINE config 5 at full size (2-orbital impurity, replica bath with H_basis={sigma_z}, Ns=16, N=165,636,900,
    Jx=Jp=Jh and Jx=Jp=0): every format against the oracle's sampled-row evaluator on > 2e4 rows; stored and on-the-fly
    formats agree on the whole vector, H is symmetric, and the non-local pass (k_ndpass) equals the fused row-wise form
    of the same terms."""
    import os
    from edgpu import model
    m = model.make_model(Norb=2, Nbath=7, bath_type="replica", H_basis=[cases.sigma_z()], lambda_imp=[1.0],
                         Uloc=(2.0, 2.0), Jh=0.25, Ust=1.5, Jx=0.25, Jp=0.25)
    N = 165636900
    rng = np.random.default_rng(9)
    u = rng.random(N) - 0.5; v = rng.random(N) - 0.5
    du, dv, dh = edgpu_mod.DeviceVector(N).upload(u), edgpu_mod.DeviceVector(N).upload(v), edgpu_mod.DeviceVector(N)
    # ORACLE parity at full size, with and without the spin-exchange / pair-hopping terms
    idx = _sample_rows(12870, 12870, np.random.default_rng(56), 20000)
    orc = oracle.Sector(m, [8], [8]).rows(idx, v)
    m0 = dict(m, Jx=0.0, Jp=0.0)
    orc0 = oracle.Sector(m0, [8], [8]).rows(idx, v)
    assert np.abs(orc - orc0).max() > 1e-3                 # the non-local terms do act on the sampled rows
    for st in ("kron", "direct"):
        G = edgpu_mod.HvSector(m0, [8], [8], storage=st)
        G.spHtimesV_device(dv, dh); hv = dh.download()
        assert np.abs(hv[idx] - orc0).max() <= HXV_RTOL * np.abs(orc0).max(), (st, "Jx=Jp=0")
        G.close(); del hv
    ref = None
    for st, env in (("kron", {}), ("direct", {}), ("kron", {"EDGPU_ND_SPLIT": "0"}), ("csr", {})):      # csr: stored non-local block (k_csr_nd)
        os.environ.update(env)
        try:
            G = edgpu_mod.HvSector(m, [8], [8], storage=st)
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert G.Dim == N
        G.spHtimesV_device(dv, dh); hv = dh.download()
        assert np.abs(hv[idx] - orc).max() <= HXV_RTOL * np.abs(orc).max(), (st, env, np.abs(hv[idx] - orc).max())
        if ref is None:
            G.spHtimesV_device(du, dh); hu = dh.download()
            scale = float(np.abs(u) @ np.abs(hv))
            assert abs(float(u @ hv) - float(hu @ v)) <= 1e-12 * scale
            ref = hv; del hu
        else:
            assert np.abs(hv - ref).max() <= 1e-12 * np.abs(ref).max(), (st, env)
        G.close()
