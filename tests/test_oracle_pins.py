"""The CPU oracle against the known answers recorded from the reference,
and its internal redundancies: stored == direct == dense."""
import numpy as np
import pytest

from oracle import oracle
import cases

RTOL = 2e-13  # the recorded sums were accumulated in a different order (Fortran sum vs numpy pairwise)


@pytest.mark.parametrize("g", cases.golden_cases(), ids=lambda g: g["name"])
def test_oracle_reproduces_reference_known_answers(g):
    m = cases.model_from_golden(g)
    S = oracle.Sector(m, g["nups"], g["ndws"])
    assert S.isector == g["isector"]
    assert S.Dim == g["N"]
    v = cases.sinvec(S.Dim)
    hv = S.spmatvec(v)
    scale = np.abs(hv).sum()
    if g.get("unpinned"):      # parameters of this record were fitted to its sums (see its _note): not a known answer
        return
    assert abs(hv.sum() - g["sum_Hv"]) <= RTOL * scale
    if "v_Hv" in g:
        assert abs(v @ hv - g["v_Hv"]) <= RTOL * scale
    if "Hv_1_3" in g:
        np.testing.assert_allclose(hv[:3], g["Hv_1_3"], rtol=1e-14, atol=0)
    if "nnz_up" in g:
        assert len(S.sp("up")[1]) == g["nnz_up"]
    if "nnz_nd" in g:
        assert len(S.sp("nd")[1]) == g["nnz_nd"]
    if "up_row1_cols" in g:
        rp, cols, vals = S.sp("up")
        assert list(cols[rp[0]:rp[1]]) == g["up_row1_cols"]
        np.testing.assert_array_equal(vals[rp[0]:rp[1]], np.full(rp[1] - rp[0], g["up_row1_val"]))
    if "E0" in g:
        w = np.linalg.eigvalsh(S.hmat())
        assert abs(w[0] - g["E0"]) < 1e-12
        if "E1" in g:
            assert abs(w[1] - g["E1"]) < 1e-12
    # stored == direct, as measured on the reference itself (<=2.7e-15 at |Hv|~1e3)
    hd = S.directmatvec(v)
    assert np.abs(hv - hd).max() <= 1e-13 * np.abs(hv).max()


@pytest.mark.parametrize("name,m,nu,nd", cases.zoo(), ids=[z[0] for z in cases.zoo()])
def test_oracle_stored_direct_dense_agree(name, m, nu, nd):
    S = oracle.Sector(m, nu, nd)
    rng = np.random.RandomState(7)
    v = rng.uniform(-0.5, 0.5, S.Dim)
    hs, hd = S.spmatvec(v), S.directmatvec(v)
    tol = 1e-13 * max(1.0, np.abs(hs).max())
    assert np.abs(hs - hd).max() <= tol
    if S.Dim <= 3000:
        H = S.hmat()
        assert np.abs(H - H.T).max() == 0.0 or np.abs(H - H.T).max() < 1e-15
        assert np.abs(H @ v - hs).max() <= tol


def test_indexing_primitives():
    """build_sector / c / cdg / binary_search / get_Sector (ED_SETUP.f90:446-457,745-831,1042-1059)."""
    from edgpu import model
    m, nu, nd = model.half_filled_c(5)
    S = oracle.Sector(m, nu, nd)
    mp = S.map(0)
    ref = np.array([x for x in range(2 ** 6) if bin(x).count("1") == 3], np.int32)
    np.testing.assert_array_equal(mp, ref)
    for i, x in enumerate(mp):
        assert oracle.binary_search(mp, int(x)) == i + 1
    assert oracle.binary_search(mp, 1) == 0
    out, sg = oracle.c_op(3, 0b10111)
    assert out == 0b10011 and sg == 1.0     # two occupied levels below position 3
    out, sg = oracle.cdg_op(4, 0b10011)
    assert out == 0b11011 and sg == 1.0
    out, sg = oracle.c_op(2, 0b00011)
    assert out == 0b00001 and sg == -1.0
    assert oracle.get_sector([3, 3], 6) == 25 and oracle.get_sector([4, 4], 8) == 41
    assert model.get_sector([2, 2, 2, 2], 4) == 313 == oracle.get_sector([2, 2, 2, 2], 4)
    assert oracle.binomial(16, 8) == 12870 and oracle.binomial(18, 9) == 48620
