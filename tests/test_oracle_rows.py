"""orc_rows -- the oracle's sampled-row evaluator -- against the oracle's own full products.

orc_rows evaluates single rows of H v from the bit patterns (no stored matrix), which is what lets the GPU parity tests
compare against the oracle at the full BASELINE sizes (Ns=16, Ns=18).  Here it is tied to orc_spmatvec, the restatement
of spMatVec_main/orbs (ED_HAMILTONIAN_SPARSE_HxV.f90:391-564): every row, every zoo case, BIT-EXACT (same entries,
same accumulation order)."""
import numpy as np
import pytest

import cases
from oracle import oracle

ZOO = cases.zoo()


@pytest.mark.parametrize("name,m,nu,nd", ZOO, ids=[z[0] for z in ZOO])
def test_rows_equal_spmatvec_bit_exact(name, m, nu, nd):
    S = oracle.Sector(m, nu, nd)
    for v in (cases.sinvec(S.Dim), np.random.RandomState(1).uniform(-0.5, 0.5, S.Dim)):
        ref = S.spmatvec(v)
        np.testing.assert_array_equal(S.rows(np.arange(S.Dim), v), ref)
        perm = np.random.RandomState(2).permutation(S.Dim)[: max(1, S.Dim // 3)]      # any order, any subset
        np.testing.assert_array_equal(S.rows(perm, v), ref[perm])


def test_rows_do_not_assume_a_symmetric_factor():
    """The reference stores (target, source) and never symmetrises; an impHloc with t_ab != t_ba gives a
    non-symmetric H_sigma.  orc_rows inverts each hop statement, so it must still equal the stored product."""
    from edgpu import model
    imp = np.zeros((1, 1, 2, 2)); imp[0, 0, 0, 1] = 0.35; imp[0, 0, 1, 0] = -0.2; imp[0, 0, 0, 0] = 0.1
    m = model.make_model(Norb=2, Nbath=2, Uloc=(2.0, 1.5), Ust=1.2, Jh=0.2, impHloc=imp)
    S = oracle.Sector(m, [3], [2])
    H = S.hmat()
    assert np.abs(H - H.T).max() > 0.1                    # really non-symmetric
    v = np.random.RandomState(3).uniform(-0.5, 0.5, S.Dim)
    ref = S.spmatvec(v)
    np.testing.assert_array_equal(S.rows(np.arange(S.Dim), v), ref)
    assert np.abs(H @ v - ref).max() < 1e-13


def test_rows_on_a_sector_too_large_to_sample_by_hand():
    """Ns=14 half-filled sector (N=11,778,624): 20,000 sampled rows against the full stored product."""
    from edgpu import model
    m, nu, nd = model.half_filled_c(13)
    S = oracle.Sector(m, nu, nd)
    assert S.Dim == 11778624
    v = np.random.RandomState(4).uniform(-0.5, 0.5, S.Dim)
    ref = S.spmatvec(v)
    idx = np.unique(np.concatenate([np.random.RandomState(5).randint(0, S.Dim, 20000), [0, S.DimUp - 1, S.DimUp, S.Dim - 1]]))
    np.testing.assert_array_equal(S.rows(idx, v), ref[idx])


def test_rows_open_the_ns18_sector_without_building_it():
    """Ns=18 (N=2,363,904,400 > 2^31): the sector opens, and rows are evaluated with 64-bit numbers.  The vector here
    is zero except near the sampled rows' columns (a full one is 19 GB): checks the indexing, not the arithmetic."""
    from edgpu import model
    m, nu, nd = model.half_filled_c(17)
    S = oracle.Sector(m, nu, nd)
    assert S.Dim == 2363904400 and S.DimUp == 48620
    # (H e_j)(i) = H[i,j]: a unit vector costs only the pages it touches (numpy's zeros are committed lazily).  j = the
    # last state (row number > 2^31); rows = its whole dw column; then H[j,i] == H[i,j] for the rows i that couple to j
    v = np.zeros(S.Dim)
    j = S.Dim - 1
    v[j] = 1.0
    rows = np.arange(S.Dim - S.DimUp, S.Dim)
    col = S.rows(rows, v)                                           # H[rows, j]
    nz = rows[np.nonzero(col)[0]]
    assert j in nz and 2 <= nz.size <= 1 + 9
    for i in nz[:4]:
        v[j] = 0.0; v[i] = 1.0
        assert S.rows(np.array([j]), v)[0] == col[i - rows[0]]      # H[j,i] == H[i,j]
        v[i] = 0.0; v[j] = 1.0
