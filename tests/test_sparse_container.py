"""a1/a2: the sparse-matrix container (ED_SPARSE_MATRIX.f90) -- duplicate-summing insert in storage order,
dense dump, and the product on the GPU -- against a plain Python restatement and against the factors the
oracle builds with the reference's own insertion sequence."""
import numpy as np
import pytest

from edgpu import sparse_matrix as spm
from oracle import oracle
import cases


def _fill_random(rng, n, m, ninsert):
    ref_rows = [([], []) for _ in range(n)]
    ops = []
    for _ in range(ninsert):
        i, j, val = int(rng.integers(1, n + 1)), int(rng.integers(1, m + 1)), float(rng.uniform(-1, 1))
        ops.append((val, i, j))
        c, v = ref_rows[i - 1]
        if j in c:
            v[c.index(j)] += val           # sp_insert_element: column exists -> sum (ED_SPARSE_MATRIX.f90:268-276)
        else:
            c.append(j); v.append(val)     # new column -> append
    return ops, ref_rows


def test_insert_semantics_and_dump_cpu():
    """host-side behaviour needs no GPU: storage order, duplicate summing, dense dump, errors"""
    rng = np.random.default_rng(0)
    A = spm.sparse_matrix_csr(); spm.sp_init_matrix(A, 7, 9)
    ops, ref = _fill_random(rng, 7, 9, 60)
    for val, i, j in ops:
        spm.sp_insert_element(A, val, i, j)
    D = np.zeros((7, 9), order="F")
    spm.sp_dump_matrix(A, D)
    for i in range(7):
        c, v = spm.sp_row(A, i + 1)
        assert list(c) == ref[i][0]
        np.testing.assert_array_equal(v, np.array(ref[i][1]))
        row = np.zeros(9); row[np.array(ref[i][0], int) - 1] = ref[i][1]
        np.testing.assert_array_equal(D[i], row)
    with pytest.raises(RuntimeError):
        spm.sp_init_matrix(A, 3)
    from edgpu import capi
    with pytest.raises(capi.EdgpuError):
        spm.sp_insert_element(A, 1.0, 8, 1)
    spm.sp_delete_matrix(A)
    assert not A.status
    # sp_load_matrix: non-zeros row by row, columns ascending; reloading replaces the content
    M = np.zeros((4, 6)); M[0, 5] = 2.0; M[0, 1] = -1.0; M[3, 0] = 4.0
    B = spm.sparse_matrix_csr(); spm.sp_load_matrix(M, B)
    c, v = spm.sp_row(B, 1); assert list(c) == [2, 6] and list(v) == [-1.0, 2.0]
    D = np.zeros((4, 6), order="F"); spm.sp_dump_matrix(B, D); np.testing.assert_array_equal(D, M)
    spm.sp_load_matrix(2 * M[:, :3], B)
    D = np.zeros((4, 3), order="F"); spm.sp_dump_matrix(B, D); np.testing.assert_array_equal(D, 2 * M[:, :3])
    spm.sp_delete_matrix(B)
    # MPI row window (sp_set_mpi_matrix + sp_init_matrix): global row numbers, local storage, dump at the global position
    W = spm.sparse_matrix_csr(); spm.sp_set_mpi_matrix(W, 3, 5, 2); spm.sp_init_matrix(W, 7, 4)
    assert W.Nrow == 3
    spm.sp_insert_element(W, 1.5, 3, 4); spm.sp_insert_element(W, -2.0, 5, 1); spm.sp_insert_element(W, 0.5, 5, 1)
    with pytest.raises(capi.EdgpuError):
        spm.sp_insert_element(W, 1.0, 2, 1)            # row outside the window
    D = np.zeros((7, 4), order="F"); spm.sp_dump_matrix(W, D)
    E = np.zeros((7, 4)); E[2, 3] = 1.5; E[4, 0] = -1.5
    np.testing.assert_array_equal(D, E)
    c, v = spm.sp_row(W, 5); assert list(c) == [1] and list(v) == [-1.5]
    spm.sp_delete_matrix(W)


@pytest.mark.gpu
def test_container_matvec_on_gpu_and_reference_insertion_sequence():
    rng = np.random.default_rng(1)
    for n, m, k in ((50, 50, 400), (5, 300, 1400), (1, 1, 3)):     # short rows, long rows (wave-per-row kernel), 1x1
        A = spm.sparse_matrix_csr(); spm.sp_init_matrix(A, n, m)
        ops, ref = _fill_random(rng, n, m, k)
        for val, i, j in ops:
            spm.sp_insert_element(A, val, i, j)
        D = np.zeros((n, m), order="F"); spm.sp_dump_matrix(A, D)
        v = rng.uniform(-1, 1, m)
        hv = spm.sp_matvec(A, v)
        np.testing.assert_allclose(hv, D @ v, rtol=0, atol=1e-13 * max(1.0, np.abs(D).sum()))
        hv2 = spm.sp_matvec(A, v, hv)                              # Hv += A v, as the reference's sweeps do
        np.testing.assert_allclose(hv2, 2 * (D @ v), rtol=0, atol=1e-12 * max(1.0, np.abs(D).sum()))
        # device-resident vectors: no PCIe per product
        import edgpu
        dv, dh = edgpu.DeviceVector(m).upload(v), edgpu.DeviceVector(n).upload(np.zeros(n))
        spm.sp_matvec_device(A, dv, dh); spm.sp_matvec_device(A, dv, dh, accumulate=True)
        np.testing.assert_array_equal(dh.download(), hv2)          # same kernels, same order: same bits (download synchronises)
        dv.free(); dh.free()
        spm.sp_delete_matrix(A)
    # row window: the local product returns the window's rows
    W = spm.sparse_matrix_csr(); spm.sp_set_mpi_matrix(W, 11, 30, 10); spm.sp_init_matrix(W, 50)
    Dfull = np.zeros((50, 50))
    for _ in range(200):
        i, j, val = int(rng.integers(11, 31)), int(rng.integers(1, 51)), float(rng.uniform(-1, 1))
        spm.sp_insert_element(W, val, i, j); Dfull[i - 1, j - 1] += val
    v = rng.uniform(-1, 1, 50)
    np.testing.assert_allclose(spm.sp_matvec(W, v), (Dfull @ v)[10:30], rtol=0, atol=1e-12)
    spm.sp_delete_matrix(W)
    # replay the reference's own insertion sequence for H_up of config 1 (column sweep, stored/H_up.f90) and compare
    name, m_, nu, nd = cases.zoo()[3]      # c_nb5_half
    S = oracle.Sector(m_, nu, nd).buildh()
    rp, cols, vals = S.sp("up")
    A = spm.sparse_matrix_csr(); spm.sp_init_matrix(A, S.DimUp)
    entries = [(int(cols[p]), i + 1, float(vals[p])) for i in range(S.DimUp) for p in range(rp[i], rp[i + 1])]
    for j, i, val in sorted(entries):      # the reference loops over the column jup outermost
        spm.sp_insert_element(A, val, i, j)
    for i in range(S.DimUp):
        c, v = spm.sp_row(A, i + 1)
        np.testing.assert_array_equal(c, cols[rp[i]:rp[i + 1]]); np.testing.assert_array_equal(v, vals[rp[i]:rp[i + 1]])
    spm.sp_delete_matrix(A)
