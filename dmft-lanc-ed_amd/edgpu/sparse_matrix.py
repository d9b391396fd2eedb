"""Host mirror of ED_SPARSE_MATRIX.f90's public interface over the C ABI (sp_init_matrix, sp_insert_element,
sp_delete_matrix, sp_dump_matrix; ED_SPARSE_MATRIX.f90:95-105), with the product computed on the GPU."""
import ctypes as C

import numpy as np

from . import capi
from .capi import check


class sparse_matrix_csr:
    def __init__(self):
        self._h = None
        self.Nrow = self.Ncol = 0
        self.status = False
        self.istart = self.iend = self.ishift = 0
        self.mpi = False


def sp_set_mpi_matrix(sparse, istart, iend, ishift):
    """ED_SPARSE_MATRIX.f90:574-585: the object will hold the global rows istart..iend only (call before sp_init_matrix)."""
    sparse.istart, sparse.iend, sparse.ishift, sparse.mpi = int(istart), int(iend), int(ishift), True


def sp_init_matrix(sparse, N, N1=None):
    if sparse.status:
        raise RuntimeError("sp_init_matrix: already allocated can not init")
    h = C.c_void_p()
    ncol = int(N1 if N1 is not None else N)
    if sparse.mpi:      # mpi_sp_init_matrix: Nloc = iend-istart+1 rows (ED_SPARSE_MATRIX.f90:155-172)
        check(capi.lib().edgpu_sp_init_matrix_mpi(sparse.istart, sparse.iend, sparse.ishift, int(N), ncol, C.byref(h)))
        sparse._h, sparse.Nrow, sparse.Ncol, sparse.status, sparse.Nglob = h, sparse.iend - sparse.istart + 1, ncol, True, int(N)
    else:
        check(capi.lib().edgpu_sp_init_matrix(int(N), ncol, C.byref(h)))
        sparse._h, sparse.Nrow, sparse.Ncol, sparse.status, sparse.Nglob = h, int(N), ncol, True, int(N)


def sp_load_matrix(matrix, sparse):
    """ED_SPARSE_MATRIX.f90:325-344: dense -> sparse (an allocated `sparse` is deleted first)."""
    m = np.asfortranarray(matrix, dtype=np.float64)
    h = sparse._h if sparse.status else C.c_void_p()
    check(capi.lib().edgpu_sp_load_matrix(m.ctypes.data, m.shape[0], m.shape[1], C.byref(h)))
    sparse._h, sparse.Nrow, sparse.Ncol, sparse.status, sparse.Nglob, sparse.mpi = h, m.shape[0], m.shape[1], True, m.shape[0], False


def sp_insert_element(sparse, value, i, j):
    check(capi.lib().edgpu_sp_insert_element(sparse._h, float(value), int(i), int(j)))


def sp_delete_matrix(sparse):
    if sparse.status:
        capi.lib().edgpu_sp_delete_matrix(sparse._h)
    sparse._h, sparse.status, sparse.Nrow, sparse.Ncol = None, False, 0, 0
    sparse.mpi = False


def sp_dump_matrix(sparse, matrix):
    m = np.asfortranarray(matrix, dtype=np.float64)
    assert m.shape == (sparse.Nglob, sparse.Ncol)
    check(capi.lib().edgpu_sp_dump_matrix(sparse._h, m.ctypes.data))
    matrix[...] = m


def sp_row(sparse, i):
    n = C.c_int32()
    check(capi.lib().edgpu_sp_row(sparse._h, int(i), C.byref(n), None, None, 0))
    cols = np.zeros(n.value, np.int64); vals = np.zeros(n.value)
    if n.value:
        check(capi.lib().edgpu_sp_row(sparse._h, int(i), C.byref(n), cols.ctypes.data, vals.ctypes.data, n.value))
    return cols, vals


def sp_matvec_device(sparse, v_dev, hv_dev, accumulate=False, stream=None):
    """Hv (+)= A v on device vectors (edgpu.DeviceVector or raw device pointers): nothing crosses PCIe; asynchronous on
    `stream` (a hipStream_t value, None = default stream) -- synchronise before reading the result."""
    pv = getattr(v_dev, "ptr", v_dev); ph = getattr(hv_dev, "ptr", hv_dev)
    check(capi.lib().edgpu_sp_matvec_device(sparse._h, pv, ph, 1 if accumulate else 0, stream))


def sp_matvec(sparse, v, hv=None):
    v = np.ascontiguousarray(v, np.float64)
    acc = hv is not None
    out = np.ascontiguousarray(hv, np.float64).copy() if acc else np.zeros(sparse.Nrow)
    check(capi.lib().edgpu_sp_matvec(sparse._h, v.ctypes.data, out.ctypes.data, 1 if acc else 0))
    return out
