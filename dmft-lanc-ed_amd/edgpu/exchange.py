"""Host-staged exchange callbacks for HvSector.comm_init_callbacks.

The production exchange is RCCL inside libedgpu.so (edgpu_comm_init_rccl).  This module
provides the same two collectives through torch.distributed with CPU tensors (gloo or MPI
backends), staging through host memory.  It exists so the multi-rank path can be exercised
where RCCL cannot run (several ranks sharing one GPU, or no GPU at all) and so a host that
already owns an MPI communicator can keep using it (MPI_AllGatherv / MPI_AllReduce of the
reference: ED_SETUP.f90:689-733, ED_HAMILTONIAN_SPARSE_HxV.f90:674-692).
"""
import ctypes as C

import numpy as np


def _host_copy(dst, src, nbytes):
    C.memmove(dst, src, nbytes)


class StagedExchange:
    """allgatherv/allreduce over torch.distributed with CPU tensors.

    d2h(host_ptr, dev_ptr, nbytes) / h2d(dev_ptr, host_ptr, nbytes) move bytes between the
    "device" pointers the library hands out and host buffers; the defaults treat the pointers
    as host memory (CPU tests)."""

    def __init__(self, dist, d2h=_host_copy, h2d=_host_copy, group=None):
        self.dist, self.d2h, self.h2d, self.group = dist, d2h, h2d, group

    def allgatherv(self, send_ptr, nsend, recv_ptr, counts, displs):
        import torch
        send = np.empty(nsend, np.float64)
        self.d2h(send.ctypes.data, send_ptr, nsend * 8)
        cmax = max(counts)
        pad = np.zeros(cmax, np.float64); pad[:nsend] = send
        outs = [torch.zeros(cmax, dtype=torch.float64) for _ in counts]
        self.dist.all_gather(outs, torch.from_numpy(pad), group=self.group)
        full = np.empty(sum(counts), np.float64)
        lo = min(displs)
        for r, (c, d) in enumerate(zip(counts, displs)):
            full[d - lo:d - lo + c] = outs[r].numpy()[:c]
        self.h2d(recv_ptr + lo * 8, full.ctypes.data, full.size * 8)

    def exchange(self, send_ptr, scounts, sdispls, recv_ptr, rcounts, rdispls):
        """all-to-all-v of doubles through pairwise isend/irecv of CPU tensors."""
        import torch
        rank = self.dist.get_rank(self.group)
        ops, rbufs = [], {}
        keep = []
        for r, (c, d) in enumerate(zip(scounts, sdispls)):
            if r == rank or c == 0:
                continue
            buf = np.empty(c, np.float64)
            self.d2h(buf.ctypes.data, send_ptr + d * 8, c * 8)
            t = torch.from_numpy(buf); keep.append(t)
            ops.append(self.dist.P2POp(self.dist.isend, t, r, self.group))
        for r, c in enumerate(rcounts):
            if r == rank or c == 0:
                continue
            rbufs[r] = torch.empty(c, dtype=torch.float64)
            ops.append(self.dist.P2POp(self.dist.irecv, rbufs[r], r, self.group))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        for r, t in rbufs.items():
            a = t.numpy()
            self.h2d(recv_ptr + rdispls[r] * 8, a.ctypes.data, a.size * 8)

    def allreduce(self, arr):
        import torch
        t = torch.from_numpy(np.array(arr, np.float64))
        self.dist.all_reduce(t, group=self.group)
        arr[:] = t.numpy()


def device_staged(dist, group=None):
    """StagedExchange whose copies go through edgpu_memcpy_d2h/h2d (real device pointers)."""
    from . import capi
    L = capi.lib()

    def d2h(h, d, n):
        capi.check(L.edgpu_memcpy_d2h(h, d, n))

    def h2d(d, h, n):
        capi.check(L.edgpu_memcpy_h2d(d, h, n))

    return StagedExchange(dist, d2h, h2d, group)
