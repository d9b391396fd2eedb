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
        self.n_allreduce = 0; self.n_allreduce_doubles = 0      # call counters (tests: collectives per Lanczos iteration)

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
        self.n_allreduce += 1; self.n_allreduce_doubles += len(arr)
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


class _RawCuda:
    """minimal __cuda_array_interface__ view of a raw device pointer (fp64)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class TorchDeviceExchange:
    """The three collectives on DEVICE memory through torch.distributed (backend "nccl" = RCCL):
    the library's raw pointers are viewed as torch tensors, nothing is staged through the host.
    A host that already runs torch.distributed can hand these to HvSector.comm_init_callbacks
    instead of letting libedgpu.so open its own RCCL communicator."""

    def __init__(self, dist, group=None):
        import torch
        self.torch, self.dist, self.group = torch, dist, group

    def _view(self, ptr, n):
        return self.torch.as_tensor(_RawCuda(ptr, n), device="cuda")

    def exchange(self, send_ptr, scounts, sdispls, recv_ptr, rcounts, rdispls):
        rank = self.dist.get_rank(self.group)
        ops = []
        for r, (c, d) in enumerate(zip(scounts, sdispls)):
            if r != rank and c:
                ops.append(self.dist.P2POp(self.dist.isend, self._view(send_ptr + d * 8, c), r, self.group))
        for r, (c, d) in enumerate(zip(rcounts, rdispls)):
            if r != rank and c:
                ops.append(self.dist.P2POp(self.dist.irecv, self._view(recv_ptr + d * 8, c), r, self.group))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        self.torch.cuda.synchronize()

    def allgatherv(self, send_ptr, nsend, recv_ptr, counts, displs):
        rank = self.dist.get_rank(self.group)
        self._view(recv_ptr + displs[rank] * 8, nsend).copy_(self._view(send_ptr, nsend))
        sc = [nsend if r != rank else 0 for r in range(len(counts))]
        rc = [c if r != rank else 0 for r, c in enumerate(counts)]
        self.exchange(send_ptr, sc, [0] * len(counts), recv_ptr, rc, displs)

    def allreduce(self, arr):
        t = self.torch.tensor(arr, dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, group=self.group)
        arr[:] = t.cpu().numpy()
