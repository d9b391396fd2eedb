"""Host-side mirror of the reference operator interface over the C ABI.

Names follow the reference (ED_HAMILTONIAN.f90:10-13, ED_VARS_GLOBAL.f90:75-81,148):
``build_Hv_sector`` / ``delete_Hv_sector`` / ``vecDim_Hv_sector`` / ``spHtimesV_p`` and the
SciFortran entry points the reference calls with that operator, ``sp_lanc_eigh`` and
``sp_lanc_tridiag`` (ED_DIAG.f90:177-185, ED_GF_NORMAL.f90:232-237).  Everything computes in
libedgpu.so on the GPU; numpy arrays only carry inputs and results.
"""
import ctypes as C
import sys

import numpy as np

from . import capi
from .capi import check


def _finalizing():
    """True while the interpreter shuts down: device objects still alive then are left to the process exit (the HIP / RCCL
    runtimes of an embedding PyTorch may already be gone, and freeing through them then corrupts their heaps)."""
    return sys is None or sys.is_finalizing()


class DeviceVector:
    """fp64 vector in HBM owned through edgpu_malloc/edgpu_free."""

    def __init__(self, n):
        self.n = int(n)
        p = C.c_void_p()
        check(capi.lib().edgpu_malloc(C.byref(p), self.n * 8))
        self.ptr = p.value

    def upload(self, a):
        a = np.ascontiguousarray(a, np.float64)
        assert a.size == self.n
        check(capi.lib().edgpu_memcpy_h2d(self.ptr, a.ctypes.data, self.n * 8))
        return self

    def zero(self):
        """all-zero vector (the start vector ED_DIAG.f90:147 passes: the library then seeds it itself) without a host copy"""
        check(capi.lib().edgpu_memset(self.ptr, 0, self.n * 8))
        return self

    def download(self):
        out = np.empty(self.n, np.float64)
        check(capi.lib().edgpu_memcpy_d2h(out.ctypes.data, self.ptr, self.n * 8))
        return out

    def free(self):
        if self.ptr:
            capi.lib().edgpu_free(self.ptr)
            self.ptr = None

    def __del__(self):
        if _finalizing():
            return
        try:
            self.free()
        except Exception:
            pass


def _state_ptr(state):
    """(pointer, keep-alive) of a state given as a numpy array (host) or a DeviceVector (already in HBM)"""
    if isinstance(state, DeviceVector):
        return state.ptr, state
    a = np.ascontiguousarray(state, np.float64)
    return a.ctypes.data, a


def _f64(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


class HvSector:
    """One live sector operator: build_Hv_sector(isector) ... delete_Hv_sector()."""

    def __init__(self, model, nups, ndws, ed_sparse_H=True, storage=None, rank=0, size=1, device=-1, force_exchange=False):
        m = model
        if storage is None:
            storage = "kron" if ed_sparse_H else "direct"
        self.storage = storage
        d = capi.Desc()
        d.abi_version = capi.EDGPU_ABI_VERSION
        d.Norb, d.Nspin, d.Nbath, d.Nph = m["Norb"], m["Nspin"], m["Nbath"], m.get("Nph", 0)
        d.bath_type = capi.BATH[m.get("bath_type", "normal")]
        d.ed_total_ud = 1 if m.get("ed_total_ud", True) else 0
        d.storage = capi.STORAGE[storage]
        d.hfmode = 1 if m.get("hfmode", True) else 0
        self._imp, self._dh, self._bd = _f64(m["impHloc"]), _f64(m["diag_hybr"]), _f64(m["bath_diag"])
        self._hb = _f64(m["hbath"]) if m.get("hbath") is not None else None
        d.Nfoo = self._bd.shape[1]
        nups = list(np.atleast_1d(nups)); ndws = list(np.atleast_1d(ndws))
        for i in range(capi.MAX_ORB):
            d.nups[i] = int(nups[i]) if i < len(nups) else 0
            d.ndws[i] = int(ndws[i]) if i < len(ndws) else 0
            d.Uloc[i] = float(m["Uloc"][i]) if i < len(m["Uloc"]) else 0.0
            g = m.get("g_ph", ())
            d.g_ph[i] = float(g[i]) if i < len(g) else 0.0
        d.mpi_rank, d.mpi_size, d.device = rank, size, device
        d.reserved = 1 if force_exchange else 0
        d.Ust, d.Jh, d.Jx, d.Jp = m.get("Ust", 0.0), m.get("Jh", 0.0), m.get("Jx", 0.0), m.get("Jp", 0.0)
        d.xmu, d.w0_ph = m.get("xmu", 0.0), m.get("w0_ph", 0.0)
        d.impHloc, d.diag_hybr, d.bath_diag = self._imp.ctypes.data, self._dh.ctypes.data, self._bd.ctypes.data
        d.hbath = self._hb.ctypes.data if self._hb is not None else None
        self._desc = d
        h = C.c_void_p()
        check(capi.lib().edgpu_sector_create(C.byref(d), C.byref(h)))
        self._h = h
        info = capi.Info()
        check(capi.lib().edgpu_sector_info(self._h, C.byref(info)))
        self.info = info
        self.isector, self.Dim, self.vecDim = info.isector, info.Dim, info.vecDim
        self.DimUp, self.DimDw, self.DimPh, self.Ns_Ud = info.DimUp, info.DimDw, info.DimPh, info.Ns_Ud
        self.mpiQdw, self.mpiIshift = info.mpiQdw, info.mpiIshift
        self._cb = None

    # ---- lifecycle ------------------------------------------------------------------
    def delete_Hv_sector(self):
        if self._h:
            capi.lib().edgpu_sector_destroy(self._h)
            self._h = None

    close = delete_Hv_sector

    def __del__(self):
        if _finalizing():
            return
        try:
            self.delete_Hv_sector()
        except Exception:
            pass

    def vecDim_Hv_sector(self):
        return int(capi.lib().edgpu_sector_vecdim(self._h))

    # ---- inspection -----------------------------------------------------------------
    def map(self, k):
        n = int(self.info.DimUps[k]) if k < self.Ns_Ud else int(self.info.DimDws[k - self.Ns_Ud])
        out = np.zeros(n, np.int32)
        check(capi.lib().edgpu_sector_map(self._h, k, out.ctypes.data))
        return out

    def sp(self, which, iud=0):
        w = {"nd": capi.SP_ND, "up": capi.SP_UP, "dw": capi.SP_DW}[which]
        nrow, nnz = C.c_int64(), C.c_int64()
        check(capi.lib().edgpu_factor_size(self._h, w, iud, C.byref(nrow), C.byref(nnz)))
        rp = np.zeros(nrow.value + 1, np.int64); cols = np.zeros(nnz.value, np.int64); vals = np.zeros(nnz.value, np.float64)
        if nrow.value:
            check(capi.lib().edgpu_factor_export(self._h, w, iud, rp.ctypes.data, cols.ctypes.data, vals.ctypes.data))
        return rp, cols, vals

    def diag(self):
        out = np.zeros(self.vecDim // self.DimPh, np.float64)
        check(capi.lib().edgpu_diag_export(self._h, out.ctypes.data))
        return out

    def hmat(self):
        """Dense Hmat of build_Hv_sector(isector,Hmat) (ED_HAMILTONIAN.f90:128-137), Dim x Dim, Fortran order."""
        H = np.zeros((int(self.Dim), int(self.Dim)), np.float64, order="F")
        check(capi.lib().edgpu_sector_hmat(self._h, H.ctypes.data))
        return H

    def csr_rows(self, row0, nrows, cap=None):
        cap = cap or nrows * 256
        rp = np.zeros(nrows + 1, np.int64); cols = np.zeros(cap, np.int32); vals = np.zeros(cap, np.float64)
        check(capi.lib().edgpu_csr_rows(self._h, row0, nrows, rp.ctypes.data, cols.ctypes.data, vals.ctypes.data, cap))
        n = int(rp[-1] - rp[0])
        return rp, cols[:n], vals[:n]

    # ---- the operator ---------------------------------------------------------------
    def spHtimesV(self, v):
        """dd_sparse_HxV(Nloc,v,Hv) with host arrays (ED_VARS_GLOBAL.f90:75-81)."""
        v = np.ascontiguousarray(v, np.float64)
        hv = np.empty_like(v)
        check(capi.lib().edgpu_hxv64(self._h, v.size, v.ctypes.data, hv.ctypes.data))
        return hv

    def spHtimesV_i32(self, v):
        """Same through the int32 Nloc-by-reference entry the Fortran pointer binds."""
        v = np.ascontiguousarray(v, np.float64)
        hv = np.empty_like(v)
        n = C.c_int32(v.size)
        check(capi.lib().edgpu_hxv(self._h, C.byref(n), v.ctypes.data, hv.ctypes.data))
        return hv

    def spHtimesV_device(self, v_dev, hv_dev):
        check(capi.lib().edgpu_hxv_device(self._h, v_dev.ptr, hv_dev.ptr))

    def hxv_time(self, v_dev, hv_dev, nwarm=3, nrep=20):
        ms = C.c_float()
        check(capi.lib().edgpu_hxv_time(self._h, v_dev.ptr, hv_dev.ptr, nwarm, nrep, C.byref(ms)))
        return ms.value

    # ---- Lanczos --------------------------------------------------------------------
    def sp_lanc_eigh(self, vect=None, nitermax=512, threshold=1e-12, ncheck=10):
        """(egs, vect, nlanc); vect=None/zeros -> library start vector."""
        v = np.zeros(self.vecDim, np.float64) if vect is None else np.array(vect, np.float64)
        e, n = C.c_double(), C.c_int32()
        check(capi.lib().edgpu_lanczos_gs(self._h, nitermax, threshold, ncheck, C.byref(e), v.ctypes.data, C.byref(n)))
        return e.value, v, n.value

    def sp_lanc_eigh_device(self, vect_dev, nitermax=512, threshold=1e-12, ncheck=10):
        e, n = C.c_double(), C.c_int32()
        check(capi.lib().edgpu_lanczos_gs_device(self._h, nitermax, threshold, ncheck, C.byref(e), vect_dev.ptr, C.byref(n)))
        return e.value, n.value

    def sp_eigh(self, neigen, nblock=None, nitermax=512, tol=1e-12, vectors=True, allow_unconverged=False):
        """sp_eigh(MatVec,eval,evec,Nblock,Nitermax,tol): lowest `neigen` pairs (ED_DIAG.f90:153-170).
        Returns (evals, evecs[vecDim, neigen] or None, info dict).  Fewer converged pairs than asked for raises
        (EDGPU_ENOTCONVERGED) unless allow_unconverged: then info["nconv"] tells how many can be trusted."""
        nblock = nblock or min(int(self.Dim), 10 * max(neigen, 2))
        ev = np.zeros(neigen); vec = np.zeros((self.vecDim, neigen), order="F") if vectors else None
        nc, nr, nm = C.c_int32(), C.c_int32(), C.c_int32()
        rc = capi.lib().edgpu_eigsh(self._h, neigen, nblock, nitermax, tol, ev.ctypes.data, vec.ctypes.data if vectors else None,
                                    C.byref(nc), C.byref(nr), C.byref(nm))
        if not (rc == capi.ENOTCONVERGED and allow_unconverged):
            check(rc)
        return ev, vec, {"nconv": nc.value, "restarts": nr.value, "matvecs": nm.value, "ncv_used": int(capi.lib().edgpu_eigsh_ncv_used(self._h))}

    def sp_dvdson_eigh(self, neigen, nitermax=512, tol=1e-12, vectors=True, allow_unconverged=False):
        """sp_dvdson_eigh(MatVec,eval,evec,nitermax,tol): lowest `neigen` pairs by Davidson (ED_DIAG.f90:189-203).
        Raises when fewer pairs converged than asked for, unless allow_unconverged."""
        ev = np.zeros(neigen); vec = np.zeros((self.vecDim, neigen), order="F") if vectors else None
        nc, nm = C.c_int32(), C.c_int32()
        rc = capi.lib().edgpu_dvdson(self._h, neigen, nitermax, tol, ev.ctypes.data, vec.ctypes.data if vectors else None, C.byref(nc), C.byref(nm))
        if not (rc == capi.ENOTCONVERGED and allow_unconverged):
            check(rc)
        return ev, vec, {"nconv": nc.value, "matvecs": nm.value}

    def twin_reorder(self, vec):
        """State of the twin sector (ndw,nup) from a state of this sector (es_return_cvector, itwin branch)."""
        nout = int(self.vecDim)
        if int(self._desc.mpi_size) > 1:   # shard of the twin sector: its dw columns are this sector's up index
            from .model import dw_partition
            qb, _ = dw_partition(int(self.info.DimUp), int(self._desc.mpi_size), int(self._desc.mpi_rank))
            nout = int(self.info.DimDw) * qb * int(self.info.DimPh)
        if isinstance(vec, DeviceVector):          # stays in HBM: returns a DeviceVector
            out = DeviceVector(nout)
            check(capi.lib().edgpu_twin_reorder(self._h, vec.ptr, out.ptr))
            return out
        vec = np.ascontiguousarray(vec, np.float64)
        out = np.empty(nout)
        check(capi.lib().edgpu_twin_reorder(self._h, vec.ctypes.data, out.ctypes.data))
        return out

    def imp_density_offdiag(self, vec):
        """<c^+_a c_b>_s, a != b, shape (2, Norb, Norb) (ED_OBSERVABLES.f90:283-313)."""
        Norb = int(self._desc.Norb)
        vec = np.ascontiguousarray(vec, np.float64)
        dm = np.zeros(2 * Norb * Norb)
        check(capi.lib().edgpu_imp_density_offdiag(self._h, vec.ctypes.data, dm.ctypes.data))
        return dm.reshape(2, Norb, Norb)

    def imp_histogram(self, vec):
        """Impurity-configuration weights (nI, nI, DimPh) of a state (numpy array or DeviceVector)."""
        nI = 1 << int(self._desc.Norb)
        hist = np.zeros(nI * nI * int(self.DimPh))
        if isinstance(vec, DeviceVector):
            check(capi.lib().edgpu_imp_histogram(self._h, vec.ptr, 1, hist.ctypes.data))
        else:
            vec = np.ascontiguousarray(vec, np.float64)
            check(capi.lib().edgpu_imp_histogram(self._h, vec.ctypes.data, 0, hist.ctypes.data))
        return hist.reshape(int(self.DimPh), nI, nI).transpose(1, 2, 0)      # [mu, md, iph]

    def local_energy(self, state, peso=1.0):
        """ed_Epot, ed_Ehartree, ed_Eknot, ed_Dust, ed_Dund, ed_Dse, ed_Dph of lanc_local_energy (ED_OBSERVABLES.f90:372-616)
        for one state (numpy array or DeviceVector) with weight `peso`."""
        sp, _keep = _state_ptr(state)
        out = np.zeros(7)
        check(capi.lib().edgpu_local_energy(self._h, sp, out.ctypes.data))
        return dict(zip(("Epot", "Ehartree", "Eknot", "Dust", "Dund", "Dse", "Dph"), out * peso))

    def observables(self, vec, peso=1.0):
        """dens, dens_up, dens_dw, docc, magz, sz2, n2, s2tot, Prob, prob_ph, dens_ph of
        lanc_observables (ED_OBSERVABLES.f90:150-215) from the impurity-configuration weights."""
        Norb = int(self._desc.Norb)
        Hst = self.imp_histogram(vec) * peso
        nI = 1 << Norb
        o = dict(dens=np.zeros(Norb), dens_up=np.zeros(Norb), dens_dw=np.zeros(Norb), docc=np.zeros(Norb), magz=np.zeros(Norb),
                 sz2=np.zeros((Norb, Norb)), n2=np.zeros((Norb, Norb)), s2tot=0.0, Prob=np.zeros(3 ** Norb),
                 prob_ph=Hst.sum(axis=(0, 1)), dens_ph=float((Hst.sum(axis=(0, 1)) * np.arange(Hst.shape[2])).sum()))
        for mu in range(nI):
            for md in range(nI):
                wgt = Hst[mu, md].sum()
                nup = np.array([(mu >> a) & 1 for a in range(Norb)], float); ndw = np.array([(md >> a) & 1 for a in range(Norb)], float)
                sz, nt = (nup - ndw) / 2.0, nup + ndw
                o["Prob"][int(sum(int(nt[a]) * 3 ** a for a in range(Norb)))] += wgt
                o["dens"] += nt * wgt; o["dens_up"] += nup * wgt; o["dens_dw"] += ndw * wgt
                o["docc"] += nup * ndw * wgt; o["magz"] += (nup - ndw) * wgt
                o["sz2"] += np.outer(sz, sz) * wgt; o["n2"] += np.outer(nt, nt) * wgt
                o["s2tot"] += sz.sum() ** 2 * wgt
        return o

    def chi_lanczos(self, cup, cdw, state, nlanc, threshold=1e-12):
        """Susceptibility leg: (norm2, alanc, blanc, nused) for the diagonal impurity operator
        sum_a cup[a] n_up,a + cdw[a] n_dw,a applied to `state` (ED_GF_CHISPIN/CHIDENS builders)."""
        Norb = int(self._desc.Norb)
        cu = np.zeros(Norb); cu[:len(cup)] = cup; cd = np.zeros(Norb); cd[:len(cdw)] = cdw
        sp, _keep = _state_ptr(state)
        a = np.zeros(nlanc); b = np.zeros(nlanc); n2 = C.c_double(); nu = C.c_int32()
        check(capi.lib().edgpu_chi_lanczos(self._h, cu.ctypes.data, cd.ctypes.data, sp, nlanc, threshold,
                                           C.byref(n2), a.ctypes.data, b.ctypes.data, C.byref(nu)))
        return n2.value, a, b, nu.value

    def phonon_lanczos(self, state, nlanc, threshold=1e-12):
        """Phonon propagator leg: (norm2, alanc, blanc, nused) for (b+b^+)|state> (ED_GF_PHONON.f90)."""
        sp, _keep = _state_ptr(state)
        a = np.zeros(nlanc); b = np.zeros(nlanc); n2 = C.c_double(); nu = C.c_int32()
        check(capi.lib().edgpu_phonon_lanczos(self._h, sp, nlanc, threshold, C.byref(n2), a.ctypes.data, b.ctypes.data, C.byref(nu)))
        return n2.value, a, b, nu.value

    def sp_lanc_tridiag(self, vin, nlanc, threshold=1e-12):
        vin = np.ascontiguousarray(vin, np.float64)
        a = np.zeros(nlanc); b = np.zeros(nlanc); n = C.c_int32()
        check(capi.lib().edgpu_lanczos_tridiag(self._h, vin.ctypes.data, nlanc, threshold, a.ctypes.data, b.ctypes.data, C.byref(n)))
        return a, b, n.value

    def lanczos_time(self, nwarm=2, niter=10):
        ms = C.c_float()
        check(capi.lib().edgpu_lanczos_time(self._h, nwarm, niter, C.byref(ms)))
        return ms.value

    def bench_begin(self, max_steps):
        check(capi.lib().edgpu_lanczos_bench_begin(self._h, max_steps))

    def bench_steps(self, n, timed=False):
        check(capi.lib().edgpu_lanczos_bench_steps(self._h, n, 1 if timed else 0))

    def bench_result(self, ntimed, ncoef=0):
        ms = C.c_float()
        a = np.zeros(max(ncoef, 1)); b = np.zeros(max(ncoef, 1))
        check(capi.lib().edgpu_lanczos_bench_result(self._h, ntimed, C.byref(ms), a.ctypes.data, b.ctypes.data, ncoef))
        return ms.value, a[:ncoef], b[:ncoef]

    # ---- exchange -------------------------------------------------------------------
    def bench_comm_ms(self, ntimed):
        ms = C.c_float()
        check(capi.lib().edgpu_lanczos_bench_comm_ms(self._h, ntimed, C.byref(ms)))
        return ms.value

    def attach_comm(self, comm):
        """Use the process-wide communicator (Communicator) for this sector's exchanges."""
        check(capi.lib().edgpu_sector_attach_comm(self._h, comm._c))
        self._comm = comm          # keep it alive as long as the sector

    @property
    def lanczos_fused(self):
        """True when the Lanczos loops of this sector run the fused two-pass iteration (edgpu_lanczos_is_fused)."""
        return bool(capi.lib().edgpu_lanczos_is_fused(self._h))

    def describe(self):
        """one line: which forms the product and the Lanczos loops of this sector take (edgpu_sector_describe)"""
        buf = C.create_string_buffer(512)
        check(capi.lib().edgpu_sector_describe(self._h, buf, 512))
        return buf.value.decode()

    def exchange_info(self):
        """(mode, send_bytes[size], recv_bytes[size]) of one product: mode 'none' | 'needed-rows' | 'all-gather' | 'transpose'."""
        P = int(self._desc.mpi_size)
        mode = C.c_int32(); sb = np.zeros(P, np.int64); rb = np.zeros(P, np.int64)
        check(capi.lib().edgpu_sector_exchange_info(self._h, C.byref(mode), sb.ctypes.data, rb.ctypes.data))
        return ("none", "needed-rows", "all-gather", "transpose")[mode.value], sb, rb

    def relay_info(self):
        """Routing of the needed-rows exchange: dict(enabled, chunks, busiest_before, busiest_after, rows_direct[P,P], rows_routed[P,P])."""
        P = int(self._desc.mpi_size)
        en = C.c_int32(); nc = C.c_int32(); b0 = C.c_int64(); b1 = C.c_int64()
        rd = np.zeros((P, P), np.int64); rr = np.zeros((P, P), np.int64)
        check(capi.lib().edgpu_sector_relay_info(self._h, C.byref(en), C.byref(nc), C.byref(b0), C.byref(b1), rd.ctypes.data, rr.ctypes.data))
        return dict(enabled=bool(en.value), chunks=nc.value, busiest_before=b0.value, busiest_after=b1.value, rows_direct=rd, rows_routed=rr)

    def comm_init_rccl(self, uid_bytes):
        buf = C.create_string_buffer(bytes(uid_bytes), 128)
        check(capi.lib().edgpu_comm_init_rccl(self._h, buf))

    def comm_init_callbacks(self, allgatherv, exchange, allreduce):
        """allgatherv(send_ptr, nsend, recv_ptr, counts, displs) / exchange(send_ptr, scounts, sdispls,
        recv_ptr, rcounts, rdispls) / allreduce(np_array) in Python."""
        self._cb = _wrap_callbacks(allgatherv, exchange, allreduce)
        check(capi.lib().edgpu_comm_init_callbacks(self._h, self._cb[0], self._cb[1], self._cb[2], None))


def _wrap_callbacks(allgatherv, exchange, allreduce):
    def _ag(ctx, send, nsend, recv, counts, displs, nr):
        try:
            allgatherv(send, nsend, recv, [counts[i] for i in range(nr)], [displs[i] for i in range(nr)])
            return 0
        except Exception as e:  # pragma: no cover
            print("allgatherv callback failed:", e)
            return 1

    def _xc(ctx, send, sc, sd, recv, rc, rd, nr):
        try:
            exchange(send, [sc[i] for i in range(nr)], [sd[i] for i in range(nr)], recv, [rc[i] for i in range(nr)], [rd[i] for i in range(nr)])
            return 0
        except Exception as e:  # pragma: no cover
            print("exchange callback failed:", e)
            return 1

    def _ar(ctx, buf, n):
        try:
            a = np.ctypeslib.as_array(buf, shape=(n,))
            allreduce(a)
            return 0
        except Exception as e:  # pragma: no cover
            print("allreduce callback failed:", e)
            return 1

    return capi.ALLGATHERV_FN(_ag), capi.EXCHANGE_FN(_xc), capi.ALLREDUCE_FN(_ar)


class Communicator:
    """The process-wide communicator (the reference's MpiComm, set once: ED_VARS_GLOBAL.f90:332-347): create it once,
    attach it to every sector (HvSector.attach_comm), close it at the end."""

    def __init__(self, c, rank, size, cb=None, owner=True):
        self._c, self.rank, self.size, self._cb, self._owner = c, rank, size, cb, owner
        self._subs = {}

    @classmethod
    def rccl(cls, uid_bytes, rank, size, device=-1):
        buf = C.create_string_buffer(bytes(uid_bytes), 128)
        c = C.c_void_p()
        check(capi.lib().edgpu_comm_create_rccl(buf, rank, size, device, C.byref(c)))
        return cls(c, rank, size)

    @classmethod
    def callbacks(cls, rank, size, allgatherv, exchange, allreduce):
        cb = _wrap_callbacks(allgatherv, exchange, allreduce)
        c = C.c_void_p()
        check(capi.lib().edgpu_comm_create_callbacks(rank, size, cb[0], cb[1], cb[2], None, C.byref(c)))
        return cls(c, rank, size, cb)

    @classmethod
    def emulated(cls, rank, size):
        """Measurement aid: stands for rank `rank` of `size` on one GPU (edgpu_comm_create_emulated): exchanges become device
        copies of the same byte count, all-reduces return the local value.  Timings are real, results are not."""
        c = C.c_void_p()
        check(capi.lib().edgpu_comm_create_emulated(rank, size, C.byref(c)))
        return cls(c, rank, size)

    def sub(self, nranks):
        """Communicator of the first `nranks` ranks (collective; cached); None on the ranks left out."""
        if nranks == self.size:
            return self
        if nranks not in self._subs:
            c = C.c_void_p(); mem = C.c_int32()
            check(capi.lib().edgpu_comm_sub(self._c, nranks, C.byref(c), C.byref(mem)))
            self._subs[nranks] = Communicator(c, self.rank, nranks, owner=False) if mem.value else None
        return self._subs[nranks]

    def allreduce_sum(self, arr):
        a = np.ascontiguousarray(arr, np.float64)
        check(capi.lib().edgpu_comm_allreduce_sum(self._c, a.ctypes.data, a.size))
        return a

    def close(self):
        if self._c and self._owner:
            capi.lib().edgpu_comm_destroy(self._c)
        self._c = None

    def __del__(self):
        if _finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


def comm_count():
    """native (RCCL) communicators created by this process so far"""
    n = C.c_int32()
    check(capi.lib().edgpu_comm_count(C.byref(n)))
    return n.value


def apply_op(sec_from, sec_to, iorb, ispin, dagger, vin_dev, vout_dev):
    """O|state> in HBM (ED_GF_NORMAL.f90:190-213, 264-283); returns norm2, vout normalised."""
    n2 = C.c_double()
    check(capi.lib().edgpu_apply_op(sec_from._h, sec_to._h, iorb, ispin, 1 if dagger else 0, vin_dev.ptr, vout_dev.ptr, C.byref(n2)))
    return n2.value


def gf_lanczos(sec_from, sec_to, iorb, ispin, dagger, state, nlanc, threshold=1e-12):
    """One leg of lanc_build_gf_normal_main: (norm2, alanc, blanc, nused)."""
    sp, _keep = _state_ptr(state)
    a = np.zeros(nlanc); b = np.zeros(nlanc); n2 = C.c_double(); nu = C.c_int32()
    check(capi.lib().edgpu_gf_lanczos(sec_from._h, sec_to._h, iorb, ispin, 1 if dagger else 0, sp, nlanc, threshold,
                                      C.byref(n2), a.ctypes.data, b.ctypes.data, C.byref(nu)))
    return n2.value, a, b, nu.value


def cache_release():
    """Return the cached work vectors (stored Krylov basis) to the device."""
    check(capi.lib().edgpu_cache_release())


def rccl_unique_id():
    buf = C.create_string_buffer(128)
    check(capi.lib().edgpu_comm_unique_id(buf))
    return buf.raw


def build_Hv_sector(model, nups, ndws, **kw):
    return HvSector(model, nups, ndws, **kw)


def device_count():
    n = C.c_int32()
    check(capi.lib().edgpu_device_count(C.byref(n)))
    return n.value
