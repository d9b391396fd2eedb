"""ctypes front-end of the CPU oracle (oracle/ed_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may
import this module.  The product package never does.

The Lanczos routines here restate SciFortran's ``sp_lanc_eigh`` /
``sp_lanc_tridiag`` (github aamaricci/SciFortran, module SF_SP_LINALG; version
un-pinned by the reference, CMakeLists.txt:92, and absent from
/root/reference), anchored on the reference call sites ED_DIAG.f90:177-185
and ED_GF_NORMAL.f90:232-237,638-640.  Iteration counts are NOT a parity
quantity (the library's random start vector is compiler specific); energies
and Lanczos coefficients for a *given* start vector are.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "ed_oracle.c")
_LIB = os.path.join(_HERE, "_build", "libed_oracle.so")

BATH_TYPES = {"normal": 0, "hybrid": 1, "replica": 2}


def build(force=False):
    """gcc the C restatement into oracle/_build/libed_oracle.so."""
    if (not force) and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= os.path.getmtime(_SRC):
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=c99", "-o", _LIB, _SRC, "-lm"])
    return _LIB


def set_threads(n):
    """OpenMP row split of the stored product (bench.py's CPU baseline; results are bit-identical to 1 thread)."""
    lib().orc_set_threads(int(n))


class _Params(C.Structure):
    _fields_ = [
        ("Norb", C.c_int), ("Nspin", C.c_int), ("Nbath", C.c_int), ("Nph", C.c_int),
        ("bath_type", C.c_int), ("ed_total_ud", C.c_int), ("hfmode", C.c_int), ("Nfoo", C.c_int),
        ("Uloc", C.c_double * 5), ("Ust", C.c_double), ("Jh", C.c_double), ("Jx", C.c_double),
        ("Jp", C.c_double), ("xmu", C.c_double), ("w0_ph", C.c_double), ("g_ph", C.c_double * 5),
        ("impHloc", C.c_void_p), ("diag_hybr", C.c_void_p), ("bath_diag", C.c_void_p), ("hbath", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_sector_open.restype = C.c_void_p
        L.orc_sector_open.argtypes = [C.POINTER(_Params), C.c_int]
        L.orc_sector_close.argtypes = [C.c_void_p]
        L.orc_buildh.argtypes = [C.c_void_p]
        L.orc_sp_nnz.restype = C.c_int64
        L.orc_sp_nnz.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.orc_sp_export.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_spmatvec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_directmatvec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hmat.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sector_map.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_sector_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_binomial.restype = C.c_int
        L.orc_get_sector.restype = C.c_int
        L.orc_get_sector.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_binary_search.restype = C.c_int
        L.orc_binary_search.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_c.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_cdg.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_set_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def get_sector(QN, N):
    """ED_SETUP.f90:446-457"""
    q = np.ascontiguousarray(QN, dtype=np.int32)
    return int(lib().orc_get_sector(q.ctypes.data, len(q), int(N)))


def binomial(n, k):
    return int(lib().orc_binomial(int(n), int(k)))


def c_op(pos, state):
    out, sg = C.c_int(0), C.c_double(0)
    lib().orc_c(pos, state, C.byref(out), C.byref(sg))
    return out.value, sg.value


def cdg_op(pos, state):
    out, sg = C.c_int(0), C.c_double(0)
    lib().orc_cdg(pos, state, C.byref(out), C.byref(sg))
    return out.value, sg.value


def binary_search(a, value):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return int(lib().orc_binary_search(a.ctypes.data, len(a), int(value)))


WHICH = {"d": 0, "nd": 1, "up": 2, "dw": 3, "ph": 4, "e_eph": 5, "ph_eph": 6}


class Sector:
    """One (Nup,Ndw) sector of the reference: build_Hv_sector .. delete_Hv_sector
    (ED_HAMILTONIAN.f90:43-222) in serial mode."""

    def __init__(self, model, nups, ndws):
        m = model
        self.model = m
        f64 = lambda a: np.asfortranarray(np.asarray(a, dtype=np.float64))
        self._imp = f64(m["impHloc"]); self._dh = f64(m["diag_hybr"]); self._bd = f64(m["bath_diag"])
        self._hb = f64(m["hbath"]) if m.get("hbath") is not None else None
        P = _Params()
        P.Norb, P.Nspin, P.Nbath, P.Nph = m["Norb"], m["Nspin"], m["Nbath"], m.get("Nph", 0)
        P.bath_type = BATH_TYPES[m.get("bath_type", "normal")]
        P.ed_total_ud = 1 if m.get("ed_total_ud", True) else 0
        P.hfmode = 1 if m.get("hfmode", True) else 0
        P.Nfoo = self._bd.shape[1]
        for i in range(5):
            P.Uloc[i] = float(m["Uloc"][i]) if i < len(m["Uloc"]) else 0.0
            P.g_ph[i] = float(m.get("g_ph", [0] * 5)[i]) if i < len(m.get("g_ph", [0] * 5)) else 0.0
        P.Ust, P.Jh, P.Jx, P.Jp = m.get("Ust", 0.0), m.get("Jh", 0.0), m.get("Jx", 0.0), m.get("Jp", 0.0)
        P.xmu, P.w0_ph = m.get("xmu", 0.0), m.get("w0_ph", 0.0)
        assert self._imp.shape == (P.Nspin, P.Nspin, P.Norb, P.Norb)
        assert self._dh.shape == (P.Nspin, P.Norb, P.Nbath)
        assert self._bd.shape[0] == P.Nspin and self._bd.shape[2] == P.Nbath
        P.impHloc, P.diag_hybr, P.bath_diag = self._imp.ctypes.data, self._dh.ctypes.data, self._bd.ctypes.data
        if P.bath_type == 2:
            assert self._hb is not None and self._hb.shape == (P.Nspin, P.Nspin, P.Norb, P.Norb, P.Nbath)
        P.hbath = self._hb.ctypes.data if self._hb is not None else None
        self._P = P
        nups = list(np.atleast_1d(nups)); ndws = list(np.atleast_1d(ndws))
        if P.ed_total_ud:
            ns_orb = (P.Nbath + P.Norb) if P.bath_type == 1 else (P.Nbath + 1) * P.Norb
        else:
            ns_orb = P.Nbath + 1
        self.isector = get_sector(nups + ndws, ns_orb)
        self._h = lib().orc_sector_open(C.byref(P), self.isector)
        v = [C.c_int() for _ in range(5)]
        dups = np.zeros(5, np.int32); ddws = np.zeros(5, np.int32); dim = C.c_int64()
        lib().orc_sector_info(self._h, *[C.byref(x) for x in v], dups.ctypes.data, ddws.ctypes.data, C.byref(dim))
        self.Ns, self.Ns_Orb, self.Ns_Ud, self.DimPh, self.Nsectors = [x.value for x in v]
        self.DimUps = dups[: self.Ns_Ud].copy(); self.DimDws = ddws[: self.Ns_Ud].copy()
        self.DimUp = int(np.prod(self.DimUps, dtype=np.int64)); self.DimDw = int(np.prod(self.DimDws, dtype=np.int64))
        self.Dim = dim.value
        self._built = False

    def close(self):
        if self._h:
            lib().orc_sector_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def map(self, k):
        """Hs(k+1)%map (0-based k over [ups..., dws...])."""
        n = int(self.DimUps[k]) if k < self.Ns_Ud else int(self.DimDws[k - self.Ns_Ud])
        out = np.zeros(n, np.int32)
        lib().orc_sector_map(self._h, k, out.ctypes.data)
        return out

    def buildh(self):
        lib().orc_buildh(self._h)
        self._built = True
        return self

    def sp(self, which, iud=0):
        """(rowptr, cols[1-based], vals) of spH0d / spH0nd / spH0ups(iud+1) / ..."""
        if not self._built:
            self.buildh()
        nrow = C.c_int64()
        nnz = lib().orc_sp_nnz(self._h, WHICH[which], iud, C.byref(nrow))
        rp = np.zeros(nrow.value + 1, np.int64); cols = np.zeros(nnz, np.int64); vals = np.zeros(nnz, np.float64)
        if nrow.value:
            lib().orc_sp_export(self._h, WHICH[which], iud, rp.ctypes.data, cols.ctypes.data, vals.ctypes.data)
        return rp, cols, vals

    def spmatvec(self, v):
        """spMatVec_main / spMatVec_orbs (ED_HAMILTONIAN_SPARSE_HxV.f90:391-564)."""
        if not self._built:
            self.buildh()
        v = np.ascontiguousarray(v, np.float64); assert v.shape == (self.Dim,)
        hv = np.empty_like(v)
        lib().orc_spmatvec(self._h, v.ctypes.data, hv.ctypes.data)
        return hv

    def directmatvec(self, v):
        """directMatVec_main / _orbs (ED_HAMILTONIAN_DIRECT_HxV.f90:21-171)."""
        v = np.ascontiguousarray(v, np.float64); assert v.shape == (self.Dim,)
        hv = np.empty_like(v)
        lib().orc_directmatvec(self._h, v.ctypes.data, hv.ctypes.data)
        return hv

    def rows(self, idx, v):
        """(H v)[idx] for 0-based global row numbers `idx`, evaluated row by row from the bit patterns (orc_rows):
        needs no stored matrix, so it also runs on the Ns=16/18 sectors.  Same per-row accumulation order as
        spmatvec()."""
        v = np.ascontiguousarray(v, np.float64); assert v.shape == (self.Dim,)
        ix = np.ascontiguousarray(np.asarray(idx, dtype=np.int64) + 1)
        out = np.empty(ix.size, np.float64)
        lib().orc_rows(self._h, ix.size, ix.ctypes.data, v.ctypes.data, out.ctypes.data)
        return out

    def hmat(self):
        """dense Hmat of build_Hv_sector(isector,Hmat)."""
        if not self._built:
            self.buildh()
        assert self.Dim <= 20000
        H = np.zeros((self.Dim, self.Dim), np.float64, order="F")
        lib().orc_hmat(self._h, H.ctypes.data)
        return H


# --------------------------------------------------------------------------
# SciFortran SF_SP_LINALG restatement (see module docstring)
# --------------------------------------------------------------------------
def _lanczos_iteration(matvec, it, vin, vout, beta_prev):
    """lanczos_iteration_d: iter 1 normalises vin; else rotates (vin,vout)."""
    if it == 1:
        nrm = np.sqrt(np.dot(vin, vin))
        if nrm == 0.0:
            raise ValueError("LANCZOS_ITERATION: norm =0!!")
        vin = vin / nrm
    else:
        tmp = vin
        vin = vout / beta_prev
        vout = -beta_prev * tmp
    tmp = matvec(vin)
    vout = vout + tmp
    alfa = float(np.dot(vin, vout))
    vout = vout - alfa * vin
    beta = float(np.sqrt(np.dot(vout, vout)))
    return vin, vout, alfa, beta


def sp_lanc_tridiag(matvec, vin, nlanc, threshold=1e-12):
    """alanc(1:n), blanc(1:n) with blanc(1)=0 unused (ED_GF_NORMAL.f90:638-640)."""
    from numpy import zeros
    alanc = zeros(nlanc); blanc = zeros(nlanc)
    vin = np.array(vin, np.float64); vout = np.zeros_like(vin); b = 0.0
    nused = 0
    for it in range(1, nlanc + 1):
        vin, vout, a, b = _lanczos_iteration(matvec, it, vin, vout, b)
        alanc[it - 1] = a; nused = it
        if abs(b) < threshold:
            break
        if it < nlanc:
            blanc[it] = b
    return alanc, blanc, nused


def sp_lanc_eigh(matvec, vect, nitermax, threshold=1e-12, ncheck=10):
    """Plain two-pass Lanczos ground state.  Returns (egs, vect, nlanc)."""
    from scipy.linalg import eigh_tridiagonal
    vect = np.array(vect, np.float64)
    if np.dot(vect, vect) == 0.0:
        raise ValueError("oracle needs an explicit start vector (the library's RNG is compiler specific)")
    vin = vect.copy(); vout = np.zeros_like(vin)
    alanc = []; blanc = [0.0]; b = 0.0; esave = 0.0; nlanc = 0
    for it in range(1, nitermax + 1):
        vin, vout, a, b = _lanczos_iteration(matvec, it, vin, vout, b)
        if abs(b) < threshold:
            alanc.append(a); nlanc += 1
            break
        nlanc += 1
        alanc.append(a); blanc.append(b)
        w = eigh_tridiagonal(np.array(alanc), np.array(blanc[1:nlanc]), eigvals_only=True) if nlanc > 1 else np.array(alanc)
        if nlanc >= ncheck:
            diff = abs(w[0] - esave)
            if diff <= threshold:
                esave = w[0]
                break
        esave = w[0]
    al = np.array(alanc[:nlanc]); bl = np.array(blanc[1:nlanc])
    if nlanc > 1:
        w, Z = eigh_tridiagonal(al, bl)
    else:
        w, Z = al.copy(), np.ones((1, 1))
    egs = float(w[0])
    vin = vect.copy(); vout = np.zeros_like(vin); out = np.zeros_like(vin); b = 0.0
    for it in range(1, nlanc + 1):
        vin, vout, a, b = _lanczos_iteration(matvec, it, vin, vout, b)
        out = out + vin * Z[it - 1, 0]
    out = out / np.sqrt(np.dot(out, out))
    return egs, out, nlanc


# --------------------------------------------------------------------------
# Green's-function leg (ED_GF_NORMAL.f90:124-334), small sectors only
# --------------------------------------------------------------------------
def apply_op(S_from, S_to, iorb, ispin, dagger, state):
    """vvinit of lanc_build_gf_normal_main: loops over the SOURCE states i and scatters
    vvinit(j) = sgn*state(i) (ED_GF_NORMAL.f90:190-213 for c^+, 264-283 for c).
    Returns (vvinit normalised, norm2)."""
    Ns_Ud, Ns_Orb = S_from.Ns_Ud, S_from.Ns_Orb
    total_ud = S_from.model.get("ed_total_ud", True)
    ialfa, iorb1 = (1, iorb) if total_ud else (iorb, 1)
    ibeta = ialfa + (ispin - 1) * Ns_Ud
    idims = [int(x) for x in list(S_from.DimUps) + list(S_from.DimDws)]
    jdims = [int(x) for x in list(S_to.DimUps) + list(S_to.DimDws)]
    HI = [S_from.map(k) for k in range(2 * Ns_Ud)]
    HJ = [S_to.map(k) for k in range(2 * Ns_Ud)]
    iel, jel = S_from.DimUp * S_from.DimDw, S_to.DimUp * S_to.DimDw
    vv = np.zeros(S_to.Dim)
    for i in range(1, S_from.Dim + 1):
        iph = (i - 1) // iel + 1
        i_el = (i - 1) % iel + 1
        c = i_el - 1; ind = []
        for d in idims:                       # state2indices
            ind.append(c % d + 1); c //= d
        m = int(HI[ibeta - 1][ind[ibeta - 1] - 1])
        occ = (m >> (iorb1 - 1)) & 1
        if dagger:
            if occ != 0:
                continue
            r, sgn = cdg_op(iorb1, m)
        else:
            if occ != 1:
                continue
            r, sgn = c_op(iorb1, m)
        jnd = list(ind)
        jnd[ibeta - 1] = binary_search(HJ[ibeta - 1], r)
        j = jnd[0]; prod = 1                  # indices2state
        for k in range(1, len(jnd)):
            prod *= jdims[k - 1]; j += (jnd[k] - 1) * prod
        j += (iph - 1) * jel
        vv[j - 1] = sgn * state[i - 1]
    norm2 = float(vv @ vv)
    return (vv / np.sqrt(norm2) if norm2 > 0 else vv), norm2


def apply_op_vec(S_from, S_to, iorb, ispin, dagger, state, normalise=True):
    """The same statements as apply_op (ED_GF_NORMAL.f90:190-213, 264-283) evaluated for whole index slices with numpy,
    so that the Ns=16 target sectors are within reach: the operator acts on ONE of the sector's index factors
    (state2indices), every other index is carried along.  Bit-equal to apply_op (tests/test_analytic_cpu.py)."""
    Ns_Ud = S_from.Ns_Ud
    total_ud = S_from.model.get("ed_total_ud", True)
    ialfa, iorb1 = (1, iorb) if total_ud else (iorb, 1)
    ibeta = ialfa + (ispin - 1) * Ns_Ud
    idims = [int(x) for x in list(S_from.DimUps) + list(S_from.DimDws)]
    jdims = [int(x) for x in list(S_to.DimUps) + list(S_to.DimDws)]
    HI = S_from.map(ibeta - 1).astype(np.int64); HJ = S_to.map(ibeta - 1).astype(np.int64)
    bit = 1 << (iorb1 - 1)
    occ = (HI & bit) != 0
    src = np.nonzero(~occ if dagger else occ)[0]
    r = HI[src] ^ bit                                  # cdg sets / c clears bit pos-1 (ED_SETUP.f90:805-831)
    low = HI[src] & (bit - 1)
    par = np.zeros(src.size, np.int64)
    for q in range(iorb1 - 1):
        par += (low >> q) & 1
    sgn = np.where(par % 2 == 0, 1.0, -1.0)
    dst = np.searchsorted(HJ, r)
    assert np.array_equal(HJ[dst], r)
    ax = len(idims) - ibeta + 1                        # C-order axis of index factor ibeta behind the phonon axis
    vin = np.asarray(state, np.float64).reshape([S_from.DimPh] + idims[::-1])
    vv = np.zeros([S_to.DimPh] + jdims[::-1])
    a, b = np.moveaxis(vv, ax, -1), np.moveaxis(vin, ax, -1)
    a[..., dst] = b[..., src] * sgn
    vv = vv.reshape(-1)
    if not normalise:
        return vv
    norm2 = float(vv @ vv)
    return (vv / np.sqrt(norm2) if norm2 > 0 else vv), norm2


def lanc_observables(S, state, peso=1.0):
    """Restatement of the per-state loop of lanc_observables (ED_OBSERVABLES.f90:150-215): loops over
    all basis states of sector S and accumulates the local observables with weight peso*|state_i|^2."""
    Norb = S.model["Norb"]; Ns_Ud, Ns_Orb = S.Ns_Ud, S.Ns_Orb
    total_ud = S.model.get("ed_total_ud", True)
    dims = [int(x) for x in list(S.DimUps) + list(S.DimDws)]
    HI = [S.map(k) for k in range(2 * Ns_Ud)]
    iel = S.DimUp * S.DimDw
    o = dict(dens=np.zeros(Norb), dens_up=np.zeros(Norb), dens_dw=np.zeros(Norb), docc=np.zeros(Norb), magz=np.zeros(Norb),
             sz2=np.zeros((Norb, Norb)), n2=np.zeros((Norb, Norb)), s2tot=0.0, Prob=np.zeros(3 ** Norb),
             prob_ph=np.zeros(S.DimPh), dens_ph=0.0)
    for i in range(1, S.Dim + 1):
        iph = (i - 1) // iel + 1
        c = (i - 1) % iel; ind = []
        for d in dims:
            ind.append(c % d + 1); c //= d
        if total_ud:       # breorder: levels 1..Norb of the single up/dw word are the impurity orbitals
            mup, mdw = int(HI[0][ind[0] - 1]), int(HI[1][ind[1] - 1])
            nup = np.array([(mup >> a) & 1 for a in range(Norb)], float); ndw = np.array([(mdw >> a) & 1 for a in range(Norb)], float)
        else:              # one (1+Nbath)-level word per orbital, level 1 = impurity
            nup = np.array([int(HI[a][ind[a] - 1]) & 1 for a in range(Norb)], float)
            ndw = np.array([int(HI[a + Ns_Ud][ind[a + Ns_Ud] - 1]) & 1 for a in range(Norb)], float)
        w = peso * abs(state[i - 1]) ** 2
        sz, nt = (nup - ndw) / 2.0, nup + ndw
        iprob = 1 + sum(int(round(nt[a])) * 3 ** a for a in range(Norb))
        o["Prob"][iprob - 1] += w
        for a in range(Norb):
            o["dens"][a] += nt[a] * w; o["dens_up"][a] += nup[a] * w; o["dens_dw"][a] += ndw[a] * w
            o["docc"][a] += nup[a] * ndw[a] * w; o["magz"][a] += (nup[a] - ndw[a]) * w
            o["sz2"][a, a] += sz[a] * sz[a] * w; o["n2"][a, a] += nt[a] * nt[a] * w
            for b in range(a + 1, Norb):
                o["sz2"][a, b] += sz[a] * sz[b] * w; o["sz2"][b, a] += sz[b] * sz[a] * w
                o["n2"][a, b] += nt[a] * nt[b] * w; o["n2"][b, a] += nt[b] * nt[a] * w
        o["s2tot"] += sz.sum() ** 2 * w
        o["prob_ph"][iph - 1] += w
        o["dens_ph"] += (iph - 1) * w
    return o


def local_energy(S, state, peso=1.0):
    """Restatement of the per-state loop of lanc_local_energy (ED_OBSERVABLES.f90:372-616): loops over the basis states
    of sector S, accumulates ed_Eknot (diagonal + off-diagonal impHloc), ed_Epot (density terms + Jx/Jp), ed_Dust,
    ed_Dund, ed_Dse, ed_Dph and ed_Ehartree with the reference's conditions and c/cdg calls; ed_Epot includes the Hartree
    shift as line 589 leaves it.  Small sectors only."""
    m = S.model
    Norb, Nspin = m["Norb"], m["Nspin"]
    total_ud = m.get("ed_total_ud", True)
    Ns_Ud = S.Ns_Ud
    imp = np.asarray(m["impHloc"], float)
    Uloc = [float(x) for x in m["Uloc"]]; Ust, Jh, Jx, Jp = m.get("Ust", 0.0), m.get("Jh", 0.0), m.get("Jx", 0.0), m.get("Jp", 0.0)
    hfmode = m.get("hfmode", True)
    Jhflag = Norb > 1 and (Jx != 0.0 or Jp != 0.0)
    dims = [int(x) for x in list(S.DimUps) + list(S.DimDws)]
    H = [S.map(k) for k in range(2 * Ns_Ud)]
    iel = S.DimUp * S.DimDw
    o = dict(Epot=0.0, Ehartree=0.0, Eknot=0.0, Dust=0.0, Dund=0.0, Dse=0.0, Dph=0.0)
    for i in range(1, S.Dim + 1):
        iph = (i - 1) // iel + 1
        c = (i - 1) % iel; ind = []
        for d in dims:
            ind.append(c % d + 1); c //= d
        if total_ud:
            mup, mdw = int(H[0][ind[0] - 1]), int(H[1][ind[1] - 1])
            nup = [(mup >> a) & 1 for a in range(Norb)]; ndw = [(mdw >> a) & 1 for a in range(Norb)]
        else:
            nup = [int(H[a][ind[a] - 1]) & 1 for a in range(Norb)]
            ndw = [int(H[a + Ns_Ud][ind[a + Ns_Ud] - 1]) & 1 for a in range(Norb)]
        gw = peso * abs(state[i - 1]) ** 2
        for a in range(Norb):
            o["Eknot"] += imp[0, 0, a, a] * nup[a] * gw
            o["Eknot"] += imp[Nspin - 1, Nspin - 1, a, a] * ndw[a] * gw
        if total_ud:
            iup, idw = ind[0], ind[1]
            for a in range(1, Norb + 1):
                for b in range(1, Norb + 1):
                    if imp[0, 0, a - 1, b - 1] != 0.0 and nup[b - 1] == 1 and nup[a - 1] == 0:
                        k1, sg1 = c_op(b, mup); k2, sg2 = cdg_op(a, k1)
                        j = binary_search(H[0], k2) + (idw - 1) * S.DimUp + (iph - 1) * iel
                        o["Eknot"] += imp[0, 0, a - 1, b - 1] * sg1 * sg2 * state[i - 1] * state[j - 1] * peso
                    if imp[Nspin - 1, Nspin - 1, a - 1, b - 1] != 0.0 and ndw[b - 1] == 1 and ndw[a - 1] == 0:
                        k1, sg1 = c_op(b, mdw); k2, sg2 = cdg_op(a, k1)
                        j = iup + (binary_search(H[1], k2) - 1) * S.DimUp + (iph - 1) * iel
                        o["Eknot"] += imp[Nspin - 1, Nspin - 1, a - 1, b - 1] * sg1 * sg2 * state[i - 1] * state[j - 1] * peso
            if Jhflag and Jx != 0.0:
                for a in range(1, Norb + 1):
                    for b in range(1, Norb + 1):
                        if a != b and nup[b - 1] == 1 and ndw[a - 1] == 1 and ndw[b - 1] == 0 and nup[a - 1] == 0:
                            k1, sg1 = c_op(a, mdw); k2, sg2 = cdg_op(b, k1)
                            k3, sg3 = c_op(b, mup); k4, sg4 = cdg_op(a, k3)
                            j = binary_search(H[0], k4) + (binary_search(H[1], k2) - 1) * S.DimUp + (iph - 1) * iel
                            t = sg1 * sg2 * sg3 * sg4 * state[i - 1] * state[j - 1] * peso
                            o["Epot"] += Jx * t; o["Dse"] += t
            if Jhflag and Jp != 0.0:
                for a in range(1, Norb + 1):
                    for b in range(1, Norb + 1):
                        if nup[b - 1] == 1 and ndw[b - 1] == 1 and ndw[a - 1] == 0 and nup[a - 1] == 0:
                            k1, sg1 = c_op(b, mdw); k2, sg2 = cdg_op(a, k1)
                            k3, sg3 = c_op(b, mup); k4, sg4 = cdg_op(a, k3)
                            j = binary_search(H[0], k4) + (binary_search(H[1], k2) - 1) * S.DimUp + (iph - 1) * iel
                            t = sg1 * sg2 * sg3 * sg4 * state[i - 1] * state[j - 1] * peso
                            o["Epot"] += Jp * t; o["Dph"] += t
        for a in range(Norb):
            o["Epot"] += Uloc[a] * nup[a] * ndw[a] * gw
        if Norb > 1:
            for a in range(Norb):
                for b in range(a + 1, Norb):
                    o["Epot"] += Ust * (nup[a] * ndw[b] + nup[b] * ndw[a]) * gw
                    o["Dust"] += (nup[a] * ndw[b] + nup[b] * ndw[a]) * gw
            for a in range(Norb):
                for b in range(a + 1, Norb):
                    o["Epot"] += (Ust - Jh) * (nup[a] * nup[b] + ndw[a] * ndw[b]) * gw
                    o["Dund"] += (nup[a] * nup[b] + ndw[a] * ndw[b]) * gw
        if hfmode:
            for a in range(Norb):
                o["Ehartree"] += -0.5 * Uloc[a] * (nup[a] + ndw[a]) * gw + 0.25 * Uloc[a] * gw
            if Norb > 1:
                for a in range(Norb):
                    for b in range(a + 1, Norb):
                        n4 = nup[a] + ndw[a] + nup[b] + ndw[b]
                        o["Ehartree"] += -0.5 * Ust * n4 * gw + 0.25 * Ust * gw
                        o["Ehartree"] += -0.5 * (Ust - Jh) * n4 * gw + 0.25 * (Ust - Jh) * gw
    o["Epot"] += o["Ehartree"]
    return o


def twin_sector_order(S):
    """Order(1:Dim) of twin_sector_order (ED_SETUP.f90:852-880): flip every state of sector S to
    |{dw}>|{up}>, take its global number (+ phonon offset) and return the sorting permutation (1-based),
    so that vector_twin(i) = cvec(Order(i)) (ED_EIGENSPACE.f90:486-497)."""
    Ns_Ud, Ns_Orb = S.Ns_Ud, S.Ns_Orb
    dims = [int(x) for x in list(S.DimUps) + list(S.DimDws)]
    H = [S.map(k) for k in range(2 * Ns_Ud)]
    iel = S.DimUp * S.DimDw
    Ns = S.Ns
    flipped = np.zeros(S.Dim, dtype=object)
    for i in range(1, S.Dim + 1):
        iph = (i - 1) // iel + 1
        c = (i - 1) % iel; ind = []
        for d in dims:
            ind.append(c % d + 1); c //= d
        ist = [int(H[k][ind[k] - 1]) for k in range(2 * Ns_Ud)]
        jv = ist[Ns_Ud:] + ist[:Ns_Ud]                      # flip_state: [jups,jdws] = [dws,ups]
        j = jv[0]; prod = 1
        for k in range(1, len(jv)):                         # indices2state with dims = 2**Ns_Orb
            prod *= 2 ** Ns_Orb; j += (jv[k] - 1) * prod
        flipped[i - 1] = j + (iph - 1) * 2 ** (2 * Ns)
    order = sorted(range(S.Dim), key=lambda k: flipped[k])
    return np.array(order, dtype=np.int64) + 1


def imp_density_offdiag(S, state):
    """Off-diagonal part of imp_density_matrix (ED_OBSERVABLES.f90:283-313, ed_total_ud=T): loops over the
    states i, applies c_jorb then c^+_iorb to the spin word and accumulates sgn1*state(i)*sgn2*state(j).
    Returns dm[s, iorb-1, jorb-1], s = 0 up / 1 dw."""
    Norb = S.model["Norb"]
    H = [S.map(0), S.map(1)]
    dims = [int(S.DimUps[0]), int(S.DimDws[0])]
    iel = S.DimUp * S.DimDw
    dm = np.zeros((2, Norb, Norb))
    for i in range(1, S.Dim + 1):
        iph = (i - 1) // iel + 1
        c = (i - 1) % iel
        ind = [c % dims[0] + 1, c // dims[0] + 1]
        iud = [int(H[0][ind[0] - 1]), int(H[1][ind[1] - 1])]
        for ispin in (1, 2):
            for iorb in range(1, Norb + 1):
                for jorb in range(1, Norb + 1):
                    nj = (iud[ispin - 1] >> (jorb - 1)) & 1; ni = (iud[ispin - 1] >> (iorb - 1)) & 1
                    if nj == 1 and ni == 0:
                        r, sgn1 = c_op(jorb, iud[ispin - 1])
                        k, sgn2 = cdg_op(iorb, r)
                        jnd = list(ind)
                        jnd[ispin - 1] = binary_search(H[ispin - 1], k)
                        j = jnd[0] + (jnd[1] - 1) * dims[0] + (iph - 1) * iel
                        dm[ispin - 1, iorb - 1, jorb - 1] += sgn1 * state[i - 1] * sgn2 * state[j - 1]
    return dm
