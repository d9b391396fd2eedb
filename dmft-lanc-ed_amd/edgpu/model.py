"""Resolved impurity-model inputs for the sector operator.

The reference resolves its flat input file plus the bath object into a few
dense arrays right before it builds a sector Hamiltonian
(ED_HAMILTONIAN_SPARSE_HxV.f90:43-75): ``impHloc``, ``diag_hybr``,
``bath_diag`` and, for replica baths, ``hbath_tmp``.  Those arrays -- not the
input file -- are what crosses the C-ABI boundary (include/edgpu.h,
``edgpu_desc``).  This module produces them for synthetic inputs:

* parameter defaults: ED_INPUT_VARS.f90:129-208
* default bath initialisation: ED_BATH/dmft_aux.f90:102-153
* level layout / sector numbering: ED_SETUP.f90:113-135, 358-375, 446-457

Arrays are Fortran-ordered with the reference's index order.
"""
import numpy as np

BATH_TYPES = ("normal", "hybrid", "replica")


def linspace_like(start, stop, num):
    """SciFortran ``linspace`` with both end points (ED_BATH/dmft_aux.f90:112)."""
    return np.linspace(start, stop, num)


def default_bath_normal(Nspin, Norb, Nbath, hwband=2.0, bath_type="normal"):
    """(e, v) of ``init_dmft_bath`` for bath_type normal/hybrid
    (ED_BATH/dmft_aux.f90:116-140; allocation shapes 29-37)."""
    nfoo = 1 if bath_type == "hybrid" else Norb
    e = np.zeros((Nspin, nfoo, Nbath), order="F")
    v = np.zeros((Nspin, Norb, Nbath), order="F")
    e[:, :, 0] = -hwband
    e[:, :, Nbath - 1] = hwband
    Nh = Nbath // 2
    if Nbath % 2 == 0 and Nbath >= 4:
        de = hwband / max(Nh - 1, 1)
        e[:, :, Nh - 1] = -0.1
        e[:, :, Nh] = 0.1
        for i in range(2, Nh):
            e[:, :, i - 1] = -hwband + (i - 1) * de
            e[:, :, Nbath - i] = hwband - (i - 1) * de
    elif Nbath % 2 != 0 and Nbath >= 3:
        de = hwband / Nh
        e[:, :, Nh] = 0.0
        for i in range(2, Nh + 1):
            e[:, :, i - 1] = -hwband + (i - 1) * de
            e[:, :, Nbath - i] = hwband - (i - 1) * de
    v[:, :, :] = max(0.1, 1.0 / np.sqrt(float(Nbath)))
    return e, v


def default_bath_replica(Nspin, Norb, Nbath, H_basis, lambda_imp, hwband=2.0):
    """(v[Nspin,Nbath], lambda[Nsym,Nbath]) of ``init_dmft_bath`` for replica
    baths (ED_BATH/dmft_aux.f90:141-160): diagonal basis elements get
    lambda_k = rescale(k)*lambda_imp, others lambda_imp."""
    rescale = linspace_like(hwband / Nbath, hwband, Nbath) if Nbath > 1 else np.zeros(1)
    v = np.full((Nspin, Nbath), max(0.1, 1.0 / np.sqrt(float(Nbath))))
    lam = np.zeros((len(lambda_imp), Nbath))
    for k in range(Nbath):
        for s, O in enumerate(H_basis):
            m = np.abs(_nn2so(O))
            off = m - np.diag(np.diag(m))
            lam[s, k] = rescale[k] * lambda_imp[s] if not (off > 1e-6).any() else lambda_imp[s]
    return v, lam


def _nn2so(O):
    Nspin, _, Norb, _ = O.shape
    M = np.zeros((Nspin * Norb, Nspin * Norb))
    for a in range(Nspin):
        for b in range(Nspin):
            for i in range(Norb):
                for j in range(Norb):
                    M[a * Norb + i, b * Norb + j] = O[a, b, i, j]
    return M


def make_model(Norb=1, Nspin=1, Nbath=5, bath_type="normal", ed_total_ud=True, Nph=0,
               Uloc=(2.0, 0.0, 0.0, 0.0, 0.0), Ust=0.0, Jh=0.0, Jx=0.0, Jp=0.0, xmu=0.0,
               hfmode=True, hwband=2.0, impHloc=None, H_basis=None, lambda_imp=None,
               w0_ph=0.0, g_ph=(0.0,) * 5, bath_e=None, bath_v=None, bath_lambda=None):
    """Resolve inputs into the arrays the sector builder consumes.

    ``impHloc`` is real (ED_VARS_GLOBAL.f90:120).  For replica baths pass
    ``H_basis`` (list of (Nspin,Nspin,Norb,Norb) arrays) and ``lambda_imp``;
    impHloc = sum lambda*H_basis (ED_HLOC_DECOMPOSITION.f90:181-198).
    """
    assert bath_type in BATH_TYPES
    Uloc = tuple(Uloc) + (0.0,) * (5 - len(Uloc))
    m = dict(Norb=Norb, Nspin=Nspin, Nbath=Nbath, bath_type=bath_type, ed_total_ud=bool(ed_total_ud),
             Nph=Nph, Uloc=Uloc, Ust=Ust, Jh=Jh, Jx=Jx, Jp=Jp, xmu=xmu, hfmode=bool(hfmode),
             w0_ph=w0_ph, g_ph=tuple(g_ph) + (0.0,) * (5 - len(g_ph)), hbath=None)
    if bath_type == "replica":
        assert H_basis is not None and lambda_imp is not None
        H_basis = [np.asarray(O, float) for O in H_basis]
        imp = sum(l * O for l, O in zip(lambda_imp, H_basis))
        v, lam = default_bath_replica(Nspin, Norb, Nbath, H_basis, lambda_imp, hwband)
        if bath_v is not None:
            v = np.asarray(bath_v, float)
        if bath_lambda is not None:
            lam = np.asarray(bath_lambda, float)
        hb = np.zeros((Nspin, Nspin, Norb, Norb, Nbath), order="F")
        for k in range(Nbath):  # bath_from_sym, ED_BATH/dmft_aux.f90:69-82
            hb[..., k] = sum(lam[s, k] * H_basis[s] for s in range(len(H_basis)))
        dh = np.zeros((Nspin, Norb, Nbath), order="F")
        bd = np.zeros((Nspin, Norb, Nbath), order="F")
        for k in range(Nbath):  # ED_HAMILTONIAN_SPARSE_HxV.f90:63-74
            for s in range(Nspin):
                for a in range(Norb):
                    dh[s, a, k] = v[s, k]
                    bd[s, a, k] = hb[s, s, a, a, k]
        m.update(impHloc=np.asfortranarray(imp), diag_hybr=dh, bath_diag=bd, hbath=hb)
    else:
        imp = np.zeros((Nspin, Nspin, Norb, Norb), order="F") if impHloc is None else np.asfortranarray(np.asarray(impHloc, float))
        e, v = default_bath_normal(Nspin, Norb, Nbath, hwband, bath_type)
        if bath_e is not None:
            e = np.asfortranarray(np.asarray(bath_e, float))
        if bath_v is not None:
            v = np.asfortranarray(np.asarray(bath_v, float))
        m.update(impHloc=imp, diag_hybr=v, bath_diag=e)
    m["Ns"] = (Nbath + Norb) if bath_type == "hybrid" else (Nbath + 1) * Norb
    m["Ns_Orb"] = m["Ns"] if ed_total_ud else m["Ns"] // Norb
    m["Ns_Ud"] = 1 if ed_total_ud else Norb
    return m


def get_sector(QN, N):
    """isector = 1 + sum_i QN(i)*(N+1)**(Nind-i)  (ED_SETUP.f90:446-457)."""
    isector = 1
    n = len(QN)
    for i in range(n, 0, -1):
        isector += int(QN[i - 1]) * (N + 1) ** (n - i)
    return isector


def dw_partition(DimDw, size, rank):
    """The reference's dw-column split (ED_HAMILTONIAN.f90:96-110):
    returns (mpiQdw, first_dw_column0) for ``rank`` of ``size``."""
    q, r = divmod(DimDw, size)
    if rank < r:
        return q + 1, rank * (q + 1)
    return q, rank * q + r


def half_filled_c(Nbath):
    """BASELINE family c1..c4: 1 orbital, normal bath, Nup=Ndw=Ns/2."""
    m = make_model(Norb=1, Nspin=1, Nbath=Nbath)
    ns = m["Ns"]
    return m, [ns // 2], [ns - ns // 2]
