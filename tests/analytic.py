"""Known answers that do NOT pass through oracle/ or through the product: test infrastructure only.

Three independent anchors for the sector Hamiltonian the reference builds:

* ``one_body`` / ``free_*``: with every interaction switched off the model is a quadratic form
  ``sum_s sum_ij h^s_ij c^+_is c_js`` and every sector eigenvalue is a sum of one-body levels; the impurity Green's
  function is ``[(z - h)^-1]_aa`` whatever the filling.  ``h`` is written down from the model arrays
  (``impHloc``, ``diag_hybr``, ``bath_diag``, ``hbath``) and the level layout of ED_SETUP.f90:358-375.
* ``dimer_spectrum``: the closed-form (1,1)-sector spectrum of the symmetric two-site Anderson model.
* ``fock_hamiltonian``: the interacting Hamiltonian as textbook operator algebra -- Jordan-Wigner matrices
  ``c_j = 1 x .. x a x Z x .. x Z`` on the full Fock space, products of those for every term of the Kanamori
  Hamiltonian (stored/H_local.f90:13-71, H_non_local.f90:21-83 give the *definition* of the terms; nothing of the
  reference's loop structure, bit tricks or sign routines is used), then restricted to the states of one sector.
  With the up modes ordered before the dw modes the JW signs of a bilinear reduce to the per-word signs the reference
  uses, and ascending Fock integers are the reference's state order ``i = iup + (idw-1)*DimUp`` -- so the matrices
  must agree ELEMENT BY ELEMENT, not just in spectrum.
"""
import itertools

import numpy as np
import scipy.sparse as sp


def _spin_index(m, spin):
    """impHloc(1,..) for up, impHloc(Nspin,..) for dw (stored/H_local.f90:17-18)"""
    return 0 if spin == 0 else m["Nspin"] - 1


def bath_site(m, a, k):
    """0-based position of bath level k (0-based) of orbital a in the Ns-bit word (getBathStride, ED_SETUP.f90:358-375)"""
    Norb, Nbath = m["Norb"], m["Nbath"]
    bt = m.get("bath_type", "normal")
    if bt == "hybrid":
        return Norb + k
    if bt == "replica":
        return a + (k + 1) * Norb
    return Norb + a * Nbath + k


def one_body(m, spin):
    """One-body matrix of one spin species for ed_total_ud=T (Ns x Ns) or the list of per-orbital matrices
    ((1+Nbath) x (1+Nbath) each) for ed_total_ud=F."""
    s = _spin_index(m, spin)
    Norb, Nbath = m["Norb"], m["Nbath"]
    imp, V, e = np.asarray(m["impHloc"]), np.asarray(m["diag_hybr"]), np.asarray(m["bath_diag"])
    xmu = m.get("xmu", 0.0)
    if not m.get("ed_total_ud", True):
        hs = []
        for a in range(Norb):
            h = np.zeros((Nbath + 1, Nbath + 1))
            h[0, 0] = imp[s, s, a, a] - xmu
            for k in range(Nbath):
                h[k + 1, k + 1] = e[s, a, k]
                h[0, k + 1] = h[k + 1, 0] = V[s, a, k]
            hs.append(h)
        return hs
    Ns = m["Ns"]
    h = np.zeros((Ns, Ns))
    for a in range(Norb):
        for b in range(Norb):
            h[a, b] = imp[s, s, a, b]
        h[a, a] -= xmu
    for a in range(e.shape[1]):
        for k in range(Nbath):
            p = bath_site(m, a, k)
            h[p, p] += e[s, a, k]
    if m.get("bath_type") == "replica":
        hb = np.asarray(m["hbath"])
        for k in range(Nbath):
            for a in range(Norb):
                for b in range(Norb):
                    if a != b:
                        h[bath_site(m, a, k), bath_site(m, b, k)] += hb[s, s, a, b, k]
    for a in range(Norb):
        for k in range(Nbath):
            p = bath_site(m, a, k)
            h[a, p] += V[s, a, k]
            h[p, a] += V[s, a, k]
    return h


def non_interacting(m, **over):
    """copy of the model with every interaction (and the Hartree shift, which only exists with them) removed"""
    z = dict(m, Uloc=(0.0,) * 5, Ust=0.0, Jh=0.0, Jx=0.0, Jp=0.0, hfmode=False, g_ph=(0.0,) * 5)
    z.update(over)
    return z


def _blocks(m, nups, ndws):
    """[(levels, filling)] of every independent species of the quadratic model"""
    nups, ndws = list(np.atleast_1d(nups)), list(np.atleast_1d(ndws))
    out = []
    for spin, ns in ((0, nups), (1, ndws)):
        h = one_body(m, spin)
        if isinstance(h, list):
            out += [(np.linalg.eigvalsh(ha), int(n)) for ha, n in zip(h, ns)]
        else:
            out.append((np.linalg.eigvalsh(h), int(ns[0])))
    return out


def free_e0(m, nups, ndws):
    """sector ground-state energy of the quadratic model and its gap to the first excited level"""
    e0, gap = 0.0, np.inf
    for lev, n in _blocks(m, nups, ndws):
        e0 += lev[:n].sum()
        if 0 < n < lev.size:
            gap = min(gap, lev[n] - lev[n - 1])
    return float(e0), float(gap)


def free_spectrum(m, nups, ndws):
    """all eigenvalues of the sector (sorted): every way of filling the one-body levels"""
    tot = np.zeros(1)
    for lev, n in _blocks(m, nups, ndws):
        sums = np.array([lev[list(c)].sum() for c in itertools.combinations(range(lev.size), n)])
        tot = (tot[:, None] + sums[None, :]).ravel()
    return np.sort(tot)


def free_gf(m, iorb, ispin, z):
    """G_aa,s(z) = [(z - h_s)^-1]_aa of the quadratic model; iorb, ispin 1-based as in the reference"""
    h = one_body(m, ispin - 1)
    z = np.atleast_1d(z)
    if isinstance(h, list):
        h, a = h[iorb - 1], 0
    else:
        a = iorb - 1
    w, U = np.linalg.eigh(h)
    return ((U[a, :] ** 2)[None, :] / (z[:, None] - w[None, :])).sum(axis=1)


def dimer_spectrum(U, V):
    """(1,1) sector of the symmetric two-site Anderson model (eps_d = eps_bath = 0, Hartree-shifted U(n_up-1/2)(n_dw-1/2)):
    singlet pair +-sqrt(U^2/16 + 4V^2), triplet -U/4, antisymmetric ionic state +U/4"""
    r = np.sqrt(U * U / 16.0 + 4.0 * V * V)
    return np.sort(np.array([-r, -U / 4.0, U / 4.0, r]))


# ---------------------------------------------------------------------------------------------------------------------
# operator algebra on the full Fock space
# ---------------------------------------------------------------------------------------------------------------------
def _popcount(x):
    x = np.asarray(x, dtype=np.uint64)
    n = np.zeros(x.shape, np.int64)
    for q in range(40):
        n += ((x >> np.uint64(q)) & np.uint64(1)).astype(np.int64)
    return n


def _jw(M):
    """annihilators c_0..c_{M-1}; mode j is bit j of the Fock integer (c_j = 1^(M-1-j) x a x Z^j, leftmost = top bit)"""
    a = sp.csr_matrix(np.array([[0.0, 1.0], [0.0, 0.0]]))
    Z = sp.csr_matrix(np.diag([1.0, -1.0]))
    ops = []
    for j in range(M):
        o = sp.identity(1, format="csr")
        for q in range(M - 1, -1, -1):
            o = sp.kron(o, a if q == j else (Z if q < j else sp.identity(2, format="csr")), format="csr")
        ops.append(o)
    return ops


def fock_hamiltonian(m, nups, ndws):
    """Dense sector Hamiltonian in the reference's state order, from operator algebra (see module docstring)."""
    Norb, Nbath, Ns = m["Norb"], m["Nbath"], m["Ns"]
    total = m.get("ed_total_ud", True)
    nups, ndws = list(np.atleast_1d(nups)), list(np.atleast_1d(ndws))
    M = 2 * Ns
    c = _jw(M)
    cd = [o.T.tocsr() for o in c]
    n = [cd[j] @ c[j] for j in range(M)]
    if total:
        mode = lambda s, p: s * Ns + p
        imp = lambda s, a: mode(s, a)
    else:
        L = Nbath + 1
        mode = lambda s, p: s * Ns + p          # p = a*L + level
        imp = lambda s, a: mode(s, a * L)
    H = sp.csr_matrix((1 << M, 1 << M))
    I = sp.identity(1 << M, format="csr")
    for s in (0, 1):
        h = one_body(m, s)
        if isinstance(h, list):
            hh = np.zeros((Ns, Ns))
            for a, ha in enumerate(h):
                hh[a * L:(a + 1) * L, a * L:(a + 1) * L] = ha
            h = hh
        for i in range(Ns):
            for j in range(Ns):
                if h[i, j] != 0.0:
                    H = H + h[i, j] * (cd[mode(s, i)] @ c[mode(s, j)])
    Uloc, Ust, Jh, Jx, Jp = m["Uloc"], m.get("Ust", 0.0), m.get("Jh", 0.0), m.get("Jx", 0.0), m.get("Jp", 0.0)
    nu = [n[imp(0, a)] for a in range(Norb)]; nd = [n[imp(1, a)] for a in range(Norb)]
    for a in range(Norb):
        H = H + Uloc[a] * (nu[a] @ nd[a])
        if m.get("hfmode", True):
            H = H - 0.5 * Uloc[a] * (nu[a] + nd[a]) + 0.25 * Uloc[a] * I
    for a in range(Norb):
        for b in range(a + 1, Norb):
            H = H + Ust * (nu[a] @ nd[b] + nu[b] @ nd[a]) + (Ust - Jh) * (nu[a] @ nu[b] + nd[a] @ nd[b])
            if m.get("hfmode", True):
                nt = nu[a] + nd[a] + nu[b] + nd[b]
                H = H - 0.5 * Ust * nt + 0.25 * Ust * I - 0.5 * (Ust - Jh) * nt + 0.25 * (Ust - Jh) * I
    if total and Norb > 1 and (Jx != 0.0 or Jp != 0.0):
        for a in range(Norb):
            for b in range(Norb):
                if a == b:
                    continue
                if Jx != 0.0:   # J [c^+_b,dw c_a,dw][c^+_a,up c_b,up]
                    H = H + Jx * (cd[imp(1, b)] @ c[imp(1, a)]) @ (cd[imp(0, a)] @ c[imp(0, b)])
                if Jp != 0.0:   # J [c^+_a,dw c_b,dw][c^+_a,up c_b,up]
                    H = H + Jp * (cd[imp(1, a)] @ c[imp(1, b)]) @ (cd[imp(0, a)] @ c[imp(0, b)])
    # the states of the sector, ascending Fock integer
    idx = np.arange(1 << M, dtype=np.int64)
    keep = np.ones(1 << M, bool)
    if total:
        up = idx & ((1 << Ns) - 1); dw = idx >> Ns
        keep &= (_popcount(up) == nups[0]) & (_popcount(dw) == ndws[0])
    else:
        for s, ns in ((0, nups), (1, ndws)):
            for a in range(Norb):
                word = (idx >> (s * Ns + a * L)) & ((1 << L) - 1)
                keep &= _popcount(word) == ns[a]
    sel = idx[keep]
    Hel = H.tocsr()[sel][:, sel].toarray()
    Nph = m.get("Nph", 0)
    if Nph <= 0:
        return Hel
    # phonons: H_el x 1 + 1 x w0 b^+b + [sum_a g_a (n_a - 1)] x (b + b^+)   (stored/H_ph.f90, H_e_ph.f90)
    D = Nph + 1
    G = sp.csr_matrix((1 << M, 1 << M))
    g = m.get("g_ph", (0.0,) * 5)
    for a in range(Norb):
        G = G + g[a] * (nu[a] + nd[a] - I)
    Gel = G.tocsr()[sel][:, sel].toarray()
    b = np.diag(np.sqrt(np.arange(1, D)), 1)
    return np.kron(np.eye(D), Hel) + np.kron(m.get("w0_ph", 0.0) * np.diag(np.arange(D, dtype=float)), np.eye(sel.size)) \
        + np.kron(b + b.T, Gel)


def apply_c_fock(m, nups, ndws, iorb, ispin, dagger, state):
    """c / c^+ on a sector state through the same operator algebra, ed_total_ud=T without phonons:
    returns the (unnormalised) image in the target sector's state order."""
    Ns = m["Ns"]; M = 2 * Ns
    ops = _jw(M)
    o = ops[(ispin - 1) * Ns + iorb - 1]
    if dagger:
        o = o.T.tocsr()
    idx = np.arange(1 << M, dtype=np.int64)
    up = _popcount(idx & ((1 << Ns) - 1)); dw = _popcount(idx >> Ns)
    nu, nd = int(np.atleast_1d(nups)[0]), int(np.atleast_1d(ndws)[0])
    src = idx[(up == nu) & (dw == nd)]
    d = 1 if dagger else -1
    nu2, nd2 = (nu + d, nd) if ispin == 1 else (nu, nd + d)
    dst = idx[(up == nu2) & (dw == nd2)]
    full = np.zeros(1 << M); full[src] = state
    out = (o @ full)[dst]
    # the reference applies the dw operator to the dw word alone (ED_GF_NORMAL.f90:196-206): no string over the up
    # electrons, i.e. a factor (-1)^Nup relative to the global Jordan-Wigner operator
    if ispin == 2 and nu % 2:
        out = -out
    return out


def free_imp_occupations(m, nups, ndws):
    """<n_{a,up}>, <n_{a,dw}> of the quadratic model's sector ground state (ed_total_ud=T): sum over the filled one-body
    levels of |<a|level>|^2.  Uncorrelated spins: <n_up n_dw> = <n_up><n_dw>."""
    Norb = m["Norb"]
    out = []
    for spin, n in ((0, int(np.atleast_1d(nups)[0])), (1, int(np.atleast_1d(ndws)[0]))):
        w, U = np.linalg.eigh(one_body(m, spin))
        out.append(np.array([(U[a, :n] ** 2).sum() for a in range(Norb)]))
    return out[0], out[1]
