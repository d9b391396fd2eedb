// sector_host.hpp -- host-side construction of one (Nup,Ndw) sector: bit-pattern maps,
// per-spin hopping factors and the per-spin energy tables the device kernels consume.
//
// What it computes is specified by the reference (cited per function); how it computes it
// is not: rows of the per-spin blocks are generated row-wise (gather form) from a flat list
// of one-body hop terms instead of the reference's column sweep with duplicate-summing
// inserts, and ranks come from a 2^Ns lookup table instead of a recursive binary search.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/edgpu.h"

namespace edgpu {

struct HopTerm {      // c^+_alpha c_beta with amplitude t (levels 1-based, one spin block)
    int32_t alpha, beta;
    double t;
};

struct Factor {       // one per-spin block H_sigma (spH0ups(iud) / spH0dws(iud))
    int32_t nbits = 0;             // levels in this block (Ns or 1+Nbath)
    int32_t npart = 0;             // electrons
    int32_t dim = 0;
    std::vector<int32_t> map;      // ascending bit patterns, Hs(k)%map
    std::vector<int32_t> rank;     // inverse: pattern -> 0-based rank, -1 elsewhere (2^nbits)
    std::vector<HopTerm> terms;
    std::vector<int64_t> rowptr;   // row-wise CSR, cols ascending (reference storage order)
    std::vector<int32_t> cols;     // 0-based
    std::vector<double> vals;
    int32_t width = 0;             // max nnz per row (ELL width)
    std::vector<double> E;         // per-state one-body + same-spin energy table
};

inline int popc(uint32_t x) { return __builtin_popcount(x); }

// ED_SETUP.f90:1017-1035 (binomial via running product, rounded)
inline int64_t binomial(int n1, int n2)
{
    if (n2 < 0) return 0;
    if (n2 == 0) return 1;
    double xh = 1.0;
    for (int i = 1; i <= n2; i++) xh = xh * double(n1 + 1 - i) / double(i);
    return (int64_t)(xh + 0.5);
}

// sign of c_pos or c^+_pos acting on pattern m: (-1)^{# occupied levels below pos}
// ED_SETUP.f90:805-831
inline double fsign(uint32_t m, int pos) { return (popc(m & ((1u << (pos - 1)) - 1u)) & 1) ? -1.0 : 1.0; }

struct SectorHost {
    edgpu_desc desc{};
    std::vector<double> impHloc, diag_hybr, bath_diag, hbath;  // owned copies
    int Ns = 0, Ns_Orb = 0, Ns_Ud = 0, DimPh = 1, Jhflag = 0, isector = 0;
    int stride[EDGPU_MAX_ORB][64];      // getBathStride(iorb,kp), 1-based level index
    std::vector<Factor> fac;            // [ups(0..Ns_Ud-1), dws(0..Ns_Ud-1)]
    // ed_total_ud=F: the composite up / dw blocks H_up = sum_a 1 x .. x H_a x .. x 1 over the mixed-radix up (dw) index,
    // as two explicit factors, so that the sector runs through the two-factor product kernels (empty when ed_total_ud=T)
    std::vector<Factor> cfac;
    int64_t DimUp = 1, DimDw = 1, Dim = 0;
    int64_t Qdw = 0, dw0 = 0;           // local dw columns [dw0, dw0+Qdw)
    double econst = 0.0;                // state-independent part of the diagonal
    std::vector<double> cross;          // cross[mu_imp*2^Norb + md_imp]: U n_up n_dw + Ust cross terms
    std::vector<double> gph_up, gph_dw; // e-ph electronic factor tables over impurity bits
    double gph_sum = 0.0;

    double imp(int s1, int s2, int a, int b) const { return impHloc[(s1 - 1) + desc.Nspin * ((s2 - 1) + desc.Nspin * ((a - 1) + (size_t)desc.Norb * (b - 1)))]; }
    double hyb(int s, int a, int k) const { return diag_hybr[(s - 1) + desc.Nspin * ((a - 1) + (size_t)desc.Norb * (k - 1))]; }
    double bdiag(int s, int a, int k) const { return bath_diag[(s - 1) + desc.Nspin * ((a - 1) + (size_t)desc.Nfoo * (k - 1))]; }
    double hb(int s1, int s2, int a, int b, int k) const
    {
        return hbath[(s1 - 1) + desc.Nspin * ((s2 - 1) + desc.Nspin * ((a - 1) + (size_t)desc.Norb * ((b - 1) + (size_t)desc.Norb * (k - 1))))];
    }

    // returns "" or an error message (the reference would `stop` with it)
    std::string init(const edgpu_desc &d)
    {
        desc = d;
        if (d.abi_version != EDGPU_ABI_VERSION) return "edgpu_desc.abi_version mismatch";
        if (d.Nspin < 1 || d.Nspin > 2) return "ED ERROR: Nspin > 2 is currently not supported";            // ED_SETUP.f90:65
        if (d.Norb < 1 || d.Norb > EDGPU_MAX_ORB) return "ED ERROR: Norb > 5 is currently not supported";   // ED_SETUP.f90:66
        if (d.Nbath < 1 || d.Nbath > 60) return "edgpu: Nbath out of range";
        if (d.Nph < 0) return "edgpu: Nph < 0";
        if (!d.impHloc || !d.diag_hybr || !d.bath_diag) return "edgpu: impHloc/diag_hybr/bath_diag must be given";
        if (d.bath_type == EDGPU_BATH_REPLICA && !d.hbath) return "edgpu: bath_type=replica needs hbath";
        if (d.mpi_size < 1 || d.mpi_rank < 0 || d.mpi_rank >= d.mpi_size) return "edgpu: bad mpi_rank/mpi_size";
        // ed_setup_dimensions, ED_SETUP.f90:113-135
        if (d.bath_type == EDGPU_BATH_HYBRID) {
            Ns = d.Nbath + d.Norb;
            if (!d.ed_total_ud) return "ED ERROR: ed_total_ud=F can not be used with bath_type=hybrid";    // ED_SETUP.f90:70
        } else
            Ns = (d.Nbath + 1) * d.Norb;
        if (d.ed_total_ud) { Ns_Orb = Ns; Ns_Ud = 1; } else { Ns_Orb = Ns / d.Norb; Ns_Ud = d.Norb; }
        if (Ns_Orb > 26) return "edgpu: more than 26 levels per spin block is not supported";
        DimPh = d.Nph + 1;
        Jhflag = (d.Norb > 1 && (d.Jx != 0.0 || d.Jp != 0.0)) ? 1 : 0;                                     // ED_SETUP.f90:147-148
        if (!d.ed_total_ud && Jhflag) return "ED ERROR: ed_total_ud=F can not be used with Jx!=0 OR Jp!=0";  // ED_SETUP.f90:71
        int nfoo_expected = d.bath_type == EDGPU_BATH_HYBRID ? 1 : d.Norb;
        if (d.Nfoo != nfoo_expected) return "edgpu: Nfoo must be size(bath_diag,2)";
        size_t n4 = (size_t)d.Nspin * d.Nspin * d.Norb * d.Norb;
        impHloc.assign(d.impHloc, d.impHloc + n4);
        diag_hybr.assign(d.diag_hybr, d.diag_hybr + (size_t)d.Nspin * d.Norb * d.Nbath);
        bath_diag.assign(d.bath_diag, d.bath_diag + (size_t)d.Nspin * d.Nfoo * d.Nbath);
        if (d.hbath) hbath.assign(d.hbath, d.hbath + n4 * d.Nbath);
        // getBathStride, ED_SETUP.f90:358-375
        for (int i = 1; i <= d.Nbath; i++)
            for (int a = 1; a <= d.Norb; a++) {
                int v;
                if (d.bath_type == EDGPU_BATH_HYBRID) v = d.Norb + i;
                else if (d.bath_type == EDGPU_BATH_REPLICA) v = a + i * d.Norb;
                else v = d.Norb + (a - 1) * d.Nbath + i;
                stride[a - 1][i - 1] = v;
            }
        // sector number, ED_SETUP.f90:446-457 with QN=[Nups,Ndws]
        {
            int64_t F = Ns_Orb + 1, s = 1, pw = 1;
            for (int q = 2 * Ns_Ud; q >= 1; q--) {
                int qn = q <= Ns_Ud ? d.nups[q - 1] : d.ndws[q - 1 - Ns_Ud];
                if (qn < 0 || qn > Ns_Orb) return "edgpu: sector quantum number out of range";
                s += qn * pw; pw *= F;
            }
            isector = (int)s;
        }
        // build_sector, ED_SETUP.f90:745-780
        fac.assign(2 * Ns_Ud, Factor());
        DimUp = DimDw = 1;
        for (int k = 0; k < 2 * Ns_Ud; k++) {
            Factor &f = fac[k];
            f.nbits = Ns_Orb;
            f.npart = k < Ns_Ud ? d.nups[k] : d.ndws[k - Ns_Ud];
            f.dim = (int32_t)binomial(Ns_Orb, f.npart);
            f.map.reserve(f.dim);
            f.rank.assign((size_t)1 << Ns_Orb, -1);
            for (uint32_t m = 0; m < (1u << Ns_Orb); m++)
                if (popc(m) == f.npart) { f.rank[m] = (int32_t)f.map.size(); f.map.push_back((int32_t)m); }
            if ((int)f.map.size() != f.dim) return "edgpu: internal sector dimension mismatch";
            (k < Ns_Ud ? DimUp : DimDw) *= f.dim;
        }
        Dim = DimUp * DimDw * DimPh;
        // dw split, ED_HAMILTONIAN.f90:96-110 (first mod(DimDw,P) ranks own one more column)
        if (DimDw < d.mpi_size) return "edgpu: DimDw < number of ranks (the reference shrinks the communicator, ED_HAMILTONIAN.f90:65-93); use fewer ranks";
        {
            int64_t q = DimDw / d.mpi_size, r = DimDw % d.mpi_size;
            if (d.mpi_rank < r) { Qdw = q + 1; dw0 = d.mpi_rank * (q + 1); } else { Qdw = q; dw0 = d.mpi_rank * q + r; }
        }
        build_terms();
        for (auto &f : fac) build_rows(f);
        build_energy_tables();
        if (!d.ed_total_ud) build_composite();
        return "";
    }

    // Composite factor of the Ns_Ud per-orbital blocks fac[first .. first+Ns_Ud): index = sum_a dig_a * prod_{b<a} dim_b (the
    // reference's state order, ED_SETUP.f90 state2indices), rows = Kronecker sum with ascending columns, every entry with
    // the sign its own orbital block gave it (stored/Orbs/H_up.f90:7-30 acts on one orbital's word at a time), E = sum of
    // the blocks' tables, map = impurity word (bit a = impurity level of orbital a+1 occupied).
    void build_composite()
    {
        cfac.assign(2, Factor());
        for (int side = 0; side < 2; side++) {
            Factor &c = cfac[side];
            const int first = side * Ns_Ud;
            int64_t dim = 1; std::vector<int64_t> st(Ns_Ud);
            for (int a = 0; a < Ns_Ud; a++) { st[a] = dim; dim *= fac[first + a].dim; }
            int64_t wsum = 0; for (int a = 0; a < Ns_Ud; a++) wsum += fac[first + a].width;
            if (dim > ((int64_t)1 << 24) || dim * std::max<int64_t>(wsum, 1) > ((int64_t)1 << 27)) { cfac.clear(); return; }   // too large to tabulate (1.6 GB of table): generic kernel
            c.nbits = Ns; c.npart = 0; c.dim = (int32_t)dim;
            c.rank.assign(1, 0);
            for (int a = 0; a < Ns_Ud; a++) for (auto &t : fac[first + a].terms) c.terms.push_back(t);
            c.map.resize(dim); c.E.resize(dim); c.rowptr.assign((size_t)dim + 1, 0);
            std::vector<int32_t> dig(Ns_Ud, 0);
            struct Ent { int32_t j; double v; };
            std::vector<Ent> row;
            for (int64_t i = 0; i < dim; i++) {
                uint32_t m = 0; double e = 0.0; row.clear();
                for (int a = 0; a < Ns_Ud; a++) {
                    const Factor &f = fac[first + a];
                    m |= ((uint32_t)f.map[dig[a]] & 1u) << a; e += f.E[dig[a]];
                    for (int64_t q = f.rowptr[dig[a]]; q < f.rowptr[dig[a] + 1]; q++) row.push_back({(int32_t)(i + (int64_t)(f.cols[q] - dig[a]) * st[a]), f.vals[q]});
                }
                std::sort(row.begin(), row.end(), [](const Ent &x, const Ent &y) { return x.j < y.j; });
                for (auto &r : row) { c.cols.push_back(r.j); c.vals.push_back(r.v); }
                c.rowptr[i + 1] = (int64_t)c.cols.size();
                c.width = std::max<int32_t>(c.width, (int32_t)row.size());
                c.map[i] = (int32_t)m; c.E[i] = e;
                for (int a = 0; a < Ns_Ud; a++) { if (++dig[a] < fac[first + a].dim) break; dig[a] = 0; }
            }
        }
    }

    // Flatten the one-body couplings into hop terms in the statement order of
    // stored/H_up.f90:8-81 (== H_dw.f90 with spin Nspin) or stored/Orbs/H_up.f90:7-30.
    void build_terms()
    {
        const edgpu_desc &d = desc;
        for (int k = 0; k < 2 * Ns_Ud; k++) {
            Factor &f = fac[k];
            int spin = k < Ns_Ud ? 1 : d.Nspin;
            f.terms.clear();
            if (d.ed_total_ud) {
                for (int a = 1; a <= d.Norb; a++)
                    for (int b = 1; b <= d.Norb; b++)
                        if (imp(spin, spin, a, b) != 0.0 && a != b) f.terms.push_back({a, b, imp(spin, spin, a, b)});
                if (d.bath_type == EDGPU_BATH_REPLICA)
                    for (int kp = 1; kp <= d.Nbath; kp++)
                        for (int a = 1; a <= d.Norb; a++)
                            for (int b = 1; b <= d.Norb; b++)
                                if (hb(spin, spin, a, b, kp) != 0.0 && a != b) f.terms.push_back({stride[a - 1][kp - 1], stride[b - 1][kp - 1], hb(spin, spin, a, b, kp)});
                for (int a = 1; a <= d.Norb; a++)
                    for (int kp = 1; kp <= d.Nbath; kp++)
                        if (hyb(spin, a, kp) != 0.0) {
                            f.terms.push_back({stride[a - 1][kp - 1], a, hyb(spin, a, kp)});  // imp -> bath
                            f.terms.push_back({a, stride[a - 1][kp - 1], hyb(spin, a, kp)});  // bath -> imp
                        }
            } else {
                int a = (k % Ns_Ud) + 1;  // orbital of this block; level 1 = impurity, 1+kp = bath kp
                for (int kp = 1; kp <= d.Nbath; kp++)
                    if (hyb(spin, a, kp) != 0.0) {
                        f.terms.push_back({1 + kp, 1, hyb(spin, a, kp)});
                        f.terms.push_back({1, 1 + kp, hyb(spin, a, kp)});
                    }
            }
        }
    }

    // Row i of H_sigma: all (j, t*sg1*sg2) with m_i = c^+_alpha c_beta m_j.  Entries are ordered
    // by ascending j (the reference appends while sweeping j upwards) and equal j are summed.
    static void build_rows(Factor &f)
    {
        struct E { int32_t j; int32_t ord; double v; };
        f.rowptr.assign((size_t)f.dim + 1, 0);
        f.cols.clear(); f.vals.clear(); f.width = 0;
        std::vector<E> row;
        for (int32_t i = 0; i < f.dim; i++) {
            uint32_t mi = (uint32_t)f.map[i];
            row.clear();
            for (size_t t = 0; t < f.terms.size(); t++) {
                const HopTerm &h = f.terms[t];
                uint32_t ba = 1u << (h.alpha - 1), bb = 1u << (h.beta - 1);
                if (!(mi & ba) || (mi & bb)) continue;       // target has alpha, lacks beta
                uint32_t k1 = mi & ~ba;                      // after c_beta on the source
                uint32_t mj = k1 | bb;                       // source pattern
                double sg1 = fsign(mj, h.beta), sg2 = fsign(k1, h.alpha);
                row.push_back({f.rank[mj], (int32_t)t, h.t * sg1 * sg2});
            }
            std::stable_sort(row.begin(), row.end(), [](const E &a, const E &b) { return a.j < b.j; });
            int32_t n = 0;
            for (size_t e = 0; e < row.size(); e++) {
                if (n > 0 && f.cols.back() == row[e].j) { f.vals.back() += row[e].v; continue; }
                f.cols.push_back(row[e].j); f.vals.push_back(row[e].v); n++;
            }
            f.rowptr[i + 1] = (int64_t)f.cols.size();
            f.width = std::max(f.width, n);
        }
    }

    // Per-spin tables for the on-the-fly diagonal (DIRECT storage); same terms as
    // stored/H_local.f90:13-71, regrouped as const + E_up + E_dw + cross(imp bits).
    void build_energy_tables()
    {
        const edgpu_desc &d = desc;
        const int Norb = d.Norb;
        int npairs = Norb * (Norb - 1) / 2;
        econst = 0.0;
        if (d.hfmode) {
            for (int a = 0; a < Norb; a++) econst += 0.25 * d.Uloc[a];
            econst += npairs * (0.25 * d.Ust + 0.25 * (d.Ust - d.Jh));
        }
        // level -> impurity orbital / (orbital,bath) for the *global* level layout
        for (int k = 0; k < 2 * Ns_Ud; k++) {
            Factor &f = fac[k];
            int spin = k < Ns_Ud ? 1 : d.Nspin;
            f.E.assign(f.dim, 0.0);
            for (int32_t i = 0; i < f.dim; i++) {
                uint32_t m = (uint32_t)f.map[i];
                double e = 0.0;
                if (d.ed_total_ud) {
                    for (int a = 1; a <= Norb; a++)
                        if (m & (1u << (a - 1))) {
                            e += imp(spin, spin, a, a) - d.xmu;
                            if (d.hfmode) e += -0.5 * d.Uloc[a - 1] - (Norb - 1) * (0.5 * d.Ust + 0.5 * (d.Ust - d.Jh));
                        }
                    for (int a = 1; a <= Norb; a++)
                        for (int b = a + 1; b <= Norb; b++)
                            if ((m & (1u << (a - 1))) && (m & (1u << (b - 1)))) e += d.Ust - d.Jh;
                    for (int a = 1; a <= d.Nfoo; a++)
                        for (int kp = 1; kp <= d.Nbath; kp++)
                            if (m & (1u << (stride[a - 1][kp - 1] - 1))) e += bdiag(spin, a, kp);
                } else {
                    int a = (k % Ns_Ud) + 1;
                    if (m & 1u) {
                        e += imp(spin, spin, a, a) - d.xmu;
                        if (d.hfmode) e += -0.5 * d.Uloc[a - 1] - (Norb - 1) * (0.5 * d.Ust + 0.5 * (d.Ust - d.Jh));
                    }
                    for (int kp = 1; kp <= d.Nbath; kp++)
                        if (m & (1u << kp)) e += bdiag(spin, a, kp);
                    // same-spin inter-orbital (Ust-Jh) n_a n_b couples different blocks: goes to `cross`
                }
                f.E[i] = e;
            }
        }
        int nI = 1 << Norb;
        cross.assign((size_t)nI * nI, 0.0);
        gph_up.assign(nI, 0.0); gph_dw.assign(nI, 0.0); gph_sum = 0.0;
        for (int a = 0; a < Norb; a++) gph_sum += d.g_ph[a];
        for (int mu = 0; mu < nI; mu++) {
            for (int a = 0; a < Norb; a++) if (mu & (1 << a)) { gph_up[mu] += d.g_ph[a]; gph_dw[mu] += d.g_ph[a]; }
            for (int md = 0; md < nI; md++) {
                double e = 0.0;
                for (int a = 0; a < Norb; a++) if ((mu >> a) & (md >> a) & 1) e += d.Uloc[a];
                for (int a = 0; a < Norb; a++)
                    for (int b = a + 1; b < Norb; b++) {
                        int ua = (mu >> a) & 1, ub = (mu >> b) & 1, da = (md >> a) & 1, db = (md >> b) & 1;
                        e += d.Ust * (ua * db + ub * da);
                        if (!d.ed_total_ud) e += (d.Ust - d.Jh) * (ua * ub + da * db);
                    }
                cross[(size_t)mu * nI + md] = e;
            }
        }
    }

    // spH0nd rows (stored/H_non_local.f90:21-83) for inspection/export only; the product path
    // evaluates these terms on the fly inside the kernels.
    void nonlocal_row(int64_t iup, int64_t idw, std::vector<int64_t> &cols, std::vector<double> &vals) const
    {
        cols.clear(); vals.clear();
        if (!Jhflag) return;
        const edgpu_desc &d = desc;
        uint32_t mup = (uint32_t)fac[0].map[iup], mdw = (uint32_t)fac[1].map[idw];
        auto add = [&](int64_t j, double v) {
            for (size_t k = 0; k < cols.size(); k++) if (cols[k] == j) { vals[k] += v; return; }
            cols.push_back(j); vals.push_back(v);
        };
        for (int pass = 0; pass < 2; pass++) {
            double J = pass == 0 ? d.Jx : d.Jp;
            if (J == 0.0) continue;
            for (int a = 1; a <= d.Norb; a++)
                for (int b = 1; b <= d.Norb; b++) {
                    uint32_t ba = 1u << (a - 1), bb = 1u << (b - 1);
                    uint32_t k2, k4; double s;
                    if (pass == 0) {  // spin exchange: c_a(dw) c^+_b(dw) ; c_b(up) c^+_a(up)
                        if (a == b || !(mup & bb) || !(mdw & ba) || (mdw & bb) || (mup & ba)) continue;
                        uint32_t k1 = mdw & ~ba; k2 = k1 | bb; uint32_t k3 = mup & ~bb; k4 = k3 | ba;
                        s = fsign(mdw, a) * fsign(k1, b) * fsign(mup, b) * fsign(k3, a);
                    } else {          // pair hopping: c_b(dw) c^+_a(dw) ; c_b(up) c^+_a(up)
                        if (!(mup & bb) || !(mdw & bb) || (mdw & ba) || (mup & ba)) continue;
                        uint32_t k1 = mdw & ~bb; k2 = k1 | ba; uint32_t k3 = mup & ~bb; k4 = k3 | ba;
                        s = fsign(mdw, b) * fsign(k1, a) * fsign(mup, b) * fsign(k3, a);
                    }
                    add((int64_t)fac[0].rank[k4] + (int64_t)fac[1].rank[k2] * DimUp, J * s);
                }
        }
    }
};

}  // namespace edgpu
