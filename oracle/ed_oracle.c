/*
 * ed_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * Plain-C restatement (single thread unless orc_set_threads is called) of the CPU algorithm of the reference
 * (lcrippa/dmft-lanc-ed) for the sector Hamiltonian-times-vector hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product (dmft-lanc-ed_amd/csrc) never links or calls it.
 *
 * Every function cites the reference file:line it follows.  Loop order,
 * accumulation order and 1-based indexing follow the reference so that the
 * floating point results are those of the reference's CPU path.
 *
 * Parity pin status: PARITY UNPINNED.  The reference holds no tests, fixtures or golden
 * vectors for this path and cannot be built in this repository (its modules need SciFortran,
 * DMFT_Tools and MPI; no stand-ins are written), so nothing of the reference itself pins this
 * restatement.  Its external anchors are the known-answer values the survey recorded
 * (tests/golden/survey_known_answers.json -- one of the six records is
 * marked unpinned there), dense diagonalisation, and the internal redundancy
 * stored == direct == dense*v == sampled rows (orc_rows).  See oracle/README.md.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_MAX_ORB 5
#define ORC_MAX_UD 5

/* ---- parameters the reference keeps in module globals --------------------
 * ED_INPUT_VARS.f90:129-208, ED_VARS_GLOBAL.f90:100-146.  Arrays are
 * column-major exactly as the Fortran ones. */
typedef struct {
    int Norb, Nspin, Nbath, Nph;
    int bath_type;   /* 0 normal, 1 hybrid, 2 replica */
    int ed_total_ud; /* 1 = T */
    int hfmode;
    int Nfoo;        /* size(bath_diag,2) */
    double Uloc[ORC_MAX_ORB], Ust, Jh, Jx, Jp, xmu, w0_ph, g_ph[ORC_MAX_ORB];
    const double *impHloc;   /* (Nspin,Nspin,Norb,Norb) */
    const double *diag_hybr; /* (Nspin,Norb,Nbath) */
    const double *bath_diag; /* (Nspin,Nfoo,Nbath) */
    const double *hbath;     /* (Nspin,Nspin,Norb,Norb,Nbath), replica only */
} orc_params;

/* growable row CSR: ED_SPARSE_MATRIX.f90:13-30 */
typedef struct { int size; double *vals; int64_t *cols; } orc_row;
typedef struct { orc_row *row; int64_t Nrow, Ncol; int status; } orc_sp;

typedef struct {
    orc_params p;
    int Ns, Ns_Orb, Ns_Ud, DimPh, Nsectors, Jhflag;
    int getBathStride[ORC_MAX_ORB][64]; /* (iorb,kp) 1-based values */
    int isector;
    int Nups[ORC_MAX_UD], Ndws[ORC_MAX_UD];
    int DimUps[ORC_MAX_UD], DimDws[ORC_MAX_UD];
    int64_t DimUp, DimDw, Dim;
    int *map[2 * ORC_MAX_UD]; /* Hs(1:2*Ns_Ud)%map, 1-based content at [0..] */
    orc_sp spH0d, spH0nd, spH0ups[ORC_MAX_UD], spH0dws[ORC_MAX_UD];
    orc_sp spH0_ph, spH0e_eph, spH0ph_eph;
    int built;
} orc_sector;

static void die(const char *msg) { fprintf(stderr, "%s\n", msg); abort(); }

/* ED_SETUP.f90:1017-1035 */
int orc_binomial(int n1, int n2)
{
    double xh = 1.0;
    if (n2 < 0) return 0;
    if (n2 == 0) return 1;
    for (int i = 1; i <= n2; i++) xh = xh * (double)(n1 + 1 - i) / (double)i;
    return (int)(xh + 0.5);
}

/* ED_SETUP.f90:446-457 */
int orc_get_sector(const int *QN, int Nind, int N)
{
    int Factor = N + 1, isector = 1;
    for (int i = Nind; i >= 1; i--) {
        int pw = 1;
        for (int k = 0; k < Nind - i; k++) pw *= Factor;
        isector += QN[i - 1] * pw;
    }
    return isector;
}

/* ED_SETUP.f90:476-500 (get_Nup, get_Ndw) */
static void get_Nup(const orc_sector *s, int isector, int *Nup)
{
    int idx[2 * ORC_MAX_UD], count = isector - 1;
    for (int i = 1; i <= 2 * s->Ns_Ud; i++) { idx[i - 1] = count % (s->Ns_Orb + 1); count /= (s->Ns_Orb + 1); }
    for (int k = 0; k < s->Ns_Ud; k++) Nup[k] = idx[2 * s->Ns_Ud - 1 - k];
}
static void get_Ndw(const orc_sector *s, int isector, int *Ndw)
{
    int idx[2 * ORC_MAX_UD], count = isector - 1;
    for (int i = 1; i <= 2 * s->Ns_Ud; i++) { idx[i - 1] = count % (s->Ns_Orb + 1); count /= (s->Ns_Orb + 1); }
    for (int k = 0; k < s->Ns_Ud; k++) Ndw[k] = idx[s->Ns_Ud - 1 - k];
}

/* ED_SETUP.f90:523-544 */
static int64_t indices2state(const int *ivec, const int *Nvec, int n)
{
    int64_t istate = ivec[0], prod = 1;
    for (int i = 2; i <= n; i++) { prod *= Nvec[i - 2]; istate += (int64_t)(ivec[i - 1] - 1) * prod; }
    return istate;
}
static void state2indices(int64_t istate, const int *Nvec, int n, int *ivec)
{
    int64_t count = istate - 1;
    for (int i = 1; i <= n; i++) { ivec[i - 1] = (int)(count % Nvec[i - 1]) + 1; count /= Nvec[i - 1]; }
}

/* ED_SETUP.f90:547-560 */
static int64_t iup_index(int64_t i, int64_t DimUp) { int64_t iup = i % DimUp; if (iup == 0) iup = DimUp; return iup; }
static int64_t idw_index(int64_t i, int64_t DimUp) { return (i - 1) / DimUp + 1; }

/* ED_SETUP.f90:937-947 */
static void bdecomp(int i, int Ntot, int *ivec) { for (int l = 0; l < Ntot; l++) ivec[l] = (i >> l) & 1; }

/* ED_SETUP.f90:805-831 */
static void op_c(int pos, int in, int *out, double *fsgn)
{
    if (!((in >> (pos - 1)) & 1)) die("C error: C_i|...0_i...>");
    *fsgn = 1.0;
    for (int l = 1; l <= pos - 1; l++) if ((in >> (l - 1)) & 1) *fsgn = -*fsgn;
    *out = in & ~(1 << (pos - 1));
}
static void op_cdg(int pos, int in, int *out, double *fsgn)
{
    if ((in >> (pos - 1)) & 1) die("C^+ error: C^+_i|...1_i...>");
    *fsgn = 1.0;
    for (int l = 1; l <= pos - 1; l++) if ((in >> (l - 1)) & 1) *fsgn = -*fsgn;
    *out = in | (1 << (pos - 1));
}
/* exported for the bit-exact sign tests */
void orc_c(int pos, int in, int *out, double *fsgn) { op_c(pos, in, out, fsgn); }
void orc_cdg(int pos, int in, int *out, double *fsgn) { op_cdg(pos, in, out, fsgn); }

/* ED_SETUP.f90:1042-1059 (recursive, 1-based, 0 = not found) */
static int binary_search_rec(const int *a, int n, int value)
{
    int mid = n / 2 + 1;
    if (n == 0) return 0;
    if (a[mid - 1] > value) return binary_search_rec(a, mid - 1, value);
    if (a[mid - 1] < value) {
        int r = binary_search_rec(a + mid, n - mid, value);
        return r != 0 ? mid + r : 0;
    }
    return mid;
}
int orc_binary_search(const int *a, int n, int value) { return binary_search_rec(a, n, value); }

/* ED_SPARSE_MATRIX.f90:127-172 */
static void sp_init(orc_sp *sp, int64_t N)
{
    if (sp->status) die("sp_init_matrix: already allocated can not init");
    sp->Nrow = N; sp->Ncol = N;
    sp->row = (orc_row *)calloc((size_t)(N > 0 ? N : 1), sizeof(orc_row));
    sp->status = 1;
}
/* ED_SPARSE_MATRIX.f90:184-237 */
static void sp_delete(orc_sp *sp)
{
    if (!sp->status) return;
    for (int64_t i = 0; i < sp->Nrow; i++) { free(sp->row[i].vals); free(sp->row[i].cols); }
    free(sp->row); sp->row = NULL; sp->status = 0; sp->Nrow = sp->Ncol = 0;
}
/* ED_SPARSE_MATRIX.f90:255-323: duplicate-summing append */
static void sp_insert(orc_sp *sp, double value, int64_t i, int64_t j)
{
    orc_row *row = &sp->row[i - 1];
    int pos = -1;
    for (int k = 0; k < row->size; k++) if (row->cols[k] == j) { pos = k; break; }
    if (pos >= 0) { row->vals[pos] += value; return; }
    row->vals = (double *)realloc(row->vals, sizeof(double) * (size_t)(row->size + 1));
    row->cols = (int64_t *)realloc(row->cols, sizeof(int64_t) * (size_t)(row->size + 1));
    row->vals[row->size] = value; row->cols[row->size] = j; row->size++;
    if (row->size > sp->Ncol) die("sp_insert_element_csr ERROR: row%Size > sparse%Ncol");
}

#define P4(a, i, j, k, l) ((a)[((i)-1) + S->p.Nspin * (((j)-1) + S->p.Nspin * (((k)-1) + (size_t)S->p.Norb * ((l)-1)))])
#define IMPHLOC(i, j, k, l) P4(S->p.impHloc, i, j, k, l)
#define DIAG_HYBR(is, io, kp) (S->p.diag_hybr[((is)-1) + S->p.Nspin * (((io)-1) + (size_t)S->p.Norb * ((kp)-1))])
#define BATH_DIAG(is, io, kp) (S->p.bath_diag[((is)-1) + S->p.Nspin * (((io)-1) + (size_t)S->p.Nfoo * ((kp)-1))])
#define HBATH(i, j, k, l, kp) (S->p.hbath[((i)-1) + S->p.Nspin * (((j)-1) + S->p.Nspin * (((k)-1) + (size_t)S->p.Norb * (((l)-1) + (size_t)S->p.Norb * ((kp)-1))))])

/* ED_SETUP.f90:113-135 (ed_setup_dimensions), 147-148 (Jhflag), 358-375 (getBathStride) */
orc_sector *orc_sector_open(const orc_params *p, int isector)
{
    orc_sector *S = (orc_sector *)calloc(1, sizeof(orc_sector));
    S->p = *p;
    if (p->bath_type == 1) { S->Ns = p->Nbath + p->Norb; if (!p->ed_total_ud) die("ed_setup_dimension: bath_type==hybrid AND .NOT.ed_total_ud"); }
    else S->Ns = (p->Nbath + 1) * p->Norb;
    if (p->ed_total_ud) { S->Ns_Orb = S->Ns; S->Ns_Ud = 1; } else { S->Ns_Orb = S->Ns / p->Norb; S->Ns_Ud = p->Norb; }
    S->DimPh = p->Nph + 1;
    { int64_t b = (int64_t)(S->Ns_Orb + 1) * (S->Ns_Orb + 1), n = 1; for (int k = 0; k < S->Ns_Ud; k++) n *= b; S->Nsectors = (int)n; }
    S->Jhflag = (p->Norb > 1 && (p->Jx != 0.0 || p->Jp != 0.0));
    if (!p->ed_total_ud && S->Jhflag) die("ED ERROR: ed_total_ud=F can not be used with Jx!=0 OR Jp!=0");
    for (int i = 1; i <= p->Nbath; i++)
        for (int iorb = 1; iorb <= p->Norb; iorb++) {
            int v;
            if (p->bath_type == 1) v = p->Norb + i;
            else if (p->bath_type == 2) v = iorb + i * p->Norb;
            else v = p->Norb + (iorb - 1) * p->Nbath + i;
            S->getBathStride[iorb - 1][i - 1] = v;
        }
    /* build_sector: ED_SETUP.f90:745-780 */
    S->isector = isector;
    get_Nup(S, isector, S->Nups); get_Ndw(S, isector, S->Ndws);
    S->DimUp = 1; S->DimDw = 1;
    for (int iud = 0; iud < S->Ns_Ud; iud++) {
        S->DimUps[iud] = orc_binomial(S->Ns_Orb, S->Nups[iud]);
        S->DimDws[iud] = orc_binomial(S->Ns_Orb, S->Ndws[iud]);
        S->DimUp *= S->DimUps[iud]; S->DimDw *= S->DimDws[iud];
    }
    S->Dim = S->DimUp * S->DimDw * S->DimPh;
    for (int iud = 0; iud < S->Ns_Ud; iud++) {
        int dim = 0;
        S->map[iud] = (int *)malloc(sizeof(int) * (size_t)S->DimUps[iud]);
        for (int iup = 0; iup <= (1 << S->Ns_Orb) - 1; iup++) {
            if (__builtin_popcount((unsigned)iup) != S->Nups[iud]) continue;
            S->map[iud][dim++] = iup;
        }
        dim = 0;
        S->map[iud + S->Ns_Ud] = (int *)malloc(sizeof(int) * (size_t)S->DimDws[iud]);
        for (int idw = 0; idw <= (1 << S->Ns_Orb) - 1; idw++) {
            if (__builtin_popcount((unsigned)idw) != S->Ndws[iud]) continue;
            S->map[iud + S->Ns_Ud][dim++] = idw;
        }
    }
    return S;
}

static void free_matrices(orc_sector *S)
{
    sp_delete(&S->spH0d); sp_delete(&S->spH0nd);
    for (int k = 0; k < ORC_MAX_UD; k++) { sp_delete(&S->spH0ups[k]); sp_delete(&S->spH0dws[k]); }
    sp_delete(&S->spH0_ph); sp_delete(&S->spH0e_eph); sp_delete(&S->spH0ph_eph);
    S->built = 0;
}

/* delete_Hv_sector: ED_HAMILTONIAN.f90:174-222 */
void orc_sector_close(orc_sector *S)
{
    if (!S) return;
    free_matrices(S);
    for (int k = 0; k < 2 * ORC_MAX_UD; k++) free(S->map[k]);
    free(S);
}

void orc_sector_info(const orc_sector *S, int *Ns, int *Ns_Orb, int *Ns_Ud, int *DimPh, int *Nsectors,
                     int *DimUps, int *DimDws, int64_t *Dim)
{
    *Ns = S->Ns; *Ns_Orb = S->Ns_Orb; *Ns_Ud = S->Ns_Ud; *DimPh = S->DimPh; *Nsectors = S->Nsectors;
    for (int k = 0; k < S->Ns_Ud; k++) { DimUps[k] = S->DimUps[k]; DimDws[k] = S->DimDws[k]; }
    *Dim = S->Dim;
}
void orc_sector_map(const orc_sector *S, int k, int *out)
{
    int n = k < S->Ns_Ud ? S->DimUps[k] : S->DimDws[k - S->Ns_Ud];
    memcpy(out, S->map[k], sizeof(int) * (size_t)n);
}

/* breorder: ED_SETUP.f90:963-979 */
static void breorder(const orc_sector *S, int Nins[ORC_MAX_UD][64], int *Ivec)
{
    if (S->p.ed_total_ud) { for (int l = 0; l < S->Ns; l++) Ivec[l] = Nins[0][l]; return; }
    for (int iud = 1; iud <= S->Ns_Ud; iud++) {
        Ivec[iud - 1] = Nins[iud - 1][0];
        for (int ib = 1; ib <= S->p.Nbath; ib++) Ivec[S->getBathStride[iud - 1][ib - 1] - 1] = Nins[iud - 1][ib];
    }
}

/* occupations of electronic state i_el (1-based); main: stored/H_local.f90:2-9,
 * orbs: stored/Orbs/H_local.f90:2-10 */
static void occupations(const orc_sector *S, int64_t i_el, int *nup, int *ndw, int *mup_out, int *mdw_out)
{
    if (S->p.ed_total_ud) {
        int64_t iup = iup_index(i_el, S->DimUp), idw = idw_index(i_el, S->DimUp);
        int mup = S->map[0][iup - 1], mdw = S->map[1][idw - 1];
        bdecomp(mup, S->Ns, nup); bdecomp(mdw, S->Ns, ndw);
        if (mup_out) *mup_out = mup;
        if (mdw_out) *mdw_out = mdw;
    } else {
        int Nvec[2 * ORC_MAX_UD], Indices[2 * ORC_MAX_UD];
        int Nups[ORC_MAX_UD][64], Ndws[ORC_MAX_UD][64];
        for (int k = 0; k < S->Ns_Ud; k++) { Nvec[k] = S->DimUps[k]; Nvec[k + S->Ns_Ud] = S->DimDws[k]; }
        state2indices(i_el, Nvec, 2 * S->Ns_Ud, Indices);
        for (int iud = 0; iud < S->Ns_Ud; iud++) {
            int mup = S->map[iud][Indices[iud] - 1], mdw = S->map[iud + S->Ns_Ud][Indices[iud + S->Ns_Ud] - 1];
            bdecomp(mup, S->Ns_Orb, Nups[iud]); bdecomp(mdw, S->Ns_Orb, Ndws[iud]);
        }
        breorder(S, Nups, nup); breorder(S, Ndws, ndw);
    }
}

/* stored/H_local.f90:13-71 == direct/HxV_local.f90:14-73 (same statement order) */
static double h_local(const orc_sector *S, const int *nup, const int *ndw)
{
    const orc_params *p = &S->p;
    int Norb = p->Norb, Nspin = p->Nspin;
    double htmp = 0.0;
    for (int iorb = 1; iorb <= Norb; iorb++) {
        htmp = htmp + IMPHLOC(1, 1, iorb, iorb) * nup[iorb - 1];
        htmp = htmp + IMPHLOC(Nspin, Nspin, iorb, iorb) * ndw[iorb - 1];
        htmp = htmp - p->xmu * (nup[iorb - 1] + ndw[iorb - 1]);
    }
    for (int iorb = 1; iorb <= Norb; iorb++) htmp = htmp + p->Uloc[iorb - 1] * nup[iorb - 1] * ndw[iorb - 1];
    if (Norb > 1) {
        for (int iorb = 1; iorb <= Norb; iorb++)
            for (int jorb = iorb + 1; jorb <= Norb; jorb++)
                htmp = htmp + p->Ust * (nup[iorb - 1] * ndw[jorb - 1] + nup[jorb - 1] * ndw[iorb - 1]);
        for (int iorb = 1; iorb <= Norb; iorb++)
            for (int jorb = iorb + 1; jorb <= Norb; jorb++)
                htmp = htmp + (p->Ust - p->Jh) * (nup[iorb - 1] * nup[jorb - 1] + ndw[iorb - 1] * ndw[jorb - 1]);
    }
    if (p->hfmode) {
        for (int iorb = 1; iorb <= Norb; iorb++)
            htmp = htmp - 0.5 * p->Uloc[iorb - 1] * (nup[iorb - 1] + ndw[iorb - 1]) + 0.25 * p->Uloc[iorb - 1];
        if (Norb > 1)
            for (int iorb = 1; iorb <= Norb; iorb++)
                for (int jorb = iorb + 1; jorb <= Norb; jorb++) {
                    htmp = htmp - 0.5 * p->Ust * (nup[iorb - 1] + ndw[iorb - 1] + nup[jorb - 1] + ndw[jorb - 1]) + 0.25 * p->Ust;
                    htmp = htmp - 0.5 * (p->Ust - p->Jh) * (nup[iorb - 1] + ndw[iorb - 1] + nup[jorb - 1] + ndw[jorb - 1]) + 0.25 * (p->Ust - p->Jh);
                }
    }
    for (int iorb = 1; iorb <= p->Nfoo; iorb++)
        for (int kp = 1; kp <= p->Nbath; kp++) {
            int ialfa = S->getBathStride[iorb - 1][kp - 1];
            htmp = htmp + BATH_DIAG(1, iorb, kp) * nup[ialfa - 1];
            htmp = htmp + BATH_DIAG(Nspin, iorb, kp) * ndw[ialfa - 1];
        }
    return htmp;
}

/* one-spin hop enumeration shared by stored/H_up.f90, H_dw.f90 (insert) and
 * direct/HxV_up.f90, HxV_dw.f90 (apply).  `spin` is 1 (up) or Nspin (dw).
 * Calls emit(ctx, irow, value) for each allowed hop of source pattern m, in
 * the statement order of the reference include files. */
typedef void (*hop_emit)(void *ctx, int irow, double val);
static void hops_main(const orc_sector *S, int spin, const int *map, int dim, int m, hop_emit emit, void *ctx)
{
    const orc_params *p = &S->p;
    int n[64], k1, k2; double sg1, sg2;
    bdecomp(m, S->Ns, n);
    /* H_imp off-diagonal: stored/H_up.f90:8-23 */
    for (int iorb = 1; iorb <= p->Norb; iorb++)
        for (int jorb = 1; jorb <= p->Norb; jorb++) {
            if (IMPHLOC(spin, spin, iorb, jorb) != 0.0 && n[jorb - 1] == 1 && n[iorb - 1] == 0) {
                op_c(jorb, m, &k1, &sg1); op_cdg(iorb, k1, &k2, &sg2);
                emit(ctx, binary_search_rec(map, dim, k2), IMPHLOC(spin, spin, iorb, jorb) * sg1 * sg2);
            }
        }
    /* replica bath-bath: stored/H_up.f90:27-51 */
    if (p->bath_type == 2)
        for (int kp = 1; kp <= p->Nbath; kp++)
            for (int iorb = 1; iorb <= p->Norb; iorb++)
                for (int jorb = 1; jorb <= p->Norb; jorb++) {
                    int ialfa = S->getBathStride[iorb - 1][kp - 1], ibeta = S->getBathStride[jorb - 1][kp - 1];
                    if (HBATH(spin, spin, iorb, jorb, kp) != 0.0 && n[ibeta - 1] == 1 && n[ialfa - 1] == 0) {
                        op_c(ibeta, m, &k1, &sg1); op_cdg(ialfa, k1, &k2, &sg2);
                        emit(ctx, binary_search_rec(map, dim, k2), HBATH(spin, spin, iorb, jorb, kp) * sg1 * sg2);
                    }
                }
    /* hybridisation: stored/H_up.f90:55-81 */
    for (int iorb = 1; iorb <= p->Norb; iorb++)
        for (int kp = 1; kp <= p->Nbath; kp++) {
            int ialfa = S->getBathStride[iorb - 1][kp - 1];
            if (DIAG_HYBR(spin, iorb, kp) != 0.0 && n[iorb - 1] == 1 && n[ialfa - 1] == 0) {
                op_c(iorb, m, &k1, &sg1); op_cdg(ialfa, k1, &k2, &sg2);
                emit(ctx, binary_search_rec(map, dim, k2), DIAG_HYBR(spin, iorb, kp) * sg1 * sg2);
            }
            if (DIAG_HYBR(spin, iorb, kp) != 0.0 && n[iorb - 1] == 0 && n[ialfa - 1] == 1) {
                op_c(ialfa, m, &k1, &sg1); op_cdg(iorb, k1, &k2, &sg2);
                emit(ctx, binary_search_rec(map, dim, k2), DIAG_HYBR(spin, iorb, kp) * sg1 * sg2);
            }
        }
}
/* stored/Orbs/H_up.f90:7-30 (per-orbital block, level 1 = impurity, 1+kp = bath) */
static void hops_orbs(const orc_sector *S, int spin, int iorb, const int *map, int dim, int m, hop_emit emit, void *ctx)
{
    const orc_params *p = &S->p;
    int n[64], k1, k2; double sg1, sg2;
    bdecomp(m, S->Ns_Orb, n);
    for (int kp = 1; kp <= p->Nbath; kp++) {
        int ialfa = 1 + kp;
        if (DIAG_HYBR(spin, iorb, kp) != 0.0 && n[0] == 1 && n[ialfa - 1] == 0) {
            op_c(1, m, &k1, &sg1); op_cdg(ialfa, k1, &k2, &sg2);
            emit(ctx, binary_search_rec(map, dim, k2), DIAG_HYBR(spin, iorb, kp) * sg1 * sg2);
        }
        if (DIAG_HYBR(spin, iorb, kp) != 0.0 && n[0] == 0 && n[ialfa - 1] == 1) {
            op_c(ialfa, m, &k1, &sg1); op_cdg(1, k1, &k2, &sg2);
            emit(ctx, binary_search_rec(map, dim, k2), DIAG_HYBR(spin, iorb, kp) * sg1 * sg2);
        }
    }
}

/* non-local terms of source state (mup,mdw); stored/H_non_local.f90:21-83 ==
 * direct/HxV_non_local.f90:17-69.  emit(ctx, jup, jdw, val) */
typedef void (*nl_emit)(void *ctx, int jup, int jdw, double val);
static void nonlocal_terms(const orc_sector *S, int mup, int mdw, const int *nup, const int *ndw, nl_emit emit, void *ctx)
{
    const orc_params *p = &S->p;
    int k1, k2, k3, k4; double sg1, sg2, sg3, sg4;
    if (S->Jhflag && p->Jx != 0.0)
        for (int iorb = 1; iorb <= p->Norb; iorb++)
            for (int jorb = 1; jorb <= p->Norb; jorb++)
                if (iorb != jorb && nup[jorb - 1] == 1 && ndw[iorb - 1] == 1 && ndw[jorb - 1] == 0 && nup[iorb - 1] == 0) {
                    op_c(iorb, mdw, &k1, &sg1); op_cdg(jorb, k1, &k2, &sg2);
                    int jdw = binary_search_rec(S->map[1], (int)S->DimDw, k2);
                    op_c(jorb, mup, &k3, &sg3); op_cdg(iorb, k3, &k4, &sg4);
                    int jup = binary_search_rec(S->map[0], (int)S->DimUp, k4);
                    emit(ctx, jup, jdw, p->Jx * sg1 * sg2 * sg3 * sg4);
                }
    if (S->Jhflag && p->Jp != 0.0)
        for (int iorb = 1; iorb <= p->Norb; iorb++)
            for (int jorb = 1; jorb <= p->Norb; jorb++)
                if (nup[jorb - 1] == 1 && ndw[jorb - 1] == 1 && ndw[iorb - 1] == 0 && nup[iorb - 1] == 0) {
                    op_c(jorb, mdw, &k1, &sg1); op_cdg(iorb, k1, &k2, &sg2);
                    int jdw = binary_search_rec(S->map[1], (int)S->DimDw, k2);
                    op_c(jorb, mup, &k3, &sg3); op_cdg(iorb, k3, &k4, &sg4);
                    int jup = binary_search_rec(S->map[0], (int)S->DimUp, k4);
                    emit(ctx, jup, jdw, p->Jp * sg1 * sg2 * sg3 * sg4);
                }
}

typedef struct { orc_sp *sp; int64_t jcol; } ins_ctx;
static void emit_insert(void *c, int irow, double val) { ins_ctx *x = (ins_ctx *)c; sp_insert(x->sp, val, irow, x->jcol); }
typedef struct { orc_sector *S; int64_t i; } nl_ins_ctx;
static void emit_nl_insert(void *c, int jup, int jdw, double val)
{
    nl_ins_ctx *x = (nl_ins_ctx *)c;
    sp_insert(&x->S->spH0nd, val, x->i, jup + (int64_t)(jdw - 1) * x->S->DimUp);
}

/* ed_buildh_main: ED_HAMILTONIAN_SPARSE_HxV.f90:25-200; ed_buildh_orbs: 206-370 (serial, MpiStatus=F) */
void orc_buildh(orc_sector *S)
{
    const orc_params *p = &S->p;
    int nup[64], ndw[64], mup, mdw;
    int64_t Nel = S->DimUp * S->DimDw;
    free_matrices(S);
    sp_init(&S->spH0d, Nel);
    if (S->DimPh > 1) sp_init(&S->spH0e_eph, Nel);
    if (p->ed_total_ud && S->Jhflag) sp_init(&S->spH0nd, Nel);
    if (p->ed_total_ud) { sp_init(&S->spH0dws[0], S->DimDw); sp_init(&S->spH0ups[0], S->DimUp); }
    else for (int iud = 0; iud < S->Ns_Ud; iud++) { sp_init(&S->spH0dws[iud], S->DimDws[iud]); sp_init(&S->spH0ups[iud], S->DimUps[iud]); }
    if (S->DimPh > 1) { sp_init(&S->spH0_ph, S->DimPh); sp_init(&S->spH0ph_eph, S->DimPh); }
    /* H_local */
    for (int64_t i = 1; i <= Nel; i++) {
        occupations(S, i, nup, ndw, NULL, NULL);
        sp_insert(&S->spH0d, h_local(S, nup, ndw), i, i);
    }
    /* H_non_local (stored transposed: row i = source state) */
    if (p->ed_total_ud && S->Jhflag)
        for (int64_t i = 1; i <= Nel; i++) {
            nl_ins_ctx c = { S, i };
            occupations(S, i, nup, ndw, &mup, &mdw);
            nonlocal_terms(S, mup, mdw, nup, ndw, emit_nl_insert, &c);
        }
    /* H_up, H_dw */
    if (p->ed_total_ud) {
        for (int jup = 1; jup <= S->DimUp; jup++) { ins_ctx c = { &S->spH0ups[0], jup }; hops_main(S, 1, S->map[0], (int)S->DimUp, S->map[0][jup - 1], emit_insert, &c); }
        for (int jdw = 1; jdw <= S->DimDw; jdw++) { ins_ctx c = { &S->spH0dws[0], jdw }; hops_main(S, p->Nspin, S->map[1], (int)S->DimDw, S->map[1][jdw - 1], emit_insert, &c); }
    } else {
        for (int iorb = 1; iorb <= S->Ns_Ud; iorb++)
            for (int jup = 1; jup <= S->DimUps[iorb - 1]; jup++) { ins_ctx c = { &S->spH0ups[iorb - 1], jup }; hops_orbs(S, 1, iorb, S->map[iorb - 1], S->DimUps[iorb - 1], S->map[iorb - 1][jup - 1], emit_insert, &c); }
        for (int iorb = 1; iorb <= S->Ns_Ud; iorb++)
            for (int jdw = 1; jdw <= S->DimDws[iorb - 1]; jdw++) { ins_ctx c = { &S->spH0dws[iorb - 1], jdw }; hops_orbs(S, p->Nspin, iorb, S->map[iorb - 1 + S->Ns_Ud], S->DimDws[iorb - 1], S->map[iorb - 1 + S->Ns_Ud][jdw - 1], emit_insert, &c); }
    }
    if (S->DimPh > 1) {
        /* stored/H_ph.f90:2-5 */
        for (int iph = 1; iph <= S->DimPh; iph++) sp_insert(&S->spH0_ph, p->w0_ph * (iph - 1), iph, iph);
        /* stored/H_e_ph.f90:2-39 */
        for (int64_t i = 1; i <= Nel; i++) {
            double htmp = 0.0;
            occupations(S, i, nup, ndw, NULL, NULL);
            for (int iorb = 1; iorb <= p->Norb; iorb++) htmp = htmp + p->g_ph[iorb - 1] * (nup[iorb - 1] + ndw[iorb - 1] - 1.0);
            sp_insert(&S->spH0e_eph, htmp, i, i);
        }
        for (int iph = 1; iph <= S->DimPh; iph++) {
            int i = iph + 1;
            if (i <= S->DimPh) sp_insert(&S->spH0ph_eph, sqrt((double)iph), iph, i);
            i = iph - 1;
            if (i > 0) sp_insert(&S->spH0ph_eph, sqrt((double)(iph - 1)), iph, i);
        }
    }
    S->built = 1;
}

/* which: 0 spH0d, 1 spH0nd, 2 spH0ups(iud), 3 spH0dws(iud), 4 spH0_ph, 5 spH0e_eph, 6 spH0ph_eph */
static orc_sp *pick(orc_sector *S, int which, int iud)
{
    switch (which) {
    case 0: return &S->spH0d; case 1: return &S->spH0nd; case 2: return &S->spH0ups[iud]; case 3: return &S->spH0dws[iud];
    case 4: return &S->spH0_ph; case 5: return &S->spH0e_eph; default: return &S->spH0ph_eph;
    }
}
int64_t orc_sp_nnz(orc_sector *S, int which, int iud, int64_t *nrow)
{
    orc_sp *sp = pick(S, which, iud); int64_t nnz = 0;
    if (!sp->status) { *nrow = 0; return 0; }
    for (int64_t i = 0; i < sp->Nrow; i++) nnz += sp->row[i].size;
    *nrow = sp->Nrow; return nnz;
}
/* rows in storage (= insertion) order, 1-based columns as in the reference */
void orc_sp_export(orc_sector *S, int which, int iud, int64_t *rowptr, int64_t *cols, double *vals)
{
    orc_sp *sp = pick(S, which, iud); int64_t k = 0;
    rowptr[0] = 0;
    for (int64_t i = 0; i < sp->Nrow; i++) {
        for (int j = 0; j < sp->row[i].size; j++) { cols[k] = sp->row[i].cols[j]; vals[k] = sp->row[i].vals[j]; k++; }
        rowptr[i + 1] = k;
    }
}

/* spMatVec_main: ED_HAMILTONIAN_SPARSE_HxV.f90:391-485 */
static int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n < 1 ? 1 : n; }

static void spmatvec_main(orc_sector *S, const double *v, double *Hv)
{
    int64_t Nloc = S->Dim, DimUp = S->DimUp, DimDw = S->DimDw, Nel = DimUp * DimDw;
#define V(j) v[(j)-1]
#define HV(i) Hv[(i)-1]
    /* Every loop below writes only HV(i) of its own row i (gather form), so the optional OpenMP split over
       rows (orc_set_threads > 1; the CPU-baseline leg of bench.py) leaves each row's summation order, and
       therefore every bit of the result, unchanged.  The reference itself is single-threaded. */
    for (int64_t i = 0; i < Nloc; i++) Hv[i] = 0.0;
    _Pragma("omp parallel for schedule(static) num_threads(orc_threads)")
    for (int64_t i = 1; i <= Nloc; i++) {
        int64_t iph = (i - 1) / Nel + 1, i_el = (i - 1) % Nel + 1;
        orc_row *r = &S->spH0d.row[i_el - 1];
        for (int j_el = 0; j_el < r->size; j_el++) { double val = r->vals[j_el]; int64_t j = r->cols[j_el] + (iph - 1) * Nel; HV(i) = HV(i) + val * V(j); }
    }
    for (int64_t iph = 1; iph <= S->DimPh; iph++) {
        _Pragma("omp parallel for schedule(static) num_threads(orc_threads)")
        for (int64_t iup = 1; iup <= DimUp; iup++)
            for (int64_t idw = 1; idw <= DimDw; idw++) {
                int64_t i = iup + (idw - 1) * DimUp + (iph - 1) * Nel;
                orc_row *r = &S->spH0dws[0].row[idw - 1];
                for (int jj = 0; jj < r->size; jj++) { int64_t j = iup + (r->cols[jj] - 1) * DimUp + (iph - 1) * Nel; HV(i) = HV(i) + r->vals[jj] * V(j); }
            }
        _Pragma("omp parallel for schedule(static) num_threads(orc_threads)")
        for (int64_t idw = 1; idw <= DimDw; idw++)
            for (int64_t iup = 1; iup <= DimUp; iup++) {
                int64_t i = iup + (idw - 1) * DimUp + (iph - 1) * Nel;
                orc_row *r = &S->spH0ups[0].row[iup - 1];
                for (int jj = 0; jj < r->size; jj++) { int64_t j = r->cols[jj] + (idw - 1) * DimUp + (iph - 1) * Nel; HV(i) = HV(i) + r->vals[jj] * V(j); }
            }
        if (S->DimPh > 1)
            for (int64_t i_el = 1; i_el <= Nel; i_el++) {
                int64_t i = i_el + (iph - 1) * Nel;
                orc_row *rp = &S->spH0_ph.row[iph - 1];
                for (int jj = 0; jj < rp->size; jj++) { int64_t j = i_el + (rp->cols[jj] - 1) * Nel; HV(i) = HV(i) + rp->vals[jj] * V(j); }
                orc_row *re = &S->spH0e_eph.row[i_el - 1], *rq = &S->spH0ph_eph.row[iph - 1];
                for (int j_el = 0; j_el < re->size; j_el++)
                    for (int jj = 0; jj < rq->size; jj++) {
                        double val = re->vals[j_el] * rq->vals[jj];
                        int64_t j = re->cols[j_el] + (rq->cols[jj] - 1) * Nel;
                        HV(i) = HV(i) + val * V(j);
                    }
            }
    }
    if (S->Jhflag)
        for (int64_t i = 1; i <= Nloc; i++) {
            int64_t iph = (i - 1) / Nel + 1, i_el = (i - 1) % Nel + 1;
            orc_row *r = &S->spH0nd.row[i_el - 1];
            for (int j_el = 0; j_el < r->size; j_el++) { int64_t j = r->cols[j_el] + (iph - 1) * Nel; HV(i) = HV(i) + r->vals[j_el] * V(j); }
        }
}

/* spMatVec_orbs: ED_HAMILTONIAN_SPARSE_HxV.f90:487-564 */
static void spmatvec_orbs(orc_sector *S, const double *v, double *Hv)
{
    int64_t Nloc = S->Dim, Nel = S->DimUp * S->DimDw;
    int Nvec[2 * ORC_MAX_UD], Indices[2 * ORC_MAX_UD], Jndices[2 * ORC_MAX_UD], n2 = 2 * S->Ns_Ud;
    for (int k = 0; k < S->Ns_Ud; k++) { Nvec[k] = S->DimUps[k]; Nvec[k + S->Ns_Ud] = S->DimDws[k]; }
    for (int64_t i = 0; i < Nloc; i++) Hv[i] = 0.0;
    for (int64_t i = 1; i <= Nloc; i++) {
        int64_t i_el = (i - 1) % Nel + 1;
        orc_row *r = &S->spH0d.row[i_el - 1];
        for (int j = 0; j < r->size; j++) HV(i) = HV(i) + r->vals[j] * V(i);
    }
    for (int64_t i = 1; i <= Nloc; i++) {
        int64_t i_el = (i - 1) % Nel + 1, iph = (i - 1) / Nel + 1;
        state2indices(i_el, Nvec, n2, Indices);
        for (int iud = 0; iud < S->Ns_Ud; iud++) {
            int iup = Indices[iud];
            orc_row *r = &S->spH0ups[iud].row[iup - 1];
            for (int jj = 0; jj < r->size; jj++) {
                memcpy(Jndices, Indices, sizeof(int) * (size_t)n2); Jndices[iud] = (int)r->cols[jj];
                int64_t j = indices2state(Jndices, Nvec, n2) + (iph - 1) * Nel;
                HV(i) = HV(i) + r->vals[jj] * V(j);
            }
            int idw = Indices[iud + S->Ns_Ud];
            r = &S->spH0dws[iud].row[idw - 1];
            for (int jj = 0; jj < r->size; jj++) {
                memcpy(Jndices, Indices, sizeof(int) * (size_t)n2); Jndices[iud + S->Ns_Ud] = (int)r->cols[jj];
                int64_t j = indices2state(Jndices, Nvec, n2) + (iph - 1) * Nel;
                HV(i) = HV(i) + r->vals[jj] * V(j);
            }
        }
    }
    if (S->DimPh > 1)
        for (int64_t i = 1; i <= Nloc; i++) {
            int64_t i_el = (i - 1) % Nel + 1, iph = (i - 1) / Nel + 1;
            orc_row *rp = &S->spH0_ph.row[iph - 1];
            for (int jj = 0; jj < rp->size; jj++) { int64_t j = i_el + (rp->cols[jj] - 1) * Nel; HV(i) = HV(i) + rp->vals[jj] * V(j); }
            orc_row *re = &S->spH0e_eph.row[i_el - 1], *rq = &S->spH0ph_eph.row[iph - 1];
            for (int j_el = 0; j_el < re->size; j_el++)
                for (int jj = 0; jj < rq->size; jj++) {
                    double val = re->vals[j_el] * rq->vals[jj];
                    int64_t j = re->cols[j_el] + (rq->cols[jj] - 1) * Nel;
                    HV(i) = HV(i) + val * V(j);
                }
        }
}

void orc_spmatvec(orc_sector *S, const double *v, double *Hv)
{
    if (!S->built) die("orc_spmatvec: call orc_buildh first (Hsector NOT set)");
    if (S->p.ed_total_ud) spmatvec_main(S, v, Hv); else spmatvec_orbs(S, v, Hv);
}

typedef struct { double *Hv; double vj; int64_t base; int64_t stride; } app_ctx;
static void emit_apply(void *c, int irow, double val) { app_ctx *x = (app_ctx *)c; x->Hv[x->base + (int64_t)(irow - 1) * x->stride] += val * x->vj; }
typedef struct { orc_sector *S; double *Hv; double vj; int64_t phoff; } nl_app_ctx;
static void emit_nl_apply(void *c, int iup, int idw, double val)
{
    nl_app_ctx *x = (nl_app_ctx *)c;
    int64_t i = iup + (int64_t)(idw - 1) * x->S->DimUp + x->phoff;
    x->Hv[i - 1] += val * x->vj;
}

/* directMatVec_main: ED_HAMILTONIAN_DIRECT_HxV.f90:21-95 (scatter form) */
static void direct_main(orc_sector *S, const double *vin, double *Hv)
{
    const orc_params *p = &S->p;
    int64_t Nloc = S->Dim, DimUp = S->DimUp, DimDw = S->DimDw, Nel = DimUp * DimDw;
    int nup[64], ndw[64], mup, mdw;
    for (int64_t i = 0; i < Nloc; i++) Hv[i] = 0.0;
    /* direct/HxV_local.f90:1-77 */
    for (int64_t i = 1; i <= Nloc; i++) {
        int64_t i_el = (i - 1) % Nel + 1;
        occupations(S, i_el, nup, ndw, NULL, NULL);
        Hv[i - 1] = Hv[i - 1] + h_local(S, nup, ndw) * vin[i - 1];
    }
    /* direct/HxV_up.f90:1-96 */
    for (int64_t iph = 1; iph <= S->DimPh; iph++)
        for (int64_t jdw = 1; jdw <= DimDw; jdw++)
            for (int64_t jup = 1; jup <= DimUp; jup++) {
                int64_t j = jup + (jdw - 1) * DimUp + (iph - 1) * Nel;
                app_ctx c = { Hv, vin[j - 1], (jdw - 1) * DimUp + (iph - 1) * Nel, 1 };
                hops_main(S, 1, S->map[0], (int)DimUp, S->map[0][jup - 1], emit_apply, &c);
            }
    /* direct/HxV_dw.f90:1-95 */
    for (int64_t iph = 1; iph <= S->DimPh; iph++)
        for (int64_t jup = 1; jup <= DimUp; jup++)
            for (int64_t jdw = 1; jdw <= DimDw; jdw++) {
                int64_t j = jup + (jdw - 1) * DimUp + (iph - 1) * Nel;
                app_ctx c = { Hv, vin[j - 1], (jup - 1) + (iph - 1) * Nel, DimUp };
                hops_main(S, p->Nspin, S->map[1], (int)DimDw, S->map[1][jdw - 1], emit_apply, &c);
            }
    if (S->DimPh > 1) {
        /* direct/HxV_ph.f90:2-6 */
        for (int64_t i = 1; i <= Nloc; i++) { int64_t iph = (i - 1) / Nel + 1; Hv[i - 1] = Hv[i - 1] + p->w0_ph * (double)(iph - 1) * vin[i - 1]; }
        /* direct/HxV_eph.f90:2-31 */
        for (int64_t i = 1; i <= Nloc; i++) {
            int64_t i_el = (i - 1) % Nel + 1, iph = (i - 1) / Nel + 1;
            double htmp = 0.0;
            occupations(S, i_el, nup, ndw, NULL, NULL);
            for (int iorb = 1; iorb <= p->Norb; iorb++) htmp = htmp + p->g_ph[iorb - 1] * (nup[iorb - 1] + ndw[iorb - 1] - 1.0);
            for (int64_t jj = 1; jj <= S->DimPh; jj++) {
                if (jj == iph + 1) { int64_t j = i_el + (jj - 1) * Nel; Hv[i - 1] = Hv[i - 1] + htmp * sqrt((double)iph) * vin[j - 1]; }
                if (jj == iph - 1) { int64_t j = i_el + (jj - 1) * Nel; Hv[i - 1] = Hv[i - 1] + htmp * sqrt((double)(iph - 1)) * vin[j - 1]; }
            }
        }
    }
    /* direct/HxV_non_local.f90:1-71 */
    if (S->Jhflag)
        for (int64_t j = 1; j <= Nloc; j++) {
            int64_t j_el = (j - 1) % Nel + 1, iph = (j - 1) / Nel + 1;
            nl_app_ctx c = { S, Hv, vin[j - 1], (iph - 1) * Nel };
            occupations(S, j_el, nup, ndw, &mup, &mdw);
            nonlocal_terms(S, mup, mdw, nup, ndw, emit_nl_apply, &c);
        }
}

typedef struct { orc_sector *S; double *Hv; double vj; int Jnd[2 * ORC_MAX_UD]; int slot; int64_t phoff; } orb_app_ctx;
static void emit_orb_apply(void *c, int irow, double val)
{
    orb_app_ctx *x = (orb_app_ctx *)c; orc_sector *S = x->S;
    int Nvec[2 * ORC_MAX_UD], Ind[2 * ORC_MAX_UD], n2 = 2 * S->Ns_Ud;
    for (int k = 0; k < S->Ns_Ud; k++) { Nvec[k] = S->DimUps[k]; Nvec[k + S->Ns_Ud] = S->DimDws[k]; }
    memcpy(Ind, x->Jnd, sizeof(int) * (size_t)n2); Ind[x->slot] = irow;
    int64_t i = indices2state(Ind, Nvec, n2) + x->phoff;
    x->Hv[i - 1] += val * x->vj;
}

/* directMatVec_orbs: ED_HAMILTONIAN_DIRECT_HxV.f90:100-171 + direct/Orbs/ (all files) */
static void direct_orbs(orc_sector *S, const double *vin, double *Hv)
{
    const orc_params *p = &S->p;
    int64_t Nloc = S->Dim, Nel = S->DimUp * S->DimDw;
    int nup[64], ndw[64], Nvec[2 * ORC_MAX_UD], n2 = 2 * S->Ns_Ud;
    for (int k = 0; k < S->Ns_Ud; k++) { Nvec[k] = S->DimUps[k]; Nvec[k + S->Ns_Ud] = S->DimDws[k]; }
    for (int64_t i = 0; i < Nloc; i++) Hv[i] = 0.0;
    for (int64_t i = 1; i <= Nloc; i++) {
        int64_t i_el = (i - 1) % Nel + 1;
        occupations(S, i_el, nup, ndw, NULL, NULL);
        Hv[i - 1] = Hv[i - 1] + h_local(S, nup, ndw) * vin[i - 1];
    }
    for (int pass = 0; pass < 2; pass++) /* HxV_up then HxV_dw */
        for (int64_t j = 1; j <= Nloc; j++) {
            int64_t i_el = (j - 1) % Nel + 1, iph = (j - 1) / Nel + 1;
            orb_app_ctx c; c.S = S; c.Hv = Hv; c.vj = vin[j - 1]; c.phoff = (iph - 1) * Nel;
            state2indices(i_el, Nvec, n2, c.Jnd);
            for (int iorb = 1; iorb <= S->Ns_Ud; iorb++) {
                int slot = pass == 0 ? iorb - 1 : iorb - 1 + S->Ns_Ud;
                int dim = pass == 0 ? S->DimUps[iorb - 1] : S->DimDws[iorb - 1];
                c.slot = slot;
                hops_orbs(S, pass == 0 ? 1 : p->Nspin, iorb, S->map[slot], dim, S->map[slot][c.Jnd[slot] - 1], emit_orb_apply, &c);
            }
        }
    if (S->DimPh > 1) {
        for (int64_t i = 1; i <= Nloc; i++) { int64_t iph = (i - 1) / Nel + 1; Hv[i - 1] = Hv[i - 1] + p->w0_ph * (double)(iph - 1) * vin[i - 1]; }
        for (int64_t i = 1; i <= Nloc; i++) {
            int64_t i_el = (i - 1) % Nel + 1, iph = (i - 1) / Nel + 1;
            double htmp = 0.0;
            occupations(S, i_el, nup, ndw, NULL, NULL);
            for (int iorb = 1; iorb <= p->Norb; iorb++) htmp = htmp + p->g_ph[iorb - 1] * (nup[iorb - 1] + ndw[iorb - 1] - 1.0);
            for (int64_t jj = 1; jj <= S->DimPh; jj++) {
                if (jj == iph + 1) { int64_t j = i_el + (jj - 1) * Nel; Hv[i - 1] = Hv[i - 1] + htmp * sqrt((double)iph) * vin[j - 1]; }
                if (jj == iph - 1) { int64_t j = i_el + (jj - 1) * Nel; Hv[i - 1] = Hv[i - 1] + htmp * sqrt((double)(iph - 1)) * vin[j - 1]; }
            }
        }
    }
}

void orc_directmatvec(orc_sector *S, const double *vin, double *Hv)
{
    if (S->p.ed_total_ud) direct_main(S, vin, Hv); else direct_orbs(S, vin, Hv);
}

/* ---- single rows of the operator, gather form ---------------------------------------------------------
 * orc_rows: Hv(i) for a list of rows i, evaluated from the bit patterns alone -- no N-sized matrix is built,
 * so it runs at Ns=16 and Ns=18 where orc_buildh cannot (spH0d alone is 2N heap blocks) and 64-bit row numbers
 * are used throughout (the reference's int32 `i` overflows at Ns=18, direct_mpi/HxV_local.f90:2-5).
 * Per row the accumulation order is that of spMatVec_main (ED_HAMILTONIAN_SPARSE_HxV.f90:399-483) /
 * spMatVec_orbs (487-564): spH0d, the spH0dws row, the spH0ups row, phonon terms, the spH0nd row.
 * Row i of spH0ups holds one entry (i, j) per source pattern j with a hop ONTO i (stored/H_up.f90: the outer
 * loop runs over the source jup and inserts at (iup=target, jup)), in ascending j with duplicates summed in
 * statement order (sp_insert).  Here each hop statement of stored/H_up.f90:8-81 is inverted -- target pattern
 * -> the one source pattern it can come from -- and the reference's own c/cdg calls are then replayed on that
 * source, so signs and values are the reference's; no symmetry of H_sigma is assumed. */
typedef struct { int64_t j; double val; } row_ent;
typedef struct { row_ent e[256]; int n; } row_list;
static void rl_add(row_list *L, int64_t j, double val)
{
    for (int k = 0; k < L->n; k++) if (L->e[k].j == j) { L->e[k].val += val; return; }     /* sp_insert: column exists -> sum */
    if (L->n >= 256) die("orc_rows: more than 256 entries in one factor row");
    L->e[L->n].j = j; L->e[L->n].val = val; L->n++;
}
static void rl_sort(row_list *L)      /* storage order of a factor row = ascending source index (stable insertion sort) */
{
    for (int a = 1; a < L->n; a++) { row_ent t = L->e[a]; int b = a; while (b > 0 && L->e[b - 1].j > t.j) { L->e[b] = L->e[b - 1]; b--; } L->e[b] = t; }
}
/* hop statement "amplitude t, c^+_alpha c_beta" arriving at target pattern mt: source = mt with alpha emptied, beta filled */
static void inv_hop(const int *map, int dim, int mt, int alpha, int beta, double t, row_list *L)
{
    int k1, k2; double sg1, sg2;
    if (alpha == beta || t == 0.0) return;
    if (!((mt >> (alpha - 1)) & 1) || ((mt >> (beta - 1)) & 1)) return;
    int src = (mt & ~(1 << (alpha - 1))) | (1 << (beta - 1));
    op_c(beta, src, &k1, &sg1); op_cdg(alpha, k1, &k2, &sg2);          /* as the reference applies them to the source */
    if (k2 != mt) die("orc_rows: inverted hop does not land on the row");
    rl_add(L, binary_search_rec(map, dim, src), t * sg1 * sg2);
}
/* stored/H_up.f90:8-81 / H_dw.f90 read backwards: every entry of row `mt` of H_sigma */
static void row_main(const orc_sector *S, int spin, const int *map, int dim, int mt, row_list *L)
{
    const orc_params *p = &S->p;
    L->n = 0;
    for (int iorb = 1; iorb <= p->Norb; iorb++)
        for (int jorb = 1; jorb <= p->Norb; jorb++) inv_hop(map, dim, mt, iorb, jorb, IMPHLOC(spin, spin, iorb, jorb), L);
    if (p->bath_type == 2)
        for (int kp = 1; kp <= p->Nbath; kp++)
            for (int iorb = 1; iorb <= p->Norb; iorb++)
                for (int jorb = 1; jorb <= p->Norb; jorb++)
                    inv_hop(map, dim, mt, S->getBathStride[iorb - 1][kp - 1], S->getBathStride[jorb - 1][kp - 1], HBATH(spin, spin, iorb, jorb, kp), L);
    for (int iorb = 1; iorb <= p->Norb; iorb++)
        for (int kp = 1; kp <= p->Nbath; kp++) {
            int ialfa = S->getBathStride[iorb - 1][kp - 1];
            inv_hop(map, dim, mt, ialfa, iorb, DIAG_HYBR(spin, iorb, kp), L);
            inv_hop(map, dim, mt, iorb, ialfa, DIAG_HYBR(spin, iorb, kp), L);
        }
    rl_sort(L);
}
/* stored/Orbs/H_up.f90:7-30 read backwards */
static void row_orbs(const orc_sector *S, int spin, int iorb, const int *map, int dim, int mt, row_list *L)
{
    const orc_params *p = &S->p;
    L->n = 0;
    for (int kp = 1; kp <= p->Nbath; kp++) {
        inv_hop(map, dim, mt, 1 + kp, 1, DIAG_HYBR(spin, iorb, kp), L);
        inv_hop(map, dim, mt, 1, 1 + kp, DIAG_HYBR(spin, iorb, kp), L);
    }
    rl_sort(L);
}
typedef struct { const orc_sector *S; row_list *L; } nl_row_ctx;
static void emit_nl_row(void *c, int jup, int jdw, double val) { nl_row_ctx *x = (nl_row_ctx *)c; rl_add(x->L, jup + (int64_t)(jdw - 1) * x->S->DimUp, val); }

/* idx: n global row numbers, 1-based (1..Dim); v: the whole input vector (Dim); out[r] = (H v)(idx[r]) */
void orc_rows(orc_sector *S, int64_t n, const int64_t *idx, const double *v, double *out)
{
    const orc_params *p = &S->p;
    const int64_t DimUp = S->DimUp, Nel = S->DimUp * S->DimDw;
    int Nvec[2 * ORC_MAX_UD], Ind[2 * ORC_MAX_UD], Jnd[2 * ORC_MAX_UD], n2 = 2 * S->Ns_Ud;
    for (int k = 0; k < S->Ns_Ud; k++) { Nvec[k] = S->DimUps[k]; Nvec[k + S->Ns_Ud] = S->DimDws[k]; }
    for (int64_t r = 0; r < n; r++) {
        const int64_t i = idx[r];
        if (i < 1 || i > S->Dim) die("orc_rows: row out of range");
        const int64_t iph = (i - 1) / Nel + 1, i_el = (i - 1) % Nel + 1, phoff = (iph - 1) * Nel;
        int nup[64], ndw[64], mup = 0, mdw = 0;
        row_list L;
        double hv = 0.0;
        occupations(S, i_el, nup, ndw, &mup, &mdw);
        hv = hv + h_local(S, nup, ndw) * V(i);                                            /* spH0d */
        if (p->ed_total_ud) {
            const int64_t iup = iup_index(i_el, DimUp), idw = idw_index(i_el, DimUp);
            row_main(S, p->Nspin, S->map[1], (int)S->DimDw, mdw, &L);                      /* 413-427: dw first */
            for (int k = 0; k < L.n; k++) { int64_t j = iup + (L.e[k].j - 1) * DimUp + phoff; hv = hv + L.e[k].val * V(j); }
            row_main(S, 1, S->map[0], (int)DimUp, mup, &L);                                /* 429-443 */
            for (int k = 0; k < L.n; k++) { int64_t j = L.e[k].j + (idw - 1) * DimUp + phoff; hv = hv + L.e[k].val * V(j); }
        } else {
            state2indices(i_el, Nvec, n2, Ind);
            for (int iud = 0; iud < S->Ns_Ud; iud++) {                                     /* 512-533 */
                row_orbs(S, 1, iud + 1, S->map[iud], S->DimUps[iud], S->map[iud][Ind[iud] - 1], &L);
                for (int k = 0; k < L.n; k++) { memcpy(Jnd, Ind, sizeof(int) * (size_t)n2); Jnd[iud] = (int)L.e[k].j; hv = hv + L.e[k].val * V(indices2state(Jnd, Nvec, n2) + phoff); }
                row_orbs(S, p->Nspin, iud + 1, S->map[iud + S->Ns_Ud], S->DimDws[iud], S->map[iud + S->Ns_Ud][Ind[iud + S->Ns_Ud] - 1], &L);
                for (int k = 0; k < L.n; k++) { memcpy(Jnd, Ind, sizeof(int) * (size_t)n2); Jnd[iud + S->Ns_Ud] = (int)L.e[k].j; hv = hv + L.e[k].val * V(indices2state(Jnd, Nvec, n2) + phoff); }
            }
        }
        if (S->DimPh > 1) {                                                               /* 445-470 / 535-562 */
            double htmp = 0.0;
            hv = hv + p->w0_ph * (iph - 1) * V(i);
            for (int iorb = 1; iorb <= p->Norb; iorb++) htmp = htmp + p->g_ph[iorb - 1] * (nup[iorb - 1] + ndw[iorb - 1] - 1.0);
            if (iph + 1 <= S->DimPh) { double val = htmp * sqrt((double)iph); hv = hv + val * V(i_el + iph * Nel); }
            if (iph - 1 > 0) { double val = htmp * sqrt((double)(iph - 1)); hv = hv + val * V(i_el + (iph - 2) * Nel); }
        }
        if (S->Jhflag) {                                                                  /* 473-483: row i of spH0nd (insertion order) */
            nl_row_ctx c = { S, &L };
            L.n = 0;
            nonlocal_terms(S, mup, mdw, nup, ndw, emit_nl_row, &c);
            for (int k = 0; k < L.n; k++) hv = hv + L.e[k].val * V(L.e[k].j + phoff);
        }
        out[r] = hv;
    }
}

/* dense Hmat of ed_buildh_main/orbs: ED_HAMILTONIAN_SPARSE_HxV.f90:132-195, 304-363.
 * Column-major Dim x Dim. */
void orc_hmat(orc_sector *S, double *Hmat)
{
    int64_t Dim = S->Dim, Nel = S->DimUp * S->DimDw, DimUp = S->DimUp;
    if (!S->built) die("orc_hmat: call orc_buildh first");
#define HM(i, j) Hmat[((i)-1) + (size_t)Dim * ((j)-1)]
    memset(Hmat, 0, sizeof(double) * (size_t)Dim * (size_t)Dim);
    for (int64_t iph = 1; iph <= S->DimPh; iph++) {
        int64_t off = (iph - 1) * Nel;
        for (int64_t i = 1; i <= Nel; i++) {
            orc_row *r = &S->spH0d.row[i - 1];
            for (int k = 0; k < r->size; k++) HM(i + off, r->cols[k] + off) += r->vals[k];
            if (S->Jhflag) { r = &S->spH0nd.row[i - 1]; for (int k = 0; k < r->size; k++) HM(i + off, r->cols[k] + off) += r->vals[k]; }
        }
        if (S->p.ed_total_ud) {
            for (int64_t idw = 1; idw <= S->DimDw; idw++)
                for (int64_t iup = 1; iup <= DimUp; iup++) {
                    int64_t i = iup + (idw - 1) * DimUp;
                    orc_row *r = &S->spH0dws[0].row[idw - 1];
                    for (int k = 0; k < r->size; k++) HM(i + off, iup + (r->cols[k] - 1) * DimUp + off) += r->vals[k];
                    r = &S->spH0ups[0].row[iup - 1];
                    for (int k = 0; k < r->size; k++) HM(i + off, r->cols[k] + (idw - 1) * DimUp + off) += r->vals[k];
                }
        } else {
            int Nvec[2 * ORC_MAX_UD], Ind[2 * ORC_MAX_UD], Jnd[2 * ORC_MAX_UD], n2 = 2 * S->Ns_Ud;
            for (int k = 0; k < S->Ns_Ud; k++) { Nvec[k] = S->DimUps[k]; Nvec[k + S->Ns_Ud] = S->DimDws[k]; }
            for (int64_t i = 1; i <= Nel; i++) {
                state2indices(i, Nvec, n2, Ind);
                for (int iud = 0; iud < S->Ns_Ud; iud++) {
                    orc_row *r = &S->spH0ups[iud].row[Ind[iud] - 1];
                    for (int k = 0; k < r->size; k++) { memcpy(Jnd, Ind, sizeof(int) * (size_t)n2); Jnd[iud] = (int)r->cols[k]; HM(i + off, indices2state(Jnd, Nvec, n2) + off) += r->vals[k]; }
                    r = &S->spH0dws[iud].row[Ind[iud + S->Ns_Ud] - 1];
                    for (int k = 0; k < r->size; k++) { memcpy(Jnd, Ind, sizeof(int) * (size_t)n2); Jnd[iud + S->Ns_Ud] = (int)r->cols[k]; HM(i + off, indices2state(Jnd, Nvec, n2) + off) += r->vals[k]; }
                }
            }
        }
    }
    if (S->DimPh > 1)
        for (int64_t iph = 1; iph <= S->DimPh; iph++)
            for (int64_t i = 1; i <= Nel; i++) {
                orc_row *rp = &S->spH0_ph.row[iph - 1];
                for (int k = 0; k < rp->size; k++) HM(i + (iph - 1) * Nel, i + (rp->cols[k] - 1) * Nel) += rp->vals[k];
                orc_row *re = &S->spH0e_eph.row[i - 1], *rq = &S->spH0ph_eph.row[iph - 1];
                for (int a = 0; a < re->size; a++)
                    for (int b = 0; b < rq->size; b++)
                        HM(i + (iph - 1) * Nel, re->cols[a] + (rq->cols[b] - 1) * Nel) += re->vals[a] * rq->vals[b];
            }
}
