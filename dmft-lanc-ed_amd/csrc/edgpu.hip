// edgpu.hip -- libedgpu.so: sector operator + Lanczos loop on MI355X behind the C ABI of
// include/edgpu.h.  Host orchestration here, device code in kernels.hpp, sector/factor
// construction in sector_host.hpp.  gfx950 only; no CPU compute path.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../../include/edgpu.h"
#include "sector_host.hpp"
#include "kernels.hpp"

namespace edgpu {

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            int c_ = (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? EDGPU_ENODEVICE         \
                     : (e_ == hipErrorOutOfMemory ? EDGPU_ENOMEM : EDGPU_EHIP);                        \
            return fail(c_, std::string(#expr) + ": " + hipGetErrorString(e_));                       \
        }                                                                                              \
    } while (0)
#define CHK(expr) do { int r_ = (expr); if (r_ != EDGPU_OK) return r_; } while (0)

static int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) return fail(EDGPU_ENODEVICE, "edgpu: no HIP device visible (this library has no CPU fallback)");
    return EDGPU_OK;
}

// ------------------------------------------------------------------------------------------
// RCCL, resolved at run time so that the library loads on hosts without it and shares the
// copy an embedding process (e.g. PyTorch) may already have loaded.
// ------------------------------------------------------------------------------------------
struct NcclUid { char b[128]; };
struct Rccl {
    int (*GetUniqueId)(NcclUid *) = nullptr;
    int (*CommInitRank)(void **, int, NcclUid, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    bool load()
    {
        if (ok) return true;
        // Use the RCCL the process already has (an embedding PyTorch ships its own copy, built against the
        // HIP runtime it loaded); only then fall back to the system library.
        void *h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclCommInitRank")) {
            h = nullptr;
            if (const char *e = getenv("EDGPU_RCCL_LIB")) h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
            if (!h) {
                if (FILE *f = fopen("/proc/self/maps", "r")) {
                    char line[4096];
                    while (!h && fgets(line, sizeof(line), f)) {
                        char *p = strstr(line, "librccl");
                        if (!p) continue;
                        char *path = strchr(line, '/');
                        if (!path) continue;
                        path[strcspn(path, "\n")] = 0;
                        h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
                    }
                    fclose(f);
                }
            }
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) return false;
        }
#define L(name) *(void **)(&name) = dlsym(h, "nccl" #name); if (!name) return false
        L(GetUniqueId); L(CommInitRank); L(CommDestroy); L(Send); L(Recv); L(AllReduce); L(GroupStart); L(GroupEnd); L(GetErrorString);
#undef L
        ok = true;
        return true;
    }
};
static Rccl g_rccl;
constexpr int kNcclDouble = 8, kNcclSum = 0;

struct Comm {
    enum Kind { NONE, RCCL, CALLBACK } kind = NONE;
    int rank = 0, size = 1;
    void *nccl = nullptr;
    edgpu_allgatherv_fn ag = nullptr; edgpu_exchange_fn xch = nullptr; edgpu_allreduce_fn ar = nullptr; void *ctx = nullptr;
};

template <class T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    int alloc(size_t count) { release(); n = count; if (!count) return EDGPU_OK; HIPCHK(hipMalloc((void **)&p, count * sizeof(T))); return EDGPU_OK; }
    int upload(const std::vector<T> &h) { CHK(alloc(h.size())); if (h.size()) HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return EDGPU_OK; }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    ~DevBuf() { release(); }
};

struct DevFactor {
    DevBuf<int32_t> map, rank, rowptr, cols, ellc, talpha, tbeta;
    DevBuf<uint32_t> pcols, pcodes, pelem, pelem8; DevBuf<double> ptval; bool packed = false, packed_elem = false, packed_elem8 = false; int32_t pterms = 0;
    DevBuf<double> vals, ellv, E, tval, lin;
    int32_t width = 0;
};

}  // namespace edgpu

using namespace edgpu;

struct edgpu_sector_s {
    SectorHost H;
    Comm comm;
    int device = 0;
    hipStream_t stream = nullptr, cstream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_gather = nullptr, ev_ready = nullptr;
    std::vector<std::unique_ptr<DevFactor>> fac;
    DevBuf<double> diag, cross, gup, gdw, bd_up, bd_dw;
    DevBuf<int32_t> stride;
    // explicit CSR
    DevBuf<int64_t> csr_rowptr, uppre, dwpre; DevBuf<int32_t> csr_cols; DevBuf<double> csr_vals;
    DevBuf<int64_t> csrD_rowptr; DevBuf<int32_t> csrD_cols; DevBuf<double> csrD_vals;   // dw block of the split form
    bool csr_split = false; int64_t csrL_nnz = 0, csrD_nnz = 0; int32_t csrD_rpb = 64, csrD_bpr = 0; int csrD_nt = 1; bool csrL_small = false, csrD_small = false;
    int64_t csr_nnz = 0; int32_t csr_rpb = 0, csr_bpr = 0, csr_tile = 1;
    // work vectors
    // halo exchange (multi-rank, ed_total_ud=T without Jx/Jp/phonon extras): x-space of the dw hops is
    // [own rows | rows needed from rank 0 | rank 1 | ...] instead of the whole gathered vector
    bool halo = false; int64_t nhalo = 0;
    std::vector<int64_t> send_rows_cnt, recv_rows_cnt, send_rows_off, recv_rows_off;
    DevBuf<int32_t> send_list, dw_ellc_x, dw_cols_x; DevBuf<double> sendbuf;
    hipEvent_t ev_pack = nullptr;
    std::vector<std::unique_ptr<DevBuf<double>>> eigV; DevBuf<double> eig_h, eig_y, eig_partial;
    DevBuf<double> vfull, tmp_in, tmp_out, lz[5], scal, alphas, betas, zvec, partial, partial2;
    int64_t nloc = 0, nloc_el = 0, vecdim = 0;
    int32_t nmax = 0;
    bool use_lds = false; int kron_threads = 256, kron_threads_override = 0; size_t kron_lds = 0;
    bool two_factor = false, extra = false, multi = false, kron_split = false; int dw_w = 64, dw_variant = 2, dw_rows = 64; bool up_persistent = true, up_pairs = true; int up_grid = 256, up_threads = 1024, row_split = 4; bool up_big = true;
    std::vector<hipEvent_t> bench_ev; int bench_k = 0; double *bv[3] = {nullptr, nullptr, nullptr};
    int64_t nnz_total = 0, bytes_stored = 0;
    int64_t npartial = 0;
};

namespace edgpu {
using S = edgpu_sector_s;

static int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static int dbg_flags() { const char *e = getenv("EDGPU_DEBUG"); return e ? atoi(e) : 0; }
static int dbg_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }

// ---------------------------------------------------------------------------------------
// device-side construction
// ---------------------------------------------------------------------------------------
static int upload_factor(S *s, int k)
{
    const Factor &f = s->H.fac[k];
    auto df = std::make_unique<DevFactor>();
    CHK(df->map.upload(f.map)); CHK(df->rank.upload(f.rank));
    std::vector<int32_t> rp32(f.rowptr.begin(), f.rowptr.end());
    CHK(df->rowptr.upload(rp32)); CHK(df->cols.upload(f.cols)); CHK(df->vals.upload(f.vals));
    CHK(df->E.upload(f.E));
    df->width = std::max(f.width, 1);
    if (df->width > 64) return fail(EDGPU_EUNSUPPORTED, "edgpu: more than 64 hops per spin configuration");
    if ((int)f.terms.size() > kMaxTerms) return fail(EDGPU_EUNSUPPORTED, "edgpu: too many one-body hop terms");
    // ELL image: up factors transposed [width][dim] (lanes run over rows), dw factors [dim][width]
    const bool transposed = k < s->H.Ns_Ud;
    std::vector<int32_t> ec((size_t)df->width * f.dim); std::vector<double> ev((size_t)df->width * f.dim);
    for (int32_t i = 0; i < f.dim; i++)
        for (int32_t q = 0; q < df->width; q++) {
            int64_t p = f.rowptr[i] + q; bool has = p < f.rowptr[i + 1];
            size_t o = transposed ? (size_t)q * f.dim + i : (size_t)i * df->width + q;
            ec[o] = has ? f.cols[p] : i; ev[o] = has ? f.vals[p] : 0.0;
        }
    CHK(df->ellc.upload(ec)); CHK(df->ellv.upload(ev));
    // packed image for the persistent local pass: 16-bit column + (sign | term index) byte per hop;
    // padded slots use code 127 -> amplitude 0.  Only when every entry is a single hop term.
    if (transposed && f.dim <= 65536 && f.terms.size() < 127) {
        const int W = df->width, ncw = (W + 1) / 2, ndw = (W + 3) / 4;
        std::vector<uint32_t> pc((size_t)ncw * f.dim, 0u), pd((size_t)ndw * f.dim, 0u);
        std::vector<double> tvl(128, 0.0);
        for (size_t t = 0; t < f.terms.size(); t++) tvl[t] = std::fabs(f.terms[t].t);
        bool ok = true;
        for (int32_t i = 0; i < f.dim && ok; i++)
            for (int32_t q = 0; q < 4 * ndw; q++) {
                int64_t p = f.rowptr[i] + q; bool has = q < W && p < f.rowptr[i + 1];
                uint32_t col = has ? (uint32_t)f.cols[p] : (uint32_t)i, code = 127u;
                if (has) {
                    int tid = -1;
                    for (size_t t = 0; t < f.terms.size(); t++) if (std::fabs(f.terms[t].t) == std::fabs(f.vals[p])) { tid = (int)t; break; }
                    if (tid < 0) { ok = false; break; }     // summed duplicates: keep the wide image
                    code = (uint32_t)tid | (f.vals[p] < 0.0 ? 0x80u : 0u);
                }
                if (q < 2 * ncw) pc[(size_t)(q / 2) * f.dim + i] |= col << (16 * (q & 1));
                pd[(size_t)(q / 4) * f.dim + i] |= code << (8 * (q & 3));
            }
        if (ok) { CHK(df->pcols.upload(pc)); CHK(df->pcodes.upload(pd)); CHK(df->ptval.upload(tvl)); df->packed = true; df->pterms = 128; }
        if (ok && W <= 10) {                        // element-major image of 8 dwords for rows that do not fit LDS
            std::vector<uint32_t> pe((size_t)8 * f.dim, 0u);
            for (int32_t i = 0; i < f.dim; i++) {
                for (int q = 0; q < 10; q++) {
                    int64_t p = f.rowptr[i] + q; bool has = q < W && p < f.rowptr[i + 1];
                    uint32_t col = has ? (uint32_t)f.cols[p] : (uint32_t)i;
                    pe[(size_t)8 * i + q / 2] |= col << (16 * (q & 1));
                }
                for (int q = 0; q < 12; q++) {
                    int64_t p = f.rowptr[i] + q; bool has = q < W && p < f.rowptr[i + 1];
                    uint32_t code = 127u;
                    if (has) { code = (pd[(size_t)(q / 4) * f.dim + i] >> (8 * (q & 3))) & 0xffu; }
                    pe[(size_t)8 * i + 5 + q / 4] |= code << (8 * (q & 3));
                }
            }
            CHK(df->pelem8.upload(pe)); df->packed_elem8 = true;
        }
        if (ok && W <= 8 && f.dim % 2 == 0) {       // element-major image for the pair kernel: 6 dwords per up configuration
            std::vector<uint32_t> pe((size_t)6 * f.dim, 0u);
            for (int32_t i = 0; i < f.dim; i++) {
                for (int c = 0; c < 4; c++) pe[(size_t)6 * i + c] = c < ncw ? pc[(size_t)c * f.dim + i] : ((uint32_t)i | ((uint32_t)i << 16));
                for (int c = 0; c < 2; c++) pe[(size_t)6 * i + 4 + c] = c < ndw ? pd[(size_t)c * f.dim + i] : 0x7f7f7f7fu;
            }
            CHK(df->pelem.upload(pe)); df->packed_elem = true;
        }
    }
    std::vector<int32_t> ta, tb; std::vector<double> tv;
    for (auto &t : f.terms) { ta.push_back(t.alpha); tb.push_back(t.beta); tv.push_back(t.t); }
    if (ta.empty()) { ta.push_back(1); tb.push_back(1); tv.push_back(0.0); }
    CHK(df->talpha.upload(ta)); CHK(df->tbeta.upload(tb)); CHK(df->tval.upload(tv));
    // per-level one-body coefficients for the on-the-fly energy table
    {
        const edgpu_desc &d = s->H.desc; const SectorHost &H = s->H;
        int spin = k < H.Ns_Ud ? 1 : d.Nspin;
        std::vector<double> lin(f.nbits, 0.0);
        auto hf = [&](int a) { return d.hfmode ? -0.5 * d.Uloc[a - 1] - (d.Norb - 1) * (0.5 * d.Ust + 0.5 * (d.Ust - d.Jh)) : 0.0; };
        if (d.ed_total_ud) {
            for (int a = 1; a <= d.Norb; a++) lin[a - 1] += H.imp(spin, spin, a, a) - d.xmu + hf(a);
            for (int a = 1; a <= d.Nfoo; a++) for (int kp = 1; kp <= d.Nbath; kp++) lin[H.stride[a - 1][kp - 1] - 1] += H.bdiag(spin, a, kp);
        } else {
            int a = (k % H.Ns_Ud) + 1;
            lin[0] += H.imp(spin, spin, a, a) - d.xmu + hf(a);
            for (int kp = 1; kp <= d.Nbath; kp++) lin[kp] += H.bdiag(spin, a, kp);
        }
        CHK(df->lin.upload(lin));
    }
    s->fac.push_back(std::move(df));
    return EDGPU_OK;
}

// Which dw rows does each rank need from each other rank?  Row j of rank r is needed here iff some
// local row i has H_dw[i,j] != 0.  H_dw is small and known to every rank, so the lists are
// computed locally on both sides (no set-up communication).  At Ns=16 this is 53 % / 36 % / 24 %
// of the all-gather volume at 2 / 4 / 8 ranks.
static int setup_halo(S *s)
{
    const SectorHost &H = s->H; const Factor &dw = H.fac[1];
    const int P = s->comm.size, me = s->comm.rank;
    auto range_of = [&](int r, int64_t &d0, int64_t &qd) { int64_t q = H.DimDw / P, rem = H.DimDw % P; qd = r < rem ? q + 1 : q; d0 = r < rem ? r * (q + 1) : r * q + rem; };
    std::vector<int> owner(H.DimDw);
    for (int r = 0; r < P; r++) { int64_t d0, qd; range_of(r, d0, qd); for (int64_t j = d0; j < d0 + qd; j++) owner[j] = r; }
    std::vector<std::vector<int32_t>> recv_rows(P), send_rows(P);
    {
        std::vector<char> mark(H.DimDw, 0);
        for (int64_t i = H.dw0; i < H.dw0 + H.Qdw; i++)
            for (int64_t p = dw.rowptr[i]; p < dw.rowptr[i + 1]; p++) mark[dw.cols[p]] = 1;
        for (int64_t j = 0; j < H.DimDw; j++) if (mark[j] && owner[j] != me) recv_rows[owner[j]].push_back((int32_t)j);
        // what rank r needs from me: rows j of mine that some row i of r couples to
        for (int r = 0; r < P; r++) {
            if (r == me) continue;
            int64_t d0, qd; range_of(r, d0, qd);
            std::fill(mark.begin(), mark.end(), 0);
            for (int64_t i = d0; i < d0 + qd; i++)
                for (int64_t p = dw.rowptr[i]; p < dw.rowptr[i + 1]; p++) mark[dw.cols[p]] = 1;
            for (int64_t j = H.dw0; j < H.dw0 + H.Qdw; j++) if (mark[j]) send_rows[r].push_back((int32_t)(j - H.dw0));
        }
    }
    s->send_rows_cnt.assign(P, 0); s->recv_rows_cnt.assign(P, 0); s->send_rows_off.assign(P, 0); s->recv_rows_off.assign(P, 0);
    std::vector<int32_t> send_list, xmap(H.DimDw, -1);
    int64_t so = 0, ro = 0;
    for (int64_t q = 0; q < H.Qdw; q++) xmap[H.dw0 + q] = (int32_t)q;
    for (int r = 0; r < P; r++) {
        s->send_rows_cnt[r] = (int64_t)send_rows[r].size(); s->send_rows_off[r] = so; so += s->send_rows_cnt[r];
        s->recv_rows_cnt[r] = (int64_t)recv_rows[r].size(); s->recv_rows_off[r] = ro;
        for (size_t k = 0; k < recv_rows[r].size(); k++) xmap[recv_rows[r][k]] = (int32_t)(H.Qdw + ro + (int64_t)k);
        ro += s->recv_rows_cnt[r];
        send_list.insert(send_list.end(), send_rows[r].begin(), send_rows[r].end());
    }
    s->nhalo = H.Qdw + ro;
    if (send_list.empty()) send_list.push_back(0);
    CHK(s->send_list.upload(send_list));
    CHK(s->sendbuf.alloc((size_t)std::max<int64_t>(so, 1) * H.DimUp * H.DimPh));
    // H_dw column indices translated to x-space for the local rows (ELL image and CSR image)
    const int W = s->fac[1]->width;
    std::vector<int32_t> ec((size_t)W * dw.dim, 0), cc(dw.cols.size(), 0);
    for (int64_t i = H.dw0; i < H.dw0 + H.Qdw; i++) {
        for (int q = 0; q < W; q++) { int64_t p = dw.rowptr[i] + q; bool has = p < dw.rowptr[i + 1]; ec[(size_t)i * W + q] = has ? xmap[dw.cols[p]] : xmap[i]; }
        for (int64_t p = dw.rowptr[i]; p < dw.rowptr[i + 1]; p++) cc[p] = xmap[dw.cols[p]];
    }
    CHK(s->dw_ellc_x.upload(ec)); CHK(s->dw_cols_x.upload(cc));
    HIPCHK(hipEventCreateWithFlags(&s->ev_pack, hipEventDisableTiming));
    s->halo = true;
    return EDGPU_OK;
}

static int build_diag(S *s)
{
    const SectorHost &H = s->H; const edgpu_desc &d = H.desc;
    DiagArgs A{};
    A.Norb = d.Norb; A.Nspin = d.Nspin; A.Nbath = d.Nbath; A.Nfoo = d.Nfoo; A.hfmode = d.hfmode; A.total_ud = d.ed_total_ud; A.Ns_Ud = H.Ns_Ud;
    for (int k = 0; k < 2 * H.Ns_Ud; k++) { A.dims[k] = H.fac[k].dim; A.maps[k] = s->fac[k]->map.p; }
    A.nloc = s->nloc_el; A.shift = H.dw0 * H.DimUp;
    for (int a = 0; a < EDGPU_MAX_ORB; a++) { A.Uloc[a] = d.Uloc[a]; A.imp_up[a] = A.imp_dw[a] = 0.0; }
    for (int a = 1; a <= d.Norb; a++) { A.imp_up[a - 1] = H.imp(1, 1, a, a); A.imp_dw[a - 1] = H.imp(d.Nspin, d.Nspin, a, a); }
    A.Ust = d.Ust; A.Jh = d.Jh; A.xmu = d.xmu;
    A.bd_up = s->bd_up.p; A.bd_dw = s->bd_dw.p; A.stride = s->stride.p;
    CHK(s->diag.alloc((size_t)s->nloc_el));
    hipLaunchKernelGGL(k_build_diag, dim3((unsigned)cdiv(s->nloc_el, 256)), dim3(256), 0, s->stream, A, s->diag.p);
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}

static int build_csr(S *s)
{
    const SectorHost &H = s->H;
    if (!s->two_factor || H.DimPh > 1 || H.Jhflag) return fail(EDGPU_EUNSUPPORTED, "edgpu: storage=CSR supports ed_total_ud=T without phonons and Jx/Jp");
    if (H.DimUp * s->nhalo >= ((int64_t)1 << 31)) return fail(EDGPU_EUNSUPPORTED, "edgpu: storage=CSR needs DimUp*(rows of x) < 2^31 (int32 columns)");
    const Factor &up = H.fac[0], &dw = H.fac[1];
    std::vector<int64_t> uppre(up.dim + 1, 0), dwpre(H.Qdw + 1, 0);
    for (int i = 0; i < up.dim; i++) uppre[i + 1] = uppre[i] + 1 + (up.rowptr[i + 1] - up.rowptr[i]);
    int64_t maxrow = 0; int upmax = 0;
    for (int i = 0; i < up.dim; i++) upmax = std::max<int>(upmax, (int)(up.rowptr[i + 1] - up.rowptr[i]));
    for (int64_t q = 0; q < H.Qdw; q++) {
        int64_t nd = dw.rowptr[H.dw0 + q + 1] - dw.rowptr[H.dw0 + q];
        dwpre[q + 1] = dwpre[q] + uppre[up.dim] + nd * up.dim;
        maxrow = std::max<int64_t>(maxrow, 1 + upmax + nd);
    }
    s->csr_nnz = dwpre[H.Qdw];
    CHK(s->uppre.upload(uppre)); CHK(s->dwpre.upload(dwpre));
    // Two storage forms of the same explicit matrix: one CSR, or (default for large sectors) two CSR
    // blocks H = H_loc + H_dw so that each block can be streamed in the order that keeps ITS gathers
    // cache resident (rows for the up hops, XCD-owned column tiles for the dw hops).
    s->csr_split = dbg_int("EDGPU_CSR_SPLIT", (H.DimUp * H.DimDw * 8 > ((int64_t)48 << 20)) ? 1 : 0) != 0;
    s->csrL_nnz = H.Qdw * uppre[up.dim]; s->csrD_nnz = s->csr_nnz - s->csrL_nnz;
    CsrBuildArgs A{};
    A.DimUp = (int32_t)H.DimUp; A.DimDw = (int32_t)H.DimDw; A.Qdw = H.Qdw; A.dw0 = H.dw0;
    A.uprp = s->fac[0]->rowptr.p; A.upc = s->fac[0]->cols.p; A.upv = s->fac[0]->vals.p;
    A.dwrp = s->fac[1]->rowptr.p; A.dwc = s->halo ? s->dw_cols_x.p : s->fac[1]->cols.p; A.dwv = s->fac[1]->vals.p;
    A.rowshift = s->halo ? H.dw0 : 0;     // x-space index of local row idw is idw - rowshift
    A.uppre = s->uppre.p; A.dwpre = s->dwpre.p; A.diag = s->diag.p;
    const unsigned gb = (unsigned)cdiv(s->nloc_el + 1, 256);
    if (!s->csr_split) {
        CHK(s->csr_rowptr.alloc((size_t)s->nloc_el + 1)); CHK(s->csr_cols.alloc((size_t)s->csr_nnz)); CHK(s->csr_vals.alloc((size_t)s->csr_nnz));
        A.rowptr = s->csr_rowptr.p; A.cols = s->csr_cols.p; A.vals = s->csr_vals.p;
        hipLaunchKernelGGL(k_csr_build, dim3(gb), dim3(256), 0, s->stream, A, 0);
    } else {
        CHK(s->csr_rowptr.alloc((size_t)s->nloc_el + 1)); CHK(s->csr_cols.alloc((size_t)s->csrL_nnz)); CHK(s->csr_vals.alloc((size_t)s->csrL_nnz));
        CHK(s->csrD_rowptr.alloc((size_t)s->nloc_el + 1)); CHK(s->csrD_cols.alloc((size_t)s->csrD_nnz)); CHK(s->csrD_vals.alloc((size_t)s->csrD_nnz));
        A.rowptr = s->csr_rowptr.p; A.cols = s->csr_cols.p; A.vals = s->csr_vals.p;
        hipLaunchKernelGGL(k_csr_build, dim3(gb), dim3(256), 0, s->stream, A, 1);
        A.rowptr = s->csrD_rowptr.p; A.cols = s->csrD_cols.p; A.vals = s->csrD_vals.p;
        hipLaunchKernelGGL(k_csr_build, dim3(gb), dim3(256), 0, s->stream, A, 2);
        int64_t ndmax = 0;
        for (int64_t q = 0; q < H.Qdw; q++) ndmax = std::max<int64_t>(ndmax, dw.rowptr[H.dw0 + q + 1] - dw.rowptr[H.dw0 + q]);
        s->csrD_rpb = (int32_t)std::min<int64_t>(dbg_int("EDGPU_CSRD_ROWS", 64), 1152 / std::max<int64_t>(ndmax, 1));
        if (s->csrD_rpb < 1) return fail(EDGPU_EUNSUPPORTED, "edgpu: a CSR row exceeds the LDS tile");
        s->csrD_bpr = (int32_t)cdiv(H.DimUp, s->csrD_rpb);
        s->csrD_nt = dbg_int("EDGPU_CSRD_NT", 1);
        maxrow = 1 + upmax;
        s->csrL_small = (1 + upmax) * 64 <= 640 && dbg_int("EDGPU_CSR_SMALL", 1);     // smaller LDS image -> more resident workgroups
        s->csrD_small = ndmax * s->csrD_rpb <= 640 && dbg_int("EDGPU_CSR_SMALL", 1);
    }
    HIPCHK(hipGetLastError());
    s->csr_rpb = (int32_t)std::min<int64_t>(64, 1152 / std::max<int64_t>(maxrow, 1));      // one wave per row block
    if (s->csr_rpb < 1) return fail(EDGPU_EUNSUPPORTED, "edgpu: a CSR row exceeds the LDS tile");
    s->csr_bpr = (int32_t)cdiv(H.DimUp, s->csr_rpb);
    s->csr_tile = std::max(1, std::min(dbg_int("EDGPU_CSR_TILE", 1 << 20), s->csr_bpr));
    return EDGPU_OK;
}

static int64_t max_partials(const S *s)
{
    int64_t n = std::max<int64_t>(s->H.Qdw * s->H.DimPh, 4096);
    if (s->two_factor) {   // column-tiled dw pass: one partial per workgroup
        const int w = s->dw_w, R = std::min(256 / w, s->dw_rows);
        n = std::max<int64_t>(n, 8 * cdiv(cdiv(s->H.DimUp, w), 8) * cdiv(s->H.Qdw * s->H.DimPh, std::max(R, 1)));
    }
    if (s->csr_bpr) n = std::max<int64_t>(n, (int64_t)s->csr_bpr * s->H.Qdw);
    if (s->csrD_bpr) n = std::max<int64_t>(n, 8 * cdiv(s->csrD_bpr, 8) * s->H.Qdw);
    return n;
}

static int create(const edgpu_desc *desc, S **out)
{
    CHK(require_device());
    auto s = std::unique_ptr<S>(new S());
    std::string e = s->H.init(*desc);
    if (!e.empty()) return fail(EDGPU_EINVAL, e);
    if (desc->storage < EDGPU_STORE_KRON || desc->storage > EDGPU_STORE_DIRECT) return fail(EDGPU_EINVAL, "edgpu: bad storage");
    if (desc->device >= 0) HIPCHK(hipSetDevice(desc->device));
    HIPCHK(hipGetDevice(&s->device));
    HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&s->cstream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&s->ev_a)); HIPCHK(hipEventCreate(&s->ev_b));
    HIPCHK(hipEventCreateWithFlags(&s->ev_gather, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
    SectorHost &H = s->H;
    s->comm.rank = desc->mpi_rank; s->comm.size = desc->mpi_size;
    s->two_factor = desc->ed_total_ud != 0;
    s->extra = H.DimPh > 1 || H.Jhflag;
    s->nloc_el = H.DimUp * H.Qdw; s->nloc = s->nloc_el * H.DimPh; s->vecdim = s->nloc;
    for (int k = 0; k < 2 * H.Ns_Ud; k++) CHK(upload_factor(s.get(), k));
    CHK(s->cross.upload(H.cross)); CHK(s->gup.upload(H.gph_up)); CHK(s->gdw.upload(H.gph_dw));
    {
        std::vector<double> bu((size_t)desc->Nfoo * desc->Nbath), bd((size_t)desc->Nfoo * desc->Nbath);
        for (int a = 1; a <= desc->Nfoo; a++) for (int k = 1; k <= desc->Nbath; k++) { bu[(a - 1) + desc->Nfoo * (k - 1)] = H.bdiag(1, a, k); bd[(a - 1) + desc->Nfoo * (k - 1)] = H.bdiag(desc->Nspin, a, k); }
        std::vector<int32_t> st((size_t)desc->Norb * desc->Nbath);
        for (int a = 1; a <= desc->Norb; a++) for (int k = 1; k <= desc->Nbath; k++) st[(a - 1) + desc->Norb * (k - 1)] = H.stride[a - 1][k - 1];
        CHK(s->bd_up.upload(bu)); CHK(s->bd_dw.upload(bd)); CHK(s->stride.upload(st));
    }
    // launch geometry of the fused two-factor kernel
    if (s->two_factor) {
        size_t rowbytes = (size_t)H.DimUp * sizeof(double);
        s->use_lds = rowbytes + 128 <= 150 * 1024 && !(dbg_flags() & 4);
        if (dbg_int("EDGPU_KRON_THREADS", 0)) s->kron_threads_override = dbg_int("EDGPU_KRON_THREADS", 0);
        s->kron_threads = H.DimUp >= 4096 ? 1024 : (H.DimUp >= 1024 ? 512 : 256);
        if (s->kron_threads_override) s->kron_threads = s->kron_threads_override;
        s->kron_lds = (s->use_lds ? rowbytes : 0) + 16 * sizeof(double);
        // two-pass product (local pass + column-tiled dw pass) once the vector no longer fits the caches
        s->kron_split = dbg_int("EDGPU_KRON_SPLIT", (H.DimUp * H.DimDw * 8 > ((int64_t)48 << 20)) ? 1 : 0) != 0;
        // tile width of the dw pass: 64 columns measured best at Ns=16 (slab 6.6 MB, half of the gathers hit
        // L2) and at Ns=18 (slab 25 MB: no re-use left, but narrower tiles lose more to short segments)
        s->dw_w = dbg_int("EDGPU_DW_W", 64);
        s->dw_variant = dbg_int("EDGPU_DW_VARIANT", 2);
        s->up_persistent = dbg_int("EDGPU_UP_PERSISTENT", 1) != 0;
        s->up_grid = dbg_int("EDGPU_UP_GRID", 256);
        s->up_pairs = dbg_int("EDGPU_UP_PAIRS", 1) != 0;
        s->row_split = std::max(1, std::min(32, dbg_int("EDGPU_ROW_SPLIT", 4)));
        s->up_big = dbg_int("EDGPU_UP_BIG", 1) != 0;
        s->up_threads = dbg_int("EDGPU_UP_THREADS", 1024) == 512 ? 512 : 1024;
        s->dw_rows = std::max(1, std::min(64, dbg_int("EDGPU_DW_ROWS", 64)));
        if (s->dw_w != 8 && s->dw_w != 16 && s->dw_w != 32 && s->dw_w != 64 && s->dw_w != 128 && s->dw_w != 256) s->dw_w = 32;
    }
    // operator storage
    int64_t nnz_fac = 0;
    if (s->two_factor) {
        const Factor &up = H.fac[0], &dw = H.fac[1];
        int64_t nd = dw.rowptr[H.dw0 + H.Qdw] - dw.rowptr[H.dw0];
        nnz_fac = (int64_t)up.cols.size() * H.Qdw + nd * H.DimUp;
    } else {
        for (int k = 0; k < 2 * H.Ns_Ud; k++) nnz_fac += (int64_t)H.fac[k].cols.size() * (s->nloc_el / H.fac[k].dim);
    }
    s->nnz_total = (s->nloc_el + nnz_fac) * H.DimPh;
    s->multi = desc->mpi_size > 1 || (desc->reserved & 1);   // reserved bit 0: run the exchange path even on one rank (tests)
    s->nhalo = H.DimDw;
    if (s->multi && s->two_factor && !s->extra && dbg_int("EDGPU_HALO", 1)) CHK(setup_halo(s.get()));
    if (desc->storage != EDGPU_STORE_DIRECT) { CHK(build_diag(s.get())); s->bytes_stored = s->nloc_el * 8; }
    if (desc->storage == EDGPU_STORE_CSR) {
        CHK(build_csr(s.get()));
        s->bytes_stored = s->csr_nnz * 12 + (s->nloc_el + 1) * 8 * (s->csr_split ? 2 : 1);
    }
    if (s->multi) CHK(s->vfull.alloc((size_t)(H.DimUp * s->nhalo * H.DimPh)));
    s->npartial = max_partials(s.get());
    CHK(s->partial.alloc((size_t)s->npartial));
    CHK(s->partial2.alloc(1024));
    CHK(s->scal.alloc(SC_COUNT));
    HIPCHK(hipMemsetAsync(s->scal.p, 0, SC_COUNT * sizeof(double), s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    *out = s.release();
    return EDGPU_OK;
}

static void destroy(S *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    if (s->comm.kind == Comm::RCCL && s->comm.nccl && g_rccl.ok) g_rccl.CommDestroy(s->comm.nccl);
    for (auto e : s->bench_ev) (void)hipEventDestroy(e);
    if (s->ev_pack) (void)hipEventDestroy(s->ev_pack);
    if (s->ev_a) (void)hipEventDestroy(s->ev_a);
    if (s->ev_b) (void)hipEventDestroy(s->ev_b);
    if (s->ev_gather) (void)hipEventDestroy(s->ev_gather);
    if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->cstream) (void)hipStreamDestroy(s->cstream);
    delete s;
}

// ---------------------------------------------------------------------------------------
// exchange step: gather the input vector of every rank (phonon block by phonon block)
// ---------------------------------------------------------------------------------------
static int halo_exchange(S *s, const double *v_local, hipStream_t st);
static int gather_vector(S *s, const double *v_local, hipStream_t st)
{
    if (s->halo) return halo_exchange(s, v_local, st);
    const SectorHost &H = s->H; Comm &c = s->comm;
    const int P = c.size;
    std::vector<int64_t> counts(P), displs(P);
    for (int r = 0; r < P; r++) {
        int64_t q = H.DimDw / P, rem = H.DimDw % P;
        int64_t qd = r < rem ? q + 1 : q, d0 = r < rem ? r * (q + 1) : r * q + rem;
        counts[r] = qd * H.DimUp; displs[r] = d0 * H.DimUp;
    }
    const int64_t nel_loc = s->nloc_el, nel = H.DimUp * H.DimDw;
    for (int iph = 0; iph < H.DimPh; iph++) {
        const double *send = v_local + (int64_t)iph * nel_loc;
        double *recv = s->vfull.p + (int64_t)iph * nel;
        if (c.kind == Comm::RCCL) {
            HIPCHK(hipMemcpyAsync(recv + displs[c.rank], send, (size_t)nel_loc * 8, hipMemcpyDeviceToDevice, st));
            int rc = g_rccl.GroupStart();
            for (int r = 0; r < P && rc == 0; r++) {
                if (r == c.rank) continue;
                rc = g_rccl.Send(send, (size_t)nel_loc, kNcclDouble, r, c.nccl, st);
                if (rc == 0) rc = g_rccl.Recv(recv + displs[r], (size_t)counts[r], kNcclDouble, r, c.nccl, st);
            }
            int rc2 = g_rccl.GroupEnd();
            if (rc || rc2) return fail(EDGPU_ECOMM, std::string("edgpu: RCCL send/recv failed: ") + g_rccl.GetErrorString(rc ? rc : rc2));
        } else if (c.kind == Comm::CALLBACK) {
            HIPCHK(hipStreamSynchronize(st));
            if (c.ag(c.ctx, send, nel_loc, recv, counts.data(), displs.data(), P)) return fail(EDGPU_ECOMM, "edgpu: allgatherv callback failed");
        } else
            return fail(EDGPU_ECOMM, "edgpu: mpi_size>1 but no communicator attached (edgpu_comm_init_*)");
    }
    return EDGPU_OK;
}

// x = [own rows | needed rows of the other ranks]: pack what each peer needs, one send/recv pair per peer
static int halo_exchange(S *s, const double *v_local, hipStream_t st)
{
    const SectorHost &H = s->H; Comm &c = s->comm;
    const int P = c.size;
    int64_t nsend_rows = 0; for (int r = 0; r < P; r++) nsend_rows += s->send_rows_cnt[r];
    const int64_t rowlen = H.DimUp;
    if (nsend_rows > 0) {
        PackArgs K{}; K.DimUp = (int32_t)H.DimUp; K.DimPh = H.DimPh; K.Qdw = H.Qdw; K.nrows = nsend_rows; K.list = s->send_list.p; K.v = v_local; K.out = s->sendbuf.p;
        const int64_t tot = nsend_rows * H.DimPh * cdiv(rowlen, 2);
        hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)std::min<int64_t>(cdiv(tot, 256), 65536)), dim3(256), 0, st, K);
        HIPCHK(hipGetLastError());
    }
    for (int iph = 0; iph < H.DimPh; iph++) {
        double *x = s->vfull.p + (int64_t)iph * s->nhalo * rowlen;
        HIPCHK(hipMemcpyAsync(x, v_local + (int64_t)iph * s->nloc_el, (size_t)s->nloc_el * 8, hipMemcpyDeviceToDevice, st));
        const double *sb = s->sendbuf.p + (int64_t)iph * nsend_rows * rowlen;
        if (P == 1) continue;
        if (c.kind == Comm::RCCL) {
            int rc = g_rccl.GroupStart();
            for (int r = 0; r < P && rc == 0; r++) {
                if (r == c.rank) continue;
                if (s->send_rows_cnt[r]) rc = g_rccl.Send(sb + s->send_rows_off[r] * rowlen, (size_t)(s->send_rows_cnt[r] * rowlen), kNcclDouble, r, c.nccl, st);
                if (rc == 0 && s->recv_rows_cnt[r]) rc = g_rccl.Recv(x + (H.Qdw + s->recv_rows_off[r]) * rowlen, (size_t)(s->recv_rows_cnt[r] * rowlen), kNcclDouble, r, c.nccl, st);
            }
            int rc2 = g_rccl.GroupEnd();
            if (rc || rc2) return fail(EDGPU_ECOMM, std::string("edgpu: RCCL send/recv failed: ") + g_rccl.GetErrorString(rc ? rc : rc2));
        } else if (c.kind == Comm::CALLBACK) {
            if (!c.xch) return fail(EDGPU_ECOMM, "edgpu: the callback exchange needs an all-to-all-v callback");
            std::vector<int64_t> sc(P), sd(P), rcv(P), rd(P);
            for (int r = 0; r < P; r++) { sc[r] = s->send_rows_cnt[r] * rowlen; sd[r] = s->send_rows_off[r] * rowlen; rcv[r] = s->recv_rows_cnt[r] * rowlen; rd[r] = (H.Qdw + s->recv_rows_off[r]) * rowlen; }
            HIPCHK(hipStreamSynchronize(st));
            if (c.xch(c.ctx, sb, sc.data(), sd.data(), x, rcv.data(), rd.data(), P)) return fail(EDGPU_ECOMM, "edgpu: exchange callback failed");
        } else
            return fail(EDGPU_ECOMM, "edgpu: mpi_size>1 but no communicator attached (edgpu_comm_init_*)");
    }
    return EDGPU_OK;
}

// sum n device doubles over ranks (in place)
static int allreduce_dev(S *s, double *p, int n)
{
    Comm &c = s->comm;
    if (!s->multi) return EDGPU_OK;
    if (c.kind == Comm::RCCL) {
        int rc = g_rccl.AllReduce(p, p, (size_t)n, kNcclDouble, kNcclSum, c.nccl, s->stream);
        if (rc) return fail(EDGPU_ECOMM, std::string("edgpu: RCCL allreduce failed: ") + g_rccl.GetErrorString(rc));
    } else if (c.kind == Comm::CALLBACK) {
        std::vector<double> h(n);
        HIPCHK(hipMemcpyAsync(h.data(), p, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
        if (c.ar(c.ctx, h.data(), n)) return fail(EDGPU_ECOMM, "edgpu: allreduce callback failed");
        HIPCHK(hipMemcpyAsync(p, h.data(), (size_t)n * 8, hipMemcpyHostToDevice, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
    } else
        return fail(EDGPU_ECOMM, "edgpu: mpi_size>1 but no communicator attached");
    return EDGPU_OK;
}
// sum scal[SC_TMP] over ranks
static int allreduce_tmp(S *s) { return allreduce_dev(s, s->scal.p + SC_TMP, 1); }

// ---------------------------------------------------------------------------------------
// one product  out = H*v  (epi=false)   or the Lanczos-fused form (epi=true):
//   out = s_cur*H*u - beta*s_prev*u_prev ;  scal[TMP] = (s_cur*u) . out
// ---------------------------------------------------------------------------------------
// scal[TMP] = sum(partial[0..n)) : one pass for short arrays, two passes for long ones
static int reduce_partials(S *s, int64_t n)
{
    const double *src = s->partial.p;
    if (n > 16384) {
        const int64_t chunk = cdiv(n, 1024);
        const int nb = (int)cdiv(n, chunk);
        hipLaunchKernelGGL(k_reduce_chunks, dim3(nb), dim3(256), 0, s->stream, src, n, chunk, s->partial2.p);
        src = s->partial2.p; n = nb;
    }
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, src, n, s->scal.p + SC_TMP);
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}

template <int F> static void launch_kron2(S *s, const Kron2Args &A, int64_t rows)
{
    size_t lds = (F & KF_LDS) ? s->kron_lds : 16 * sizeof(double);
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) { (void)hipFuncSetAttribute((const void *)k_kron2<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    hipLaunchKernelGGL(k_kron2<F>, dim3((unsigned)rows), dim3(s->kron_threads), lds, s->stream, A);
}

static int dispatch_kron2(S *s, int flags, const Kron2Args &A, int64_t rows)
{
#define CASE(f) case (f): launch_kron2<(f)>(s, A, rows); break;
#define CASE4(b) CASE(b) CASE((b) | KF_DIRECT) CASE((b) | KF_EPI) CASE((b) | KF_DIRECT | KF_EPI)
#define CASE8(b) CASE4(b) CASE4((b) | KF_PH | KF_ND)
    switch (flags) {
        CASE8(0) CASE8(KF_LDS) CASE8(KF_NODW) CASE8(KF_NODW | KF_LDS) CASE8(KF_NOLOCAL)
    default: return fail(EDGPU_EINVAL, "edgpu: internal kernel variant missing");
    }
#undef CASE8
#undef CASE4
#undef CASE
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}

static bool direct_tables_dirty(S *) { return false; }
static bool epi_partials_needed(int flags) { return (flags & KF_EPI) && !(flags & KF_NODW); }
static int direct_tables(S *s)
{
    const SectorHost &H = s->H; const edgpu_desc &d = H.desc;
    for (int k = 0; k < 2 * H.Ns_Ud; k++) {
        DevFactor &f = *s->fac[k];
        TabArgs T{};
        T.dim = H.fac[k].dim; T.nterms = (int32_t)H.fac[k].terms.size(); T.width = f.width; T.transposed = k < H.Ns_Ud;
        T.map = f.map.p; T.rank = f.rank.p; T.talpha = f.talpha.p; T.tbeta = f.tbeta.p; T.tval = f.tval.p;
        T.ellc = f.ellc.p; T.ellv = f.ellv.p; T.lin = f.lin.p; T.nlev = H.fac[k].nbits;
        T.pcols = f.packed ? f.pcols.p : nullptr; T.pcodes = f.packed ? f.pcodes.p : nullptr; T.pelem = f.packed_elem ? f.pelem.p : nullptr; T.pelem8 = f.packed_elem8 ? f.pelem8.p : nullptr;
        T.impmask = d.ed_total_ud ? (1 << d.Norb) - 1 : 0; T.pair = d.ed_total_ud ? d.Ust - d.Jh : 0.0; T.E = f.E.p;
        hipLaunchKernelGGL(k_direct_tables, dim3((unsigned)cdiv(T.dim, 256)), dim3(256), 0, s->stream, T);
    }
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}

static int hxv(S *s, const double *v, double *out, bool epi, const double *uprev)
{
    const SectorHost &H = s->H; const edgpu_desc &d = H.desc;
    const bool multi = s->multi;
    const bool direct = d.storage == EDGPU_STORE_DIRECT;
    if (direct) CHK(direct_tables(s));
    if (multi) {  // start the exchange on the comm stream; local work overlaps it
        HIPCHK(hipEventRecord(s->ev_ready, s->stream));
        HIPCHK(hipStreamWaitEvent(s->cstream, s->ev_ready, 0));
        CHK(gather_vector(s, v, s->cstream));
        HIPCHK(hipEventRecord(s->ev_gather, s->cstream));
    }
    int64_t nblocks = 0;
    if (d.storage == EDGPU_STORE_CSR) {
        // split CSR: the local block only reads columns of the rank's own dw columns, so it runs on the
        // shard itself (pointer shifted to global column numbering) while the exchange is in flight
        const bool overlap = multi && s->csr_split && H.DimPh == 1;
        if (multi && !overlap) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_gather, 0));
        CsrArgs A{};
        A.nrows = s->nloc_el; A.rows_per_block = s->csr_rpb; A.DimUp = (int32_t)H.DimUp; A.blocks_per_row = s->csr_bpr; A.Qdw = H.Qdw; A.tile = s->csr_tile;
        A.rowptr = s->csr_rowptr.p; A.cols = s->csr_cols.p; A.vals = s->csr_vals.p;
        A.x = overlap ? (s->halo ? v : v - H.dw0 * H.DimUp) : (multi ? s->vfull.p : v); A.y = out; A.v_local = v; A.uprev = uprev; A.scal = s->scal.p; A.partial = s->partial.p; A.epi = epi; A.xcd_rows = dbg_int("EDGPU_CSR_XCD", 0);
        // the column index is global; on one GPU local == global
        nblocks = (int64_t)s->csr_bpr * H.Qdw;
        A.pass = s->csr_split ? 1 : 0; A.coltiles = 0;
        if (s->csr_split && s->csrL_small) hipLaunchKernelGGL((k_csr_spmv2<64, 640, false>), dim3((unsigned)nblocks), dim3(64), 0, s->stream, A);
        else hipLaunchKernelGGL((k_csr_spmv2<64, 1152, false>), dim3((unsigned)nblocks), dim3(64), 0, s->stream, A);
        HIPCHK(hipGetLastError());
        if (s->csr_split) {
            if (overlap) { HIPCHK(hipStreamWaitEvent(s->stream, s->ev_gather, 0)); A.x = s->vfull.p; }
            A.rowptr = s->csrD_rowptr.p; A.cols = s->csrD_cols.p; A.vals = s->csrD_vals.p;
            A.rows_per_block = s->csrD_rpb; A.blocks_per_row = s->csrD_bpr; A.pass = 2; A.coltiles = 1; A.xcd_rows = 0;
            nblocks = 8 * cdiv(s->csrD_bpr, 8) * H.Qdw;
            if (s->csrD_small) {
                if (s->csrD_nt) hipLaunchKernelGGL((k_csr_spmv2<64, 640, true>), dim3((unsigned)nblocks), dim3(64), 0, s->stream, A);
                else hipLaunchKernelGGL((k_csr_spmv2<64, 640, false>), dim3((unsigned)nblocks), dim3(64), 0, s->stream, A);
            } else {
                if (s->csrD_nt) hipLaunchKernelGGL((k_csr_spmv2<64, 1152, true>), dim3((unsigned)nblocks), dim3(64), 0, s->stream, A);
                else hipLaunchKernelGGL((k_csr_spmv2<64, 1152, false>), dim3((unsigned)nblocks), dim3(64), 0, s->stream, A);
            }
            HIPCHK(hipGetLastError());
        }
    } else if (s->two_factor) {
        Kron2Args A{};
        A.DimUp = (int32_t)H.DimUp; A.DimDw = (int32_t)(s->halo ? s->nhalo : H.DimDw); A.DimPh = H.DimPh; A.Norb = d.Norb; A.Qdw = H.Qdw; A.dw0 = H.dw0;
        A.upc = s->fac[0]->ellc.p; A.upv = s->fac[0]->ellv.p; A.upw = s->fac[0]->width;
        A.dwc = s->halo ? s->dw_ellc_x.p : s->fac[1]->ellc.p; A.dwv = s->fac[1]->ellv.p; A.dww = s->fac[1]->width;
        A.diag = s->diag.p; A.Eup = s->fac[0]->E.p; A.Edw = s->fac[1]->E.p; A.cross = s->cross.p;
        A.mapup = s->fac[0]->map.p; A.mapdw = s->fac[1]->map.p; A.rankup = s->fac[0]->rank.p; A.rankdw = s->fac[1]->rank.p;
        A.impmask = (1 << d.Norb) - 1; A.nI = 1 << d.Norb; A.econst = H.econst; A.w0 = d.w0_ph; A.gsum = H.gph_sum; A.Jx = d.Jx; A.Jp = d.Jp;
        A.gup = s->gup.p; A.gdw = s->gdw.p;
        A.v = v; A.vfull = multi ? s->vfull.p : v; A.out = out; A.uprev = uprev; A.scal = s->scal.p; A.partial = s->partial.p;
        if (dbg_flags() & 1) A.upw = 0;
        if (dbg_flags() & 2) A.dww = 0;
        int base = (direct ? KF_DIRECT : 0) | (epi ? KF_EPI : 0) | (s->extra ? (KF_PH | KF_ND) : 0);
        nblocks = H.Qdw * H.DimPh;
        const bool split = multi || (s->kron_split && !s->extra);
        if (!split) CHK(dispatch_kron2(s, base | (s->use_lds ? KF_LDS : 0), A, nblocks));
        else {
            const bool persistent = s->up_persistent && !s->extra && s->fac[0]->packed && s->fac[0]->width <= 16 && s->use_lds && H.DimUp <= 13312 && !direct_tables_dirty(s);
            if (persistent && s->up_pairs && s->fac[0]->packed_elem) {
                UpArgs2 U{};
                U.DimUp = A.DimUp; U.DimPh = A.DimPh; U.Qdw = A.Qdw; U.dw0 = A.dw0; U.nrows = H.Qdw * H.DimPh;
                U.packed = reinterpret_cast<const uint4 *>(s->fac[0]->pelem.p); U.tval = s->fac[0]->ptval.p; U.nterms = s->fac[0]->pterms;
                U.diag = A.diag; U.Eup = A.Eup; U.Edw = A.Edw; U.cross = A.cross; U.mapup = A.mapup; U.mapdw = A.mapdw; U.impmask = A.impmask; U.nI = A.nI; U.econst = A.econst;
                U.v = v; U.out = out; U.uprev = uprev; U.scal = s->scal.p;
                const size_t lds = (size_t)H.DimUp * 8 + 128 * 8;
                const int grid = (int)std::min<int64_t>(U.nrows, s->up_grid);
                static bool attr2 = false;
                if (!attr2) {
                    (void)hipFuncSetAttribute((const void *)k_uppass2<false, false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    (void)hipFuncSetAttribute((const void *)k_uppass2<false, true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    attr2 = true;
                }
                if (direct) {
                    static bool attr3 = false;
                    if (!attr3) {
                        (void)hipFuncSetAttribute((const void *)k_uppass2<true, false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        (void)hipFuncSetAttribute((const void *)k_uppass2<true, true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        attr3 = true;
                    }
                    if (epi) hipLaunchKernelGGL((k_uppass2<true, true, 1024>), dim3(grid), dim3(1024), lds, s->stream, U);
                    else hipLaunchKernelGGL((k_uppass2<true, false, 1024>), dim3(grid), dim3(1024), lds, s->stream, U);
                } else if (epi) hipLaunchKernelGGL((k_uppass2<false, true, 1024>), dim3(grid), dim3(1024), lds, s->stream, U);
                else hipLaunchKernelGGL((k_uppass2<false, false, 1024>), dim3(grid), dim3(1024), lds, s->stream, U);
                HIPCHK(hipGetLastError());
            } else if (persistent) {
                UpArgs U{};
                U.DimUp = A.DimUp; U.DimPh = A.DimPh; U.Qdw = A.Qdw; U.dw0 = A.dw0; U.nrows = H.Qdw * H.DimPh;
                U.pcols = s->fac[0]->pcols.p; U.pcodes = s->fac[0]->pcodes.p; U.upw = s->fac[0]->width; U.tval = s->fac[0]->ptval.p; U.nterms = s->fac[0]->pterms;
                U.diag = A.diag; U.Eup = A.Eup; U.Edw = A.Edw; U.cross = A.cross; U.mapup = A.mapup; U.mapdw = A.mapdw; U.impmask = A.impmask; U.nI = A.nI; U.econst = A.econst;
                U.v = v; U.out = out; U.uprev = uprev; U.scal = s->scal.p;
                const size_t lds = (((size_t)H.DimUp + 1) & ~(size_t)1) * 8 + 128 * 8;
                const int grid = (int)std::min<int64_t>(U.nrows, s->up_grid);
#define UP_LAUNCH(D_, E_, T_)                                                                                                   \
    do {                                                                                                                            \
        static bool attr_ = false;                                                                                                  \
        if (!attr_) { (void)hipFuncSetAttribute((const void *)k_uppass<D_, E_, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_ = true; } \
        hipLaunchKernelGGL((k_uppass<D_, E_, T_>), dim3(grid), dim3(T_), lds, s->stream, U);                                        \
    } while (0)
                if (s->up_threads == 512) {
                    if (direct) { if (epi) UP_LAUNCH(true, true, 512); else UP_LAUNCH(true, false, 512); }
                    else { if (epi) UP_LAUNCH(false, true, 512); else UP_LAUNCH(false, false, 512); }
                } else {
                    if (direct) { if (epi) UP_LAUNCH(true, true, 1024); else UP_LAUNCH(true, false, 1024); }
                    else { if (epi) UP_LAUNCH(false, true, 1024); else UP_LAUNCH(false, false, 1024); }
                }
#undef UP_LAUNCH
                HIPCHK(hipGetLastError());
            } else if (!s->use_lds && !s->extra && s->fac[0]->packed_elem8 && s->up_big) {
                UpBigArgs U{};
                U.DimUp = A.DimUp; U.DimPh = A.DimPh; U.rsplit = s->row_split; U.Qdw = A.Qdw; U.dw0 = A.dw0; U.nrows = H.Qdw * H.DimPh;
                U.packed8 = reinterpret_cast<const uint4 *>(s->fac[0]->pelem8.p); U.tval = s->fac[0]->ptval.p; U.nterms = s->fac[0]->pterms;
                U.diag = A.diag; U.Eup = A.Eup; U.Edw = A.Edw; U.cross = A.cross; U.mapup = A.mapup; U.mapdw = A.mapdw; U.impmask = A.impmask; U.nI = A.nI; U.econst = A.econst;
                U.v = v; U.out = out; U.uprev = uprev; U.scal = s->scal.p;
                const unsigned grid = (unsigned)(cdiv(U.nrows, 8) * 8 * s->row_split);
                if (direct) { if (epi) hipLaunchKernelGGL((k_uppass_big<true, true>), dim3(grid), dim3(512), 0, s->stream, U); else hipLaunchKernelGGL((k_uppass_big<true, false>), dim3(grid), dim3(512), 0, s->stream, U); }
                else { if (epi) hipLaunchKernelGGL((k_uppass_big<false, true>), dim3(grid), dim3(512), 0, s->stream, U); else hipLaunchKernelGGL((k_uppass_big<false, false>), dim3(grid), dim3(512), 0, s->stream, U); }
                HIPCHK(hipGetLastError());
            } else if (!s->use_lds && !epi_partials_needed(base | KF_NODW)) {
                Kron2Args A2 = A; A2.rsplit = s->row_split;
                CHK(dispatch_kron2(s, base | KF_NODW, A2, s->row_split > 1 ? cdiv(nblocks, 8) * 8 * s->row_split : nblocks));
            } else
                CHK(dispatch_kron2(s, base | KF_NODW | (s->use_lds ? KF_LDS : 0), A, nblocks));
            if (multi) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_gather, 0));
            if (s->extra) CHK(dispatch_kron2(s, base | KF_NOLOCAL, A, nblocks));   // Jx/Jp and phonon cases: row-wise dw pass
            else {
                DwArgs D{};
                D.DimUp = A.DimUp; D.DimDw = A.DimDw; D.DimPh = A.DimPh; D.Qdw = A.Qdw; D.dw0 = A.dw0;
                D.dwc = A.dwc; D.dwv = A.dwv; D.dww = A.dww; D.v = v; D.vfull = A.vfull; D.out = out; D.scal = s->scal.p; D.partial = s->partial.p;
                D.w = s->dw_w; D.wshift = 0; while ((1 << D.wshift) < D.w) D.wshift++;
                const int dv = s->dw_variant;
                const bool vec2 = dv >= 2 && (H.DimUp % 2 == 0) && D.w >= 4 && D.dww <= 64;
                D.ntiles = (int32_t)cdiv(H.DimUp, D.w);
                D.R = vec2 ? std::min(64, s->dw_rows) : 256 / D.w;
                D.chunks = (int32_t)cdiv(H.Qdw * H.DimPh, D.R);
                nblocks = 8 * cdiv(D.ntiles, 8) * (int64_t)D.chunks;
                if (!vec2) {
                    if (epi) hipLaunchKernelGGL(k_dwpass<true>, dim3((unsigned)nblocks), dim3(256), 0, s->stream, D);
                    else hipLaunchKernelGGL(k_dwpass<false>, dim3((unsigned)nblocks), dim3(256), 0, s->stream, D);
                } else if (dv == 2) {
                    if (epi) hipLaunchKernelGGL((k_dwpass2<true, false>), dim3((unsigned)nblocks), dim3(256), 0, s->stream, D);
                    else hipLaunchKernelGGL((k_dwpass2<false, false>), dim3((unsigned)nblocks), dim3(256), 0, s->stream, D);
                } else {
                    if (epi) hipLaunchKernelGGL((k_dwpass2<true, true>), dim3((unsigned)nblocks), dim3(256), 0, s->stream, D);
                    else hipLaunchKernelGGL((k_dwpass2<false, true>), dim3((unsigned)nblocks), dim3(256), 0, s->stream, D);
                }
                HIPCHK(hipGetLastError());
            }
        }
    } else {
        GenArgs A{};
        A.nfac = 2 * H.Ns_Ud; A.DimPh = H.DimPh; A.Norb = d.Norb; A.direct = direct;
        int64_t st = 1;
        for (int k = 0; k < A.nfac; k++) {
            A.dims[k] = H.fac[k].dim; A.strides[k] = st; st *= H.fac[k].dim;
            A.rowptr[k] = s->fac[k]->rowptr.p; A.cols[k] = s->fac[k]->cols.p; A.vals[k] = s->fac[k]->vals.p; A.E[k] = s->fac[k]->E.p; A.maps[k] = s->fac[k]->map.p;
        }
        A.diag = s->diag.p; A.cross = s->cross.p; A.gup = s->gup.p; A.gdw = s->gdw.p; A.nI = 1 << d.Norb; A.total_ud = d.ed_total_ud; A.impmask = (1 << d.Norb) - 1;
        A.econst = H.econst; A.w0 = d.w0_ph; A.gsum = H.gph_sum; A.Nel = s->nloc_el;
        A.v = v; A.out = out; A.uprev = uprev; A.scal = s->scal.p; A.partial = s->partial.p; A.epi = epi;
        nblocks = std::min<int64_t>(cdiv(s->nloc, 256), 4096);
        hipLaunchKernelGGL(k_generic, dim3((unsigned)nblocks), dim3(256), 0, s->stream, A);
        HIPCHK(hipGetLastError());
    }
    if (epi) {
        CHK(reduce_partials(s, nblocks));
        CHK(allreduce_tmp(s));
    }
    return EDGPU_OK;
}

// ---------------------------------------------------------------------------------------
// symmetric tridiagonal eigen-solver (implicit QL with Wilkinson shifts), host, n <= ~1000.
// d: diagonal (n), e: sub-diagonal e[1..n-1] (e[0] unused).  On exit d = eigenvalues
// (unsorted); if z != nullptr (n*n, row-major z[i*n+j], identity on entry) columns = vectors.
// ---------------------------------------------------------------------------------------
static bool tqli(std::vector<double> &d, std::vector<double> &e, int n, double *z)
{
    for (int i = 1; i < n; i++) e[i - 1] = e[i];
    if (n > 0) e[n - 1] = 0.0;
    for (int l = 0; l < n; l++) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; m++) {
                double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 120) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double sn = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; i--) {
                    double f = sn * e[i], b = c * e[i];
                    e[i + 1] = (r = std::hypot(f, g));
                    if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                    sn = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * sn + 2.0 * c * b;
                    d[i + 1] = g + (p = sn * r);
                    g = c * r - b;
                    if (z)
                        for (int k = 0; k < n; k++) {
                            double f2 = z[(size_t)k * n + i + 1];
                            z[(size_t)k * n + i + 1] = sn * z[(size_t)k * n + i] + c * f2;
                            z[(size_t)k * n + i] = c * z[(size_t)k * n + i] - sn * f2;
                        }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    return true;
}

// lowest eigenvalue (and optionally its eigenvector) of T_k = tridiag(alphas[0..k), betas[1..k))
static bool tridiag_lowest(const double *al, const double *be, int k, double *e0, std::vector<double> *zcol)
{
    std::vector<double> d(al, al + k), e(k, 0.0);
    for (int i = 1; i < k; i++) e[i] = be[i];
    std::vector<double> z;
    if (zcol) { z.assign((size_t)k * k, 0.0); for (int i = 0; i < k; i++) z[(size_t)i * k + i] = 1.0; }
    if (!tqli(d, e, k, zcol ? z.data() : nullptr)) return false;
    int im = 0;
    for (int i = 1; i < k; i++) if (d[i] < d[im]) im = i;
    *e0 = d[im];
    if (zcol) { zcol->resize(k); for (int i = 0; i < k; i++) (*zcol)[i] = z[(size_t)i * k + im]; }
    return true;
}

// ---------------------------------------------------------------------------------------
// Lanczos loop, device resident.  Restates SciFortran's lanczos_iteration / sp_lanc_eigh /
// sp_lanc_tridiag (module SF_SP_LINALG, not part of the reference tree; call sites
// ED_DIAG.f90:177-185, ED_GF_NORMAL.f90:232-237) with vectors kept unnormalised in HBM and the
// 1/beta scale carried in a device scalar, so one iteration is: fused product+alpha, one
// axpy+norm pass, two single-workgroup reductions -- and no host synchronisation.
// ---------------------------------------------------------------------------------------
static int lz_prepare(S *s, int nmax)
{
    for (int i = 0; i < 3; i++) if (s->lz[i].n != (size_t)s->nloc) CHK(s->lz[i].alloc((size_t)s->nloc));
    if (s->nmax < nmax) {
        CHK(s->alphas.alloc((size_t)nmax + 2)); CHK(s->betas.alloc((size_t)nmax + 2)); CHK(s->zvec.alloc((size_t)nmax + 2));
        s->nmax = nmax;
    }
    HIPCHK(hipMemsetAsync(s->alphas.p, 0, ((size_t)nmax + 2) * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->betas.p, 0, ((size_t)nmax + 2) * 8, s->stream));
    return EDGPU_OK;
}

static int reduce_dot(S *s, const double *a, const double *b)
{
    int nb = (int)std::min<int64_t>(cdiv(s->nloc, 256 * 8), 2048);
    nb = std::max(nb, 1);
    hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(256), 0, s->stream, a, b, s->nloc, s->partial.p);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, s->partial.p, (int64_t)nb, s->scal.p + SC_TMP);
    HIPCHK(hipGetLastError());
    return allreduce_tmp(s);
}

struct LzVecs { double *uprev, *ucur, *w; };

// start: ucur holds the (unnormalised) start vector
static int lz_start(S *s, LzVecs &V)
{
    HIPCHK(hipMemsetAsync(V.uprev, 0, (size_t)s->nloc * 8, s->stream));
    CHK(reduce_dot(s, V.ucur, V.ucur));
    hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, 0, 2);
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}

// iteration k (0-based): fills alphas[k], betas[k+1]; optional eigenvector accumulation
static int lz_iter_tail(S *s, LzVecs &V, int k, double *acc)
{
    hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, k, 0);
    if (acc) hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)s->zvec.p, k, 3);
    int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(s->nloc, 256 * 4), 4096));
    hipLaunchKernelGGL(k_axpy_norm, dim3(nb), dim3(256), 0, s->stream, V.w, V.ucur, acc, s->scal.p, s->nloc, s->partial.p);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, s->partial.p, (int64_t)nb, s->scal.p + SC_TMP);
    HIPCHK(hipGetLastError());
    CHK(allreduce_tmp(s));
    hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, k, 1);
    HIPCHK(hipGetLastError());
    double *t = V.uprev; V.uprev = V.ucur; V.ucur = V.w; V.w = t;
    return EDGPU_OK;
}
static int lz_iter(S *s, LzVecs &V, int k, double *acc)
{
    CHK(hxv(s, V.ucur, V.w, true, V.uprev));
    return lz_iter_tail(s, V, k, acc);
}

static int fetch_coeffs(S *s, int k, std::vector<double> &al, std::vector<double> &be)
{
    al.resize(k + 1); be.resize(k + 2);
    HIPCHK(hipMemcpyAsync(al.data(), s->alphas.p, (size_t)(k + 1) * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(be.data(), s->betas.p, (size_t)(k + 2) * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return EDGPU_OK;
}

static int lanczos_gs(S *s, int nitermax, double threshold, int ncheck, double *egs, double *vect_dev, int *nlanc_out)
{
    if (nitermax < 1) return fail(EDGPU_EINVAL, "edgpu_lanczos_gs: nitermax < 1");
    if (ncheck < 1) ncheck = 10;
    if ((int64_t)nitermax > s->H.Dim) nitermax = (int)s->H.Dim;   // ED_DIAG.f90:100 Nitermax=min(dim,lanc_niter)
    CHK(lz_prepare(s, nitermax));
    for (int i = 3; i < 5; i++) if (s->lz[i].n != (size_t)s->nloc) CHK(s->lz[i].alloc((size_t)s->nloc));
    double *start = s->lz[3].p, *acc = s->lz[4].p;
    // zero input => library-generated start vector (ED_DIAG.f90:147 passes eig_basis=0)
    HIPCHK(hipMemcpyAsync(start, vect_dev, (size_t)s->nloc * 8, hipMemcpyDeviceToDevice, s->stream));
    CHK(reduce_dot(s, start, start));
    double nrm2;
    HIPCHK(hipMemcpyAsync(&nrm2, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
    if (nrm2 == 0.0) {
        const SectorHost &H = s->H;
        hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, s->stream, start, H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
        HIPCHK(hipGetLastError());
    }
    LzVecs V{s->lz[0].p, s->lz[1].p, s->lz[2].p};
    HIPCHK(hipMemcpyAsync(V.ucur, start, (size_t)s->nloc * 8, hipMemcpyDeviceToDevice, s->stream));
    CHK(lz_start(s, V));
    std::vector<double> al, be;
    int nlanc = 0; double esave = 0.0, e0 = 0.0;
    for (int k = 0; k < nitermax; k++) {
        CHK(lz_iter(s, V, k, nullptr));
        CHK(fetch_coeffs(s, k, al, be));
        nlanc = k + 1;
        if (std::fabs(be[k + 1]) < threshold) break;       // invariant subspace reached
        if (!tridiag_lowest(al.data(), be.data(), nlanc, &e0, nullptr)) return fail(EDGPU_ENOTCONVERGED, "edgpu: tridiagonal QL did not converge");
        if (nlanc >= ncheck && std::fabs(e0 - esave) <= threshold) break;
        esave = e0;
    }
    std::vector<double> z;
    if (!tridiag_lowest(al.data(), be.data(), nlanc, &e0, &z)) return fail(EDGPU_ENOTCONVERGED, "edgpu: tridiagonal QL did not converge");
    *egs = e0; *nlanc_out = nlanc;
    // second pass: regenerate the Krylov vectors and accumulate vect = sum_k Z(k,1) v_k
    HIPCHK(hipMemcpyAsync(s->zvec.p, z.data(), (size_t)nlanc * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemsetAsync(acc, 0, (size_t)s->nloc * 8, s->stream));
    V = LzVecs{s->lz[0].p, s->lz[1].p, s->lz[2].p};
    HIPCHK(hipMemcpyAsync(V.ucur, start, (size_t)s->nloc * 8, hipMemcpyDeviceToDevice, s->stream));
    CHK(lz_start(s, V));
    for (int k = 0; k < nlanc; k++) CHK(lz_iter(s, V, k, acc));
    CHK(reduce_dot(s, acc, acc));
    hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, 0, 2);
    int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(s->nloc, 256 * 4), 4096));
    hipLaunchKernelGGL(k_scale, dim3(nb), dim3(256), 0, s->stream, acc, s->scal.p, (int)SC_SCUR, s->nloc);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(vect_dev, acc, (size_t)s->nloc * 8, hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return EDGPU_OK;
}

static int lanczos_tridiag(S *s, const double *vin_dev, int nlanc, double threshold, double *alanc, double *blanc, int *nused)
{
    if (nlanc < 1) return fail(EDGPU_EINVAL, "edgpu_lanczos_tridiag: nlanc < 1");
    CHK(lz_prepare(s, nlanc));
    LzVecs V{s->lz[0].p, s->lz[1].p, s->lz[2].p};
    HIPCHK(hipMemcpyAsync(V.ucur, vin_dev, (size_t)s->nloc * 8, hipMemcpyDeviceToDevice, s->stream));
    CHK(lz_start(s, V));
    for (int i = 0; i < nlanc; i++) { alanc[i] = 0.0; blanc[i] = 0.0; }
    std::vector<double> al, be;
    int used = 0;
    const int chunk = 8;  // look at beta on the host every `chunk` iterations
    for (int k = 0; k < nlanc;) {
        int kend = std::min(nlanc, k + chunk);
        for (; k < kend; k++) CHK(lz_iter(s, V, k, nullptr));
        CHK(fetch_coeffs(s, k - 1, al, be));
        bool stop = false;
        for (int i = used; i < k; i++) {
            alanc[i] = al[i]; used = i + 1;
            if (std::fabs(be[i + 1]) < threshold) { stop = true; break; }
            if (i + 1 < nlanc) blanc[i + 1] = be[i + 1];
        }
        if (stop) break;
    }
    *nused = used;
    return EDGPU_OK;
}

// ---------------------------------------------------------------------------------------
// Lowest `nev` eigenpairs: thick-restart Lanczos with full re-orthogonalisation in a basis of
// `ncv` vectors held in HBM, explicit Rayleigh-Ritz on the host.  Stands in for the ARPACK call
// sp_eigh(MatVec,eval,evec,Nblock,Nitermax,tol) of ED_DIAG.f90:153-170 (which='SA', ncv=Nblock,
// maxiter=Nitermax restarts, residual test |r_i| <= tol*max(eps^(2/3),|theta_i|)).
// ---------------------------------------------------------------------------------------
static void jacobi_eig(std::vector<double> &A, int m, std::vector<double> &w, std::vector<double> &Y)
{   // cyclic Jacobi on the symmetric m x m matrix A (row-major); eigenvalues ascending in w, vectors = columns of Y (Y[i + m*j])
    std::vector<double> V((size_t)m * m, 0.0);
    for (int i = 0; i < m; i++) V[(size_t)i * m + i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, dg = 0.0;
        for (int p = 0; p < m; p++) { dg += A[(size_t)p * m + p] * A[(size_t)p * m + p]; for (int q = p + 1; q < m; q++) off += A[(size_t)p * m + q] * A[(size_t)p * m + q]; }
        if (off <= 1e-32 * (dg + 1e-300)) break;
        for (int p = 0; p < m; p++)
            for (int q = p + 1; q < m; q++) {
                const double apq = A[(size_t)p * m + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[(size_t)q * m + q] - A[(size_t)p * m + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < m; k++) { double akp = A[(size_t)k * m + p], akq = A[(size_t)k * m + q]; A[(size_t)k * m + p] = c * akp - sn * akq; A[(size_t)k * m + q] = sn * akp + c * akq; }
                for (int k = 0; k < m; k++) { double apk = A[(size_t)p * m + k], aqk = A[(size_t)q * m + k]; A[(size_t)p * m + k] = c * apk - sn * aqk; A[(size_t)q * m + k] = sn * apk + c * aqk; }
                for (int k = 0; k < m; k++) { double vkp = V[(size_t)k * m + p], vkq = V[(size_t)k * m + q]; V[(size_t)k * m + p] = c * vkp - sn * vkq; V[(size_t)k * m + q] = sn * vkp + c * vkq; }
            }
    }
    std::vector<int> idx(m);
    for (int i = 0; i < m; i++) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return A[(size_t)a * m + a] < A[(size_t)b * m + b]; });
    w.resize(m); Y.assign((size_t)m * m, 0.0);
    for (int j = 0; j < m; j++) { w[j] = A[(size_t)idx[j] * m + idx[j]]; for (int i = 0; i < m; i++) Y[i + (size_t)m * j] = V[(size_t)i * m + idx[j]]; }
}

static int eigsh(S *s, int nev, int ncv, int maxiter, double tol, double *evals, double *evecs_host, int *nconv_out, int *nrestart_out, int *nmatvec_out)
{
    const int64_t Dim = s->H.Dim, n = s->nloc;
    if (nev < 1) return fail(EDGPU_EINVAL, "edgpu_eigsh: nev < 1");
    if ((int64_t)nev >= Dim) return fail(EDGPU_EINVAL, "edgpu_eigsh: nev must be smaller than the sector dimension (the reference switches to LAPACK there, ED_DIAG.f90:105-107)");
    if (ncv > kMaxBasis) ncv = kMaxBasis;
    if ((int64_t)ncv > Dim) ncv = (int)Dim;
    if (ncv <= nev) ncv = (int)std::min<int64_t>(std::min<int64_t>(2 * nev + 1, kMaxBasis), Dim);
    if (ncv <= nev) return fail(EDGPU_EINVAL, "edgpu_eigsh: ncv must exceed nev");
    if (maxiter < 1) maxiter = 1;
    const double tol_eff = std::max(tol, 1e-12), eps23 = std::pow(2.220446049250313e-16, 2.0 / 3.0);
    const int m = ncv;
    while ((int)s->eigV.size() < m + 1) { auto b = std::make_unique<DevBuf<double>>(); CHK(b->alloc((size_t)n)); s->eigV.push_back(std::move(b)); }
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256 * 8), 1024));
    if (s->eig_h.n < (size_t)(m + 2)) CHK(s->eig_h.alloc((size_t)m + 2));
    if (s->eig_partial.n < (size_t)nb * (m + 1)) CHK(s->eig_partial.alloc((size_t)nb * (m + 1)));
    if (s->eig_y.n < (size_t)m * m) CHK(s->eig_y.alloc((size_t)m * m));
    auto Vp = [&](int i) { return s->eigV[i]->p; };
    const SectorHost &H = s->H;
    hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, s->stream, Vp(0), H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
    HIPCHK(hipGetLastError());
    auto norm_of = [&](const double *x, double &out) -> int {
        CHK(reduce_dot(s, x, x));
        HIPCHK(hipMemcpyAsync(&out, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
        out = std::sqrt(out);
        return EDGPU_OK;
    };
    const int gs = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256 * 4), 4096));
    double nrm; CHK(norm_of(Vp(0), nrm));
    hipLaunchKernelGGL(k_scale_copy, dim3(gs), dim3(256), 0, s->stream, Vp(0), (const double *)Vp(0), 1.0 / nrm, n);
    std::vector<double> Hm((size_t)m * m, 0.0), theta, Y, hh(m + 1), h2(m + 1);
    int l = 0, nconv = 0, it = 0, nmv = 0, meff = m; double beta_last = 0.0;
    for (it = 0; it < maxiter; it++) {
        meff = m;
        for (int j = l; j < m; j++) {
            double *w = Vp(j + 1);
            CHK(hxv(s, Vp(j), w, false, nullptr)); nmv++;
            BasisArgs B{}; B.nv = j + 1; for (int i = 0; i <= j; i++) B.V[i] = Vp(i);
            std::fill(hh.begin(), hh.end(), 0.0);
            for (int pass = 0; pass < 2; pass++) {      // classical Gram-Schmidt, twice
                hipLaunchKernelGGL(k_multidot, dim3(nb), dim3(256), 0, s->stream, B, (const double *)w, n, s->eig_partial.p);
                hipLaunchKernelGGL(k_multidot_final, dim3(1), dim3(256), 0, s->stream, (const double *)s->eig_partial.p, nb, B.nv, s->eig_h.p);
                HIPCHK(hipGetLastError());
                CHK(allreduce_dev(s, s->eig_h.p, B.nv));
                hipLaunchKernelGGL(k_multiaxpy, dim3(gs), dim3(256), 0, s->stream, B, (const double *)s->eig_h.p, w, n);
                HIPCHK(hipMemcpyAsync(h2.data(), s->eig_h.p, (size_t)B.nv * 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
                for (int i = 0; i <= j; i++) hh[i] += h2[i];
            }
            for (int i = 0; i <= j; i++) { Hm[(size_t)i * m + j] = hh[i]; Hm[(size_t)j * m + i] = hh[i]; }
            double beta; CHK(norm_of(w, beta));
            beta_last = beta;
            if (j + 1 < m) { Hm[(size_t)(j + 1) * m + j] = beta; Hm[(size_t)j * m + (j + 1)] = beta; }
            if (beta <= 1e-13 * std::max(1.0, std::fabs(hh[j]))) { meff = j + 1; break; }   // invariant subspace
            hipLaunchKernelGGL(k_scale_copy, dim3(gs), dim3(256), 0, s->stream, w, (const double *)w, 1.0 / beta, n);
            HIPCHK(hipGetLastError());
        }
        std::vector<double> Hs((size_t)meff * meff);
        for (int i = 0; i < meff; i++) for (int j = 0; j < meff; j++) Hs[(size_t)i * meff + j] = Hm[(size_t)i * m + j];
        jacobi_eig(Hs, meff, theta, Y);
        nconv = 0;
        const int want = std::min(nev, meff);
        for (int i = 0; i < want; i++) {
            const double resid = std::fabs(beta_last * Y[(meff - 1) + (size_t)meff * i]);
            if (resid <= tol_eff * std::max(eps23, std::fabs(theta[i]))) nconv++;
        }
        if (nconv >= want || meff < m || it + 1 >= maxiter) break;
        // thick restart: keep the lowest l Ritz vectors, append the residual direction
        l = std::min(m - 1, nev + std::max(1, (m - nev) / 2));
        HIPCHK(hipMemcpyAsync(s->eig_y.p, Y.data(), (size_t)m * l * 8, hipMemcpyHostToDevice, s->stream));
        RotArgs R{}; R.nv = m; R.nout = l; R.Y = s->eig_y.p; for (int i = 0; i < m; i++) R.V[i] = Vp(i);
        hipLaunchKernelGGL(k_basis_rotate, dim3(gs), dim3(256), (size_t)m * l * 8, s->stream, R, n);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(Vp(l), Vp(m), (size_t)n * 8, hipMemcpyDeviceToDevice, s->stream));
        std::fill(Hm.begin(), Hm.end(), 0.0);
        for (int i = 0; i < l; i++) Hm[(size_t)i * m + i] = theta[i];
    }
    const int nout = std::min(nev, meff);
    HIPCHK(hipMemcpyAsync(s->eig_y.p, Y.data(), (size_t)meff * nout * 8, hipMemcpyHostToDevice, s->stream));
    RotArgs R{}; R.nv = meff; R.nout = nout; R.Y = s->eig_y.p; for (int i = 0; i < meff; i++) R.V[i] = Vp(i);
    hipLaunchKernelGGL(k_basis_rotate, dim3(gs), dim3(256), (size_t)meff * nout * 8, s->stream, R, n);
    HIPCHK(hipGetLastError());
    for (int i = 0; i < nev; i++) evals[i] = i < nout ? theta[i] : 0.0;
    if (evecs_host)
        for (int i = 0; i < nout; i++) HIPCHK(hipMemcpyAsync(evecs_host + (int64_t)i * n, Vp(i), (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (nconv_out) *nconv_out = nconv;
    if (nrestart_out) *nrestart_out = it + 1;
    if (nmatvec_out) *nmatvec_out = nmv;
    return EDGPU_OK;
}

}  // namespace edgpu

// ---------------------------------------------------------------------------------------
// Sparse-matrix container with the semantics of ED_SPARSE_MATRIX.f90 (row-wise growable storage,
// duplicate-summing insert in storage order) whose product runs on the GPU.
// ---------------------------------------------------------------------------------------
struct edgpu_spmat_s {
    int64_t nrow = 0, ncol = 0;
    std::vector<std::vector<int64_t>> cols;      // 1-based columns, insertion order (ED_SPARSE_MATRIX.f90:13-17)
    std::vector<std::vector<double>> vals;
    bool dirty = true;
    edgpu::DevBuf<int64_t> d_rp; edgpu::DevBuf<int32_t> d_cols; edgpu::DevBuf<double> d_vals, d_x, d_y;
    int64_t nnz = 0;
};

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

int edgpu_abi_version(void) { return EDGPU_ABI_VERSION; }
const char *edgpu_last_error(void) { return g_err.c_str(); }

int edgpu_sector_create(const edgpu_desc *desc, edgpu_handle *out)
{
    if (!desc || !out) return fail(EDGPU_EINVAL, "edgpu_sector_create: null argument");
    *out = nullptr;
    return create(desc, out);
}
int edgpu_sector_destroy(edgpu_handle h) { destroy(h); return EDGPU_OK; }
int64_t edgpu_sector_vecdim(edgpu_handle h) { return h ? h->vecdim : -1; }

int edgpu_sector_info(edgpu_handle h, edgpu_info *o)
{
    if (!h || !o) return fail(EDGPU_EINVAL, "edgpu_sector_info: null argument");
    const SectorHost &H = h->H;
    memset(o, 0, sizeof(*o));
    o->isector = H.isector; o->Ns = H.Ns; o->Ns_Orb = H.Ns_Orb; o->Ns_Ud = H.Ns_Ud; o->DimPh = H.DimPh; o->Jhflag = H.Jhflag;
    for (int k = 0; k < H.Ns_Ud; k++) { o->DimUps[k] = H.fac[k].dim; o->DimDws[k] = H.fac[k + H.Ns_Ud].dim; }
    o->DimUp = H.DimUp; o->DimDw = H.DimDw; o->Dim = H.Dim; o->mpiQdw = H.Qdw; o->mpiIshift = H.dw0 * H.DimUp; o->vecDim = h->vecdim;
    o->nnz_total = h->nnz_total; o->bytes_stored = h->bytes_stored;
    return EDGPU_OK;
}

int edgpu_sector_map(edgpu_handle h, int32_t k, int32_t *out)
{
    if (!h || !out || k < 0 || k >= 2 * h->H.Ns_Ud) return fail(EDGPU_EINVAL, "edgpu_sector_map: bad argument");
    // read back from the device copy the kernels use
    HIPCHK(hipMemcpy(out, h->fac[k]->map.p, (size_t)h->H.fac[k].dim * 4, hipMemcpyDeviceToHost));
    return EDGPU_OK;
}

int edgpu_factor_size(edgpu_handle h, int32_t which, int32_t iud, int64_t *nrow, int64_t *nnz)
{
    if (!h || !nrow || !nnz) return fail(EDGPU_EINVAL, "edgpu_factor_size: null argument");
    const SectorHost &H = h->H;
    if (which == EDGPU_SP_UP || which == EDGPU_SP_DW) {
        if (iud < 0 || iud >= H.Ns_Ud) return fail(EDGPU_EINVAL, "edgpu_factor_size: bad iud");
        const Factor &f = H.fac[which == EDGPU_SP_UP ? iud : iud + H.Ns_Ud];
        *nrow = f.dim; *nnz = (int64_t)f.cols.size();
        return EDGPU_OK;
    }
    if (which == EDGPU_SP_ND) {
        if (!H.Jhflag) { *nrow = 0; *nnz = 0; return EDGPU_OK; }
        if (H.DimUp * H.DimDw > 50000000) return fail(EDGPU_EUNSUPPORTED, "edgpu_factor_size: spH0nd export is an inspection aid for small sectors");
        std::vector<int64_t> c; std::vector<double> v; int64_t n = 0;
        for (int64_t q = 0; q < H.Qdw; q++) for (int64_t iup = 0; iup < H.DimUp; iup++) { H.nonlocal_row(iup, H.dw0 + q, c, v); n += (int64_t)c.size(); }
        *nrow = h->nloc_el; *nnz = n;
        return EDGPU_OK;
    }
    return fail(EDGPU_EINVAL, "edgpu_factor_size: bad which");
}

int edgpu_factor_export(edgpu_handle h, int32_t which, int32_t iud, int64_t *row_ptr, int64_t *cols, double *vals)
{
    if (!h || !row_ptr) return fail(EDGPU_EINVAL, "edgpu_factor_export: null argument");
    const SectorHost &H = h->H;
    if (which == EDGPU_SP_UP || which == EDGPU_SP_DW) {
        if (iud < 0 || iud >= H.Ns_Ud) return fail(EDGPU_EINVAL, "edgpu_factor_export: bad iud");
        const Factor &f = H.fac[which == EDGPU_SP_UP ? iud : iud + H.Ns_Ud];
        DevFactor &df = *h->fac[which == EDGPU_SP_UP ? iud : iud + H.Ns_Ud];
        // values and columns come back from the device copy (what the kernels actually read)
        std::vector<int32_t> c(f.cols.size()), rp(f.dim + 1);
        HIPCHK(hipMemcpy(rp.data(), df.rowptr.p, rp.size() * 4, hipMemcpyDeviceToHost));
        if (!c.empty()) { HIPCHK(hipMemcpy(c.data(), df.cols.p, c.size() * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(vals, df.vals.p, c.size() * 8, hipMemcpyDeviceToHost)); }
        for (size_t i = 0; i < rp.size(); i++) row_ptr[i] = rp[i];
        for (size_t i = 0; i < c.size(); i++) cols[i] = (int64_t)c[i] + 1;
        return EDGPU_OK;
    }
    if (which == EDGPU_SP_ND) {
        std::vector<int64_t> c; std::vector<double> v; int64_t n = 0, r = 0;
        row_ptr[0] = 0;
        for (int64_t q = 0; q < H.Qdw; q++)
            for (int64_t iup = 0; iup < H.DimUp; iup++) {
                H.nonlocal_row(iup, H.dw0 + q, c, v);
                for (size_t k = 0; k < c.size(); k++) { cols[n] = c[k] + 1; vals[n] = v[k]; n++; }
                row_ptr[++r] = n;
            }
        return EDGPU_OK;
    }
    return fail(EDGPU_EINVAL, "edgpu_factor_export: bad which");
}

int edgpu_diag_export(edgpu_handle h, double *d_host)
{
    if (!h || !d_host) return fail(EDGPU_EINVAL, "edgpu_diag_export: null argument");
    if (!h->diag.p) { CHK(build_diag(h)); }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(d_host, h->diag.p, (size_t)h->nloc_el * 8, hipMemcpyDeviceToHost));
    return EDGPU_OK;
}

int edgpu_csr_rows(edgpu_handle h, int64_t row0, int64_t nrows, int64_t *row_ptr, int32_t *cols, double *vals, int64_t cap)
{
    if (!h || !row_ptr) return fail(EDGPU_EINVAL, "edgpu_csr_rows: null argument");
    if (!h->csr_rowptr.p) return fail(EDGPU_EINVAL, "edgpu_csr_rows: sector was not created with storage=CSR");
    if (row0 < 0 || nrows < 0 || row0 + nrows > h->nloc_el) return fail(EDGPU_EINVAL, "edgpu_csr_rows: row window out of range");
    HIPCHK(hipStreamSynchronize(h->stream));
    auto fetch = [&](DevBuf<int64_t> &rp, DevBuf<int32_t> &cc, DevBuf<double> &vv, std::vector<int64_t> &hrp, std::vector<int32_t> &hc, std::vector<double> &hv) -> int {
        hrp.resize(nrows + 1);
        HIPCHK(hipMemcpy(hrp.data(), rp.p + row0, (size_t)(nrows + 1) * 8, hipMemcpyDeviceToHost));
        int64_t n = hrp[nrows] - hrp[0];
        hc.resize(n); hv.resize(n);
        if (n) { HIPCHK(hipMemcpy(hc.data(), cc.p + hrp[0], (size_t)n * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(hv.data(), vv.p + hrp[0], (size_t)n * 8, hipMemcpyDeviceToHost)); }
        return EDGPU_OK;
    };
    std::vector<int64_t> rl, rd; std::vector<int32_t> cl, cd; std::vector<double> vl, vd;
    CHK(fetch(h->csr_rowptr, h->csr_cols, h->csr_vals, rl, cl, vl));
    if (h->csr_split) CHK(fetch(h->csrD_rowptr, h->csrD_cols, h->csrD_vals, rd, cd, vd));
    // merged view (ascending columns) of the one or two stored blocks
    int64_t n = 0;
    row_ptr[0] = 0;
    for (int64_t r = 0; r < nrows; r++) {
        int64_t a = rl[r] - rl[0], ae = rl[r + 1] - rl[0], b = 0, be = 0;
        if (h->csr_split) { b = rd[r] - rd[0]; be = rd[r + 1] - rd[0]; }
        while (a < ae || b < be) {
            bool takeL = b >= be || (a < ae && cl[a] < cd[b]);
            if (n >= cap) return fail(EDGPU_EINVAL, "edgpu_csr_rows: capacity too small");
            if (cols && vals) { cols[n] = takeL ? cl[a] : cd[b]; vals[n] = takeL ? vl[a] : vd[b]; }
            if (takeL) a++; else b++;
            n++;
        }
        row_ptr[r + 1] = n;
    }
    return EDGPU_OK;
}

int edgpu_hxv_device(edgpu_handle h, const double *v_dev, double *hv_dev)
{
    if (!h || !v_dev || !hv_dev) return fail(EDGPU_EINVAL, "edgpu_hxv_device: null argument");
    HIPCHK(hipSetDevice(h->device));
    CHK(hxv(h, v_dev, hv_dev, false, nullptr));
    HIPCHK(hipStreamSynchronize(h->stream));
    return EDGPU_OK;
}

int edgpu_hxv64(edgpu_handle h, int64_t nloc, const double *v, double *hv)
{
    if (!h || !v || !hv) return fail(EDGPU_EINVAL, "edgpu_hxv: null argument");
    if (nloc != h->vecdim) return fail(EDGPU_EINVAL, "directMatVec_cc ERROR: Nloc != dim(isector)");  // ED_HAMILTONIAN_DIRECT_HxV.f90:35
    HIPCHK(hipSetDevice(h->device));
    if (h->tmp_in.n != (size_t)nloc) { CHK(h->tmp_in.alloc((size_t)nloc)); CHK(h->tmp_out.alloc((size_t)nloc)); }
    HIPCHK(hipMemcpyAsync(h->tmp_in.p, v, (size_t)nloc * 8, hipMemcpyHostToDevice, h->stream));
    CHK(hxv(h, h->tmp_in.p, h->tmp_out.p, false, nullptr));
    HIPCHK(hipMemcpyAsync(hv, h->tmp_out.p, (size_t)nloc * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return EDGPU_OK;
}
int edgpu_hxv(edgpu_handle h, const int32_t *nloc, const double *v, double *hv)
{
    if (!nloc) return fail(EDGPU_EINVAL, "edgpu_hxv: null Nloc");
    return edgpu_hxv64(h, (int64_t)*nloc, v, hv);
}

int edgpu_hxv_time(edgpu_handle h, const double *v_dev, double *hv_dev, int32_t nwarm, int32_t nrep, float *ms_mean)
{
    if (!h || !v_dev || !hv_dev || !ms_mean || nrep < 1) return fail(EDGPU_EINVAL, "edgpu_hxv_time: bad argument");
    HIPCHK(hipSetDevice(h->device));
    for (int i = 0; i < nwarm; i++) CHK(hxv(h, v_dev, hv_dev, false, nullptr));
    HIPCHK(hipEventRecord(h->ev_a, h->stream));
    for (int i = 0; i < nrep; i++) CHK(hxv(h, v_dev, hv_dev, false, nullptr));
    HIPCHK(hipEventRecord(h->ev_b, h->stream));
    HIPCHK(hipEventSynchronize(h->ev_b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    *ms_mean = ms / nrep;
    return EDGPU_OK;
}

int edgpu_lanczos_gs_device(edgpu_handle h, int32_t nitermax, double threshold, int32_t ncheck, double *egs, double *vect_dev, int32_t *nlanc)
{
    if (!h || !egs || !vect_dev || !nlanc) return fail(EDGPU_EINVAL, "edgpu_lanczos_gs: null argument");
    HIPCHK(hipSetDevice(h->device));
    return lanczos_gs(h, nitermax, threshold, ncheck, egs, vect_dev, nlanc);
}
int edgpu_lanczos_gs(edgpu_handle h, int32_t nitermax, double threshold, int32_t ncheck, double *egs, double *vect, int32_t *nlanc)
{
    if (!h || !egs || !vect || !nlanc) return fail(EDGPU_EINVAL, "edgpu_lanczos_gs: null argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->tmp_in.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
    HIPCHK(hipMemcpy(h->tmp_in.p, vect, (size_t)h->nloc * 8, hipMemcpyHostToDevice));
    CHK(lanczos_gs(h, nitermax, threshold, ncheck, egs, h->tmp_in.p, nlanc));
    HIPCHK(hipMemcpy(vect, h->tmp_in.p, (size_t)h->nloc * 8, hipMemcpyDeviceToHost));
    return EDGPU_OK;
}
int edgpu_lanczos_tridiag_device(edgpu_handle h, const double *vin_dev, int32_t nlanc, double threshold, double *alanc, double *blanc, int32_t *nused)
{
    if (!h || !vin_dev || !alanc || !blanc || !nused) return fail(EDGPU_EINVAL, "edgpu_lanczos_tridiag: null argument");
    HIPCHK(hipSetDevice(h->device));
    return lanczos_tridiag(h, vin_dev, nlanc, threshold, alanc, blanc, nused);
}
int edgpu_lanczos_tridiag(edgpu_handle h, const double *vin, int32_t nlanc, double threshold, double *alanc, double *blanc, int32_t *nused)
{
    if (!h || !vin || !alanc || !blanc || !nused) return fail(EDGPU_EINVAL, "edgpu_lanczos_tridiag: null argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->tmp_in.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
    HIPCHK(hipMemcpy(h->tmp_in.p, vin, (size_t)h->nloc * 8, hipMemcpyHostToDevice));
    return lanczos_tridiag(h, h->tmp_in.p, nlanc, threshold, alanc, blanc, nused);
}

int edgpu_lanczos_time(edgpu_handle h, int32_t nwarm, int32_t niter, float *ms_per_iter)
{
    if (!h || !ms_per_iter || niter < 1) return fail(EDGPU_EINVAL, "edgpu_lanczos_time: bad argument");
    HIPCHK(hipSetDevice(h->device));
    CHK(lz_prepare(h, nwarm + niter + 1));
    LzVecs V{h->lz[0].p, h->lz[1].p, h->lz[2].p};
    const SectorHost &H = h->H;
    hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, h->stream, V.ucur, H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
    HIPCHK(hipGetLastError());
    CHK(lz_start(h, V));
    int k = 0;
    for (; k < nwarm; k++) CHK(lz_iter(h, V, k, nullptr));
    HIPCHK(hipEventRecord(h->ev_a, h->stream));
    for (; k < nwarm + niter; k++) CHK(lz_iter(h, V, k, nullptr));
    HIPCHK(hipEventRecord(h->ev_b, h->stream));
    HIPCHK(hipEventSynchronize(h->ev_b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    *ms_per_iter = ms / niter;
    return EDGPU_OK;
}

int edgpu_eigsh(edgpu_handle h, int32_t nev, int32_t ncv, int32_t maxiter, double tol, double *evals, double *evecs,
                int32_t *nconv, int32_t *nrestarts, int32_t *nmatvec)
{
    if (!h || !evals) return fail(EDGPU_EINVAL, "edgpu_eigsh: null argument");
    HIPCHK(hipSetDevice(h->device));
    return eigsh(h, nev, ncv, maxiter, tol, evals, evecs, nconv, nrestarts, nmatvec);
}

// off-diagonal impurity density matrix of a host state: dm[(s*Norb + a)*Norb + b], s=0 up, 1 dw (diagonal left 0:
// it follows from the impurity-configuration weights).  ed_total_ud=T, single rank.
int edgpu_imp_density_offdiag(edgpu_handle h, const double *vec_host, double *dm)
{
    if (!h || !vec_host || !dm) return fail(EDGPU_EINVAL, "edgpu_imp_density_offdiag: null argument");
    if (!h->two_factor) return fail(EDGPU_EUNSUPPORTED, "edgpu_imp_density_offdiag: ed_total_ud=F has no off-diagonal density matrix (ED_OBSERVABLES.f90:283)");
    if (h->multi && h->comm.size > 1) return fail(EDGPU_EUNSUPPORTED, "edgpu_imp_density_offdiag: single rank only");
    HIPCHK(hipSetDevice(h->device));
    const SectorHost &H = h->H; const int Norb = H.desc.Norb;
    for (int t = 0; t < 2 * Norb * Norb; t++) dm[t] = 0.0;
    if (Norb < 2) return EDGPU_OK;
    if (h->tmp_in.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
    HIPCHK(hipMemcpy(h->tmp_in.p, vec_host, (size_t)h->nloc * 8, hipMemcpyHostToDevice));
    const int64_t nrows = H.Qdw * H.DimPh;
    DevBuf<double> out; CHK(out.alloc((size_t)nrows * 50));
    HIPCHK(hipMemsetAsync(out.p, 0, (size_t)nrows * 50 * 8, h->stream));
    DmArgs A{}; A.DimUp = (int32_t)H.DimUp; A.DimDw = (int32_t)H.DimDw; A.Norb = Norb; A.Qdw = H.Qdw; A.dw0 = H.dw0; A.nrows = nrows;
    A.mapup = h->fac[0]->map.p; A.mapdw = h->fac[1]->map.p; A.rankup = h->fac[0]->rank.p; A.rankdw = h->fac[1]->rank.p; A.v = h->tmp_in.p; A.out = out.p;
    hipLaunchKernelGGL(k_imp_dm_rows, dim3((unsigned)nrows), dim3(256), 0, h->stream, A);
    HIPCHK(hipGetLastError());
    std::vector<double> part((size_t)nrows * 50);
    HIPCHK(hipMemcpyAsync(part.data(), out.p, part.size() * 8, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
    for (int64_t r = 0; r < nrows; r++)
        for (int s2 = 0; s2 < 2; s2++) for (int a = 0; a < Norb; a++) for (int b = 0; b < Norb; b++)
            dm[(s2 * Norb + a) * Norb + b] += part[(size_t)r * 50 + s2 * 25 + a * 5 + b];
    return EDGPU_OK;
}

// cvec of the twin sector from the cvec of this sector (host vectors, single rank)
int edgpu_twin_reorder(edgpu_handle h, const double *vin_host, double *vout_host)
{
    if (!h || !vin_host || !vout_host) return fail(EDGPU_EINVAL, "edgpu_twin_reorder: null argument");
    if (h->multi && h->comm.size > 1) return fail(EDGPU_EUNSUPPORTED, "edgpu_twin_reorder: single rank only");
    HIPCHK(hipSetDevice(h->device));
    const SectorHost &H = h->H;
    if (h->tmp_in.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
    HIPCHK(hipMemcpy(h->tmp_in.p, vin_host, (size_t)h->nloc * 8, hipMemcpyHostToDevice));
    TwinArgs A{}; A.nud = H.Ns_Ud; A.DimPh = H.DimPh; A.nel = h->nloc_el; A.vin = h->tmp_in.p; A.vout = h->tmp_out.p;
    for (int k = 0; k < 2 * H.Ns_Ud; k++) A.dimsA[k] = H.fac[k].dim;
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(h->nloc, 256), 8192));
    hipLaunchKernelGGL(k_twin_reorder, dim3(nb), dim3(256), 0, h->stream, A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(vout_host, h->tmp_out.p, (size_t)h->nloc * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return EDGPU_OK;
}

// Impurity-configuration weights of a state vector (host or device resident), summed over ranks.
// hist: (2^Norb * 2^Norb * DimPh) doubles, index (mu_imp*2^Norb + md_imp) + 4^Norb*iph.
int edgpu_imp_histogram(edgpu_handle h, const double *vec, int32_t vec_on_device, double *hist)
{
    if (!h || !vec || !hist) return fail(EDGPU_EINVAL, "edgpu_imp_histogram: null argument");
    HIPCHK(hipSetDevice(h->device));
    const SectorHost &H = h->H; const int nI = 1 << H.desc.Norb, nb = nI * nI * H.DimPh;
    const double *v = vec;
    if (!vec_on_device) {
        if (h->tmp_in.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
        HIPCHK(hipMemcpy(h->tmp_in.p, vec, (size_t)h->nloc * 8, hipMemcpyHostToDevice));
        v = h->tmp_in.p;
    }
    std::vector<double> acc(nb, 0.0);
    if (h->two_factor) {
        const int64_t nrows = H.Qdw * H.DimPh;
        DevBuf<double> out; CHK(out.alloc((size_t)nrows * nI));
        HistArgs A{}; A.DimUp = (int32_t)H.DimUp; A.nI = nI; A.impmask = nI - 1; A.nrows = nrows; A.mapup = h->fac[0]->map.p; A.v = v; A.out = out.p;
        hipLaunchKernelGGL(k_imp_hist_rows, dim3((unsigned)nrows), dim3(256), 0, h->stream, A);
        HIPCHK(hipGetLastError());
        std::vector<double> part((size_t)nrows * nI);
        HIPCHK(hipMemcpyAsync(part.data(), out.p, part.size() * 8, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
        for (int64_t r = 0; r < nrows; r++) {            // fixed order: deterministic
            const int64_t iph = r / H.Qdw, idw = H.dw0 + (r - iph * H.Qdw);
            const int md = H.fac[1].map[idw] & (nI - 1);
            for (int mu = 0; mu < nI; mu++) acc[(size_t)(mu * nI + md) + (size_t)nI * nI * iph] += part[(size_t)r * nI + mu];
        }
    } else {
        const int grid = 256;
        DevBuf<double> out; CHK(out.alloc((size_t)grid * nb));
        HistGenArgs A{}; A.nfac = 2 * H.Ns_Ud; A.nI = nI; A.DimPh = H.DimPh; A.Nel = h->nloc_el; A.v = v; A.out = out.p;
        for (int k = 0; k < A.nfac; k++) { A.dims[k] = H.fac[k].dim; A.maps[k] = h->fac[k]->map.p; }
        hipLaunchKernelGGL(k_imp_hist_generic, dim3(grid), dim3(256), (size_t)nb * 8, h->stream, A);
        HIPCHK(hipGetLastError());
        std::vector<double> part((size_t)grid * nb);
        HIPCHK(hipMemcpyAsync(part.data(), out.p, part.size() * 8, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
        for (int b = 0; b < grid; b++) for (int t = 0; t < nb; t++) acc[t] += part[(size_t)b * nb + t];
    }
    if (h->multi && h->comm.size > 1) {
        DevBuf<double> d; CHK(d.upload(acc));
        CHK(allreduce_dev(h, d.p, nb));
        HIPCHK(hipMemcpyAsync(acc.data(), d.p, (size_t)nb * 8, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
    }
    for (int t = 0; t < nb; t++) hist[t] = acc[t];
    return EDGPU_OK;
}

// ---- O|state> between two live sectors (GF builders, ED_GF_NORMAL.f90:190-213, 264-283) ------
int edgpu_apply_op(edgpu_handle from, edgpu_handle to, int32_t iorb, int32_t ispin, int32_t dagger,
                   const double *vin_dev, double *vout_dev, double *norm2)
{
    if (!from || !to || !vin_dev || !vout_dev || !norm2) return fail(EDGPU_EINVAL, "edgpu_apply_op: null argument");
    const SectorHost &F = from->H, &T = to->H;
    if (F.Ns_Ud != T.Ns_Ud || F.Ns_Orb != T.Ns_Orb || F.DimPh != T.DimPh || F.desc.Norb != T.desc.Norb)
        return fail(EDGPU_EINVAL, "edgpu_apply_op: sectors belong to different models");
    if (from->multi || to->multi) return fail(EDGPU_EUNSUPPORTED, "edgpu_apply_op: single rank only for now");
    if (iorb < 1 || iorb > F.desc.Norb || ispin < 1 || ispin > 2) return fail(EDGPU_EINVAL, "edgpu_apply_op: bad iorb/ispin");
    // ED_GF_NORMAL.f90:138-145: (ialfa,iorb1) = (1,iorb) for ed_total_ud else (iorb,1); ibeta = ialfa+(ispin-1)*Ns_Ud
    const int ialfa = F.desc.ed_total_ud ? 1 : iorb, iorb1 = F.desc.ed_total_ud ? iorb : 1;
    const int fb = (ialfa - 1) + (ispin - 1) * F.Ns_Ud;
    for (int k = 0; k < 2 * F.Ns_Ud; k++) {
        int want = F.fac[k].npart + ((k == fb) ? (dagger ? 1 : -1) : 0);
        if (T.fac[k].npart != want) return fail(EDGPU_EINVAL, "edgpu_apply_op: target sector is not getCDGsector/getCsector of the source");
    }
    HIPCHK(hipSetDevice(to->device));
    OpArgs A{};
    A.nfac = 2 * F.Ns_Ud; A.fb = fb; A.bitpos = iorb1 - 1; A.dagger = dagger ? 1 : 0; A.DimPh = T.DimPh;
    for (int k = 0; k < A.nfac; k++) { A.dims_to[k] = T.fac[k].dim; A.dims_from[k] = F.fac[k].dim; }
    A.map_to_fb = to->fac[fb]->map.p; A.rank_from_fb = from->fac[fb]->rank.p;
    A.nel_to = to->nloc_el; A.nel_from = from->nloc_el; A.vin = vin_dev; A.vout = vout_dev;
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(to->nloc, 256), 8192));
    HIPCHK(hipStreamSynchronize(from->stream));
    hipLaunchKernelGGL(k_apply_op, dim3(nb), dim3(256), 0, to->stream, A);
    HIPCHK(hipGetLastError());
    CHK(reduce_dot(to, vout_dev, vout_dev));                 // norm2 = <v|v>, then v /= sqrt(norm2)
    HIPCHK(hipMemcpyAsync(norm2, to->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, to->stream));
    HIPCHK(hipStreamSynchronize(to->stream));
    if (*norm2 > 0.0) {
        hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, to->stream, to->scal.p, to->alphas.p, to->betas.p, (const double *)nullptr, 0, 2);
        hipLaunchKernelGGL(k_scale, dim3(nb), dim3(256), 0, to->stream, vout_dev, to->scal.p, (int)SC_SCUR, to->nloc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(to->stream));
    }
    return EDGPU_OK;
}

// normalise a device vector in place; norm2 returned to the host (0 leaves it untouched)
static int normalise_dev(S *s, double *v_dev, double *norm2)
{
    CHK(reduce_dot(s, v_dev, v_dev));
    HIPCHK(hipMemcpyAsync(norm2, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (*norm2 > 0.0) {
        const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(s->nloc, 256), 8192));
        hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, 0, 2);
        hipLaunchKernelGGL(k_scale, dim3(nb), dim3(256), 0, s->stream, v_dev, s->scal.p, (int)SC_SCUR, s->nloc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return EDGPU_OK;
}

// susceptibility leg: diagonal impurity operator on a state of this sector, normalise, sp_lanc_tridiag
int edgpu_chi_lanczos(edgpu_handle h, const double *cup, const double *cdw, const double *state_host, int32_t nlanc,
                      double threshold, double *norm2, double *alanc, double *blanc, int32_t *nused)
{
    if (!h || !cup || !cdw || !state_host || !norm2 || !alanc || !blanc || !nused) return fail(EDGPU_EINVAL, "edgpu_chi_lanczos: null argument");
    HIPCHK(hipSetDevice(h->device));
    const SectorHost &H = h->H;
    if (h->tmp_in.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
    HIPCHK(hipMemcpy(h->tmp_in.p, state_host, (size_t)h->nloc * 8, hipMemcpyHostToDevice));
    CHK(lz_prepare(h, nlanc));
    DiagOpArgs A{};
    A.nfac = 2 * H.Ns_Ud; A.Norb = H.desc.Norb; A.total_ud = H.desc.ed_total_ud; A.Nel = h->nloc_el; A.N = h->nloc;
    if (h->multi && h->comm.size > 1 && !h->two_factor) return fail(EDGPU_EUNSUPPORTED, "edgpu_chi_lanczos: ed_total_ud=F is single rank");
    for (int k = 0; k < A.nfac; k++) { A.dims[k] = H.fac[k].dim; A.maps[k] = h->fac[k]->map.p; }
    if (h->two_factor) { A.dims[1] = (int32_t)H.Qdw; A.maps[1] = h->fac[1]->map.p + H.dw0; }   // local dw columns
    for (int a = 0; a < EDGPU_MAX_ORB; a++) { A.cu[a] = a < A.Norb ? cup[a] : 0.0; A.cd[a] = a < A.Norb ? cdw[a] : 0.0; }
    A.vin = h->tmp_in.p; A.vout = h->tmp_out.p;
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(h->nloc, 256), 8192));
    hipLaunchKernelGGL(k_apply_diagop, dim3(nb), dim3(256), 0, h->stream, A);
    HIPCHK(hipGetLastError());
    CHK(normalise_dev(h, h->tmp_out.p, norm2));
    for (int i = 0; i < nlanc; i++) { alanc[i] = 0.0; blanc[i] = 0.0; }
    *nused = 0;
    if (*norm2 <= 0.0) return EDGPU_OK;
    return lanczos_tridiag(h, h->tmp_out.p, nlanc, threshold, alanc, blanc, nused);
}

// phonon propagator leg (lanc_build_gf_phonon_main, ED_GF_PHONON.f90:40-123): (b+b^+)|state>, normalise, tridiag
int edgpu_phonon_lanczos(edgpu_handle h, const double *state_host, int32_t nlanc, double threshold, double *norm2,
                         double *alanc, double *blanc, int32_t *nused)
{
    if (!h || !state_host || !norm2 || !alanc || !blanc || !nused) return fail(EDGPU_EINVAL, "edgpu_phonon_lanczos: null argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->tmp_in.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
    HIPCHK(hipMemcpy(h->tmp_in.p, state_host, (size_t)h->nloc * 8, hipMemcpyHostToDevice));
    CHK(lz_prepare(h, nlanc));
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(h->nloc, 256), 8192));
    hipLaunchKernelGGL(k_apply_phonon_x, dim3(nb), dim3(256), 0, h->stream, (const double *)h->tmp_in.p, h->tmp_out.p, h->nloc_el, h->H.DimPh);
    HIPCHK(hipGetLastError());
    CHK(normalise_dev(h, h->tmp_out.p, norm2));
    for (int i = 0; i < nlanc; i++) { alanc[i] = 0.0; blanc[i] = 0.0; }
    *nused = 0;
    if (*norm2 <= 0.0) return EDGPU_OK;
    return lanczos_tridiag(h, h->tmp_out.p, nlanc, threshold, alanc, blanc, nused);
}

// One Green's-function Lanczos leg entirely in HBM: O|state>, normalise, sp_lanc_tridiag.
// state_host: vecdim(from) doubles on the host.  Returns norm2, alanc/blanc (nlanc) and nused.
int edgpu_gf_lanczos(edgpu_handle from, edgpu_handle to, int32_t iorb, int32_t ispin, int32_t dagger,
                     const double *state_host, int32_t nlanc, double threshold, double *norm2,
                     double *alanc, double *blanc, int32_t *nused)
{
    if (!from || !to || !state_host || !alanc || !blanc || !nused || !norm2) return fail(EDGPU_EINVAL, "edgpu_gf_lanczos: null argument");
    HIPCHK(hipSetDevice(to->device));
    if (from->tmp_in.n != (size_t)from->nloc) { CHK(from->tmp_in.alloc((size_t)from->nloc)); CHK(from->tmp_out.alloc((size_t)from->nloc)); }
    if (to->tmp_in.n != (size_t)to->nloc) { CHK(to->tmp_in.alloc((size_t)to->nloc)); CHK(to->tmp_out.alloc((size_t)to->nloc)); }
    HIPCHK(hipMemcpy(from->tmp_in.p, state_host, (size_t)from->nloc * 8, hipMemcpyHostToDevice));
    CHK(lz_prepare(to, nlanc));
    CHK(edgpu_apply_op(from, to, iorb, ispin, dagger, from->tmp_in.p, to->tmp_in.p, norm2));
    for (int i = 0; i < nlanc; i++) { alanc[i] = 0.0; blanc[i] = 0.0; }
    *nused = 0;
    if (*norm2 <= 0.0) return EDGPU_OK;
    return lanczos_tridiag(to, to->tmp_in.p, nlanc, threshold, alanc, blanc, nused);
}

// ---- step-wise Lanczos for external timing (bench.py): begin, then enqueue steps -----------
int edgpu_lanczos_bench_begin(edgpu_handle h, int32_t max_steps)
{
    if (!h || max_steps < 1) return fail(EDGPU_EINVAL, "edgpu_lanczos_bench_begin: bad argument");
    HIPCHK(hipSetDevice(h->device));
    CHK(lz_prepare(h, max_steps + 1));
    h->bv[0] = h->lz[0].p; h->bv[1] = h->lz[1].p; h->bv[2] = h->lz[2].p;
    const SectorHost &H = h->H;
    hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, h->stream, h->bv[1], H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
    HIPCHK(hipGetLastError());
    LzVecs V{h->bv[0], h->bv[1], h->bv[2]};
    CHK(lz_start(h, V));
    h->bench_k = 0;
    while ((int)h->bench_ev.size() < 2 * max_steps) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->bench_ev.push_back(e); }
    HIPCHK(hipStreamSynchronize(h->stream));
    return EDGPU_OK;
}
// enqueue `nsteps` iterations (asynchronous; caller synchronises the device).  When `timed`
// the product kernel(s) of every step are bracketed by HIP events on the launch stream.
int edgpu_lanczos_bench_steps(edgpu_handle h, int32_t nsteps, int32_t timed)
{
    if (!h || nsteps < 0) return fail(EDGPU_EINVAL, "edgpu_lanczos_bench_steps: bad argument");
    if (!h->bv[0]) return fail(EDGPU_EINVAL, "edgpu_lanczos_bench_steps: call edgpu_lanczos_bench_begin first");
    if (h->bench_k + nsteps > h->nmax - 1) return fail(EDGPU_EINVAL, "edgpu_lanczos_bench_steps: more steps than announced");
    HIPCHK(hipSetDevice(h->device));
    LzVecs V{h->bv[0], h->bv[1], h->bv[2]};
    for (int i = 0; i < nsteps; i++) {
        const int k = h->bench_k;
        if (timed) {   // lz_iter with the product bracketed by events
            HIPCHK(hipEventRecord(h->bench_ev[2 * i], h->stream));
            CHK(hxv(h, V.ucur, V.w, true, V.uprev));
            HIPCHK(hipEventRecord(h->bench_ev[2 * i + 1], h->stream));
            CHK(lz_iter_tail(h, V, k, nullptr));
        } else
            CHK(lz_iter(h, V, k, nullptr));
        h->bench_k++;
    }
    h->bv[0] = V.uprev; h->bv[1] = V.ucur; h->bv[2] = V.w;
    return EDGPU_OK;
}
// after a device synchronise: mean duration (ms) of the product over the last timed batch, and
// the Lanczos coefficients produced so far (for checks)
int edgpu_lanczos_bench_result(edgpu_handle h, int32_t nsteps_timed, float *hxv_ms_mean, double *alphas, double *betas, int32_t ncoef)
{
    if (!h) return fail(EDGPU_EINVAL, "edgpu_lanczos_bench_result: null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (hxv_ms_mean) {
        double tot = 0.0;
        for (int i = 0; i < nsteps_timed; i++) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, h->bench_ev[2 * i], h->bench_ev[2 * i + 1])); tot += ms; }
        *hxv_ms_mean = nsteps_timed > 0 ? (float)(tot / nsteps_timed) : 0.f;
    }
    if (alphas && betas && ncoef > 0) {
        if (ncoef > h->bench_k) ncoef = h->bench_k;
        HIPCHK(hipMemcpy(alphas, h->alphas.p, (size_t)ncoef * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(betas, h->betas.p + 1, (size_t)ncoef * 8, hipMemcpyDeviceToHost));
    }
    return EDGPU_OK;
}

// ---- sparse container: sp_init_matrix / sp_insert_element / sp_delete_matrix / sp_dump_matrix ---------
int edgpu_sp_init_matrix(int64_t nrow, int64_t ncol, edgpu_spmat *out)
{
    if (!out || nrow < 0 || ncol < 0) return fail(EDGPU_EINVAL, "edgpu_sp_init_matrix: bad argument");
    if (ncol >= ((int64_t)1 << 31)) return fail(EDGPU_EUNSUPPORTED, "edgpu_sp_init_matrix: Ncol must fit int32");
    auto *m = new edgpu_spmat_s();
    m->nrow = nrow; m->ncol = ncol ? ncol : nrow;
    m->cols.resize((size_t)nrow); m->vals.resize((size_t)nrow);
    *out = m;
    return EDGPU_OK;
}
int edgpu_sp_delete_matrix(edgpu_spmat m) { delete m; return EDGPU_OK; }
int edgpu_sp_insert_element(edgpu_spmat m, double value, int64_t i, int64_t j)
{
    if (!m) return fail(EDGPU_EINVAL, "sp_insert_element: matrix not allocated");
    if (i < 1 || i > m->nrow || j < 1 || j > m->ncol) return fail(EDGPU_EINVAL, "sp_insert_element: index out of range");
    auto &c = m->cols[(size_t)i - 1]; auto &v = m->vals[(size_t)i - 1];
    for (size_t k = 0; k < c.size(); k++) if (c[k] == j) { v[k] += value; m->dirty = true; return EDGPU_OK; }   // column exists: sum it up
    c.push_back(j); v.push_back(value);                                                                       // new column: append
    if ((int64_t)c.size() > m->ncol) return fail(EDGPU_EINVAL, "sp_insert_element_csr ERROR: row%Size > sparse%Ncol");
    m->dirty = true;
    return EDGPU_OK;
}
int edgpu_sp_row(edgpu_spmat m, int64_t i, int32_t *size, int64_t *cols, double *vals, int32_t cap)
{
    if (!m || !size || i < 1 || i > m->nrow) return fail(EDGPU_EINVAL, "edgpu_sp_row: bad argument");
    const auto &c = m->cols[(size_t)i - 1]; const auto &v = m->vals[(size_t)i - 1];
    *size = (int32_t)c.size();
    if (cols && vals) { if ((int32_t)c.size() > cap) return fail(EDGPU_EINVAL, "edgpu_sp_row: capacity too small"); for (size_t k = 0; k < c.size(); k++) { cols[k] = c[k]; vals[k] = v[k]; } }
    return EDGPU_OK;
}
int edgpu_sp_dump_matrix(edgpu_spmat m, double *dense)
{
    if (!m || !dense) return fail(EDGPU_EINVAL, "edgpu_sp_dump_matrix: null argument");
    for (int64_t i = 0; i < m->nrow; i++)
        for (size_t k = 0; k < m->cols[(size_t)i].size(); k++) dense[i + m->nrow * (m->cols[(size_t)i][k] - 1)] += m->vals[(size_t)i][k];   // ED_SPARSE_MATRIX.f90:399-405
    return EDGPU_OK;
}
int edgpu_sp_matvec(edgpu_spmat m, const double *v_host, double *hv_host, int32_t accumulate)
{
    if (!m || !v_host || !hv_host) return fail(EDGPU_EINVAL, "edgpu_sp_matvec: null argument");
    CHK(require_device());
    if (m->dirty) {
        std::vector<int64_t> rp((size_t)m->nrow + 1, 0); std::vector<int32_t> cc; std::vector<double> vv;
        for (int64_t i = 0; i < m->nrow; i++) {
            for (size_t k = 0; k < m->cols[(size_t)i].size(); k++) { cc.push_back((int32_t)(m->cols[(size_t)i][k] - 1)); vv.push_back(m->vals[(size_t)i][k]); }
            rp[(size_t)i + 1] = (int64_t)cc.size();
        }
        m->nnz = (int64_t)cc.size();
        if (cc.empty()) { cc.push_back(0); vv.push_back(0.0); }
        CHK(m->d_rp.upload(rp)); CHK(m->d_cols.upload(cc)); CHK(m->d_vals.upload(vv));
        CHK(m->d_x.alloc((size_t)std::max<int64_t>(m->ncol, 1))); CHK(m->d_y.alloc((size_t)std::max<int64_t>(m->nrow, 1)));
        m->dirty = false;
    }
    if (m->nrow == 0) return EDGPU_OK;
    HIPCHK(hipMemcpy(m->d_x.p, v_host, (size_t)m->ncol * 8, hipMemcpyHostToDevice));
    if (accumulate) HIPCHK(hipMemcpy(m->d_y.p, hv_host, (size_t)m->nrow * 8, hipMemcpyHostToDevice));
    const bool longrows = m->nnz > 32 * m->nrow;
    if (longrows) hipLaunchKernelGGL(k_spgen_wave, dim3((unsigned)std::min<int64_t>(cdiv(m->nrow, 4), 65536)), dim3(256), 0, 0, m->nrow, (const int64_t *)m->d_rp.p, (const int32_t *)m->d_cols.p, (const double *)m->d_vals.p, (const double *)m->d_x.p, m->d_y.p, (int)accumulate);
    else hipLaunchKernelGGL(k_spgen_lane, dim3((unsigned)std::min<int64_t>(cdiv(m->nrow, 256), 65536)), dim3(256), 0, 0, m->nrow, (const int64_t *)m->d_rp.p, (const int32_t *)m->d_cols.p, (const double *)m->d_vals.p, (const double *)m->d_x.p, m->d_y.p, (int)accumulate);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(hv_host, m->d_y.p, (size_t)m->nrow * 8, hipMemcpyDeviceToHost));
    return EDGPU_OK;
}

int edgpu_comm_unique_id(void *id128)
{
    if (!id128) return fail(EDGPU_EINVAL, "edgpu_comm_unique_id: null argument");
    if (!g_rccl.load()) return fail(EDGPU_ECOMM, "edgpu: RCCL (librccl.so) could not be loaded");
    int rc = g_rccl.GetUniqueId((NcclUid *)id128);
    if (rc) return fail(EDGPU_ECOMM, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(rc));
    return EDGPU_OK;
}
int edgpu_comm_init_rccl(edgpu_handle h, const void *id128)
{
    if (!h || !id128) return fail(EDGPU_EINVAL, "edgpu_comm_init_rccl: null argument");
    if (!g_rccl.load()) return fail(EDGPU_ECOMM, "edgpu: RCCL (librccl.so) could not be loaded");
    HIPCHK(hipSetDevice(h->device));
    NcclUid uid; memcpy(&uid, id128, sizeof(uid));
    int rc = g_rccl.CommInitRank(&h->comm.nccl, h->comm.size, uid, h->comm.rank);
    if (rc) return fail(EDGPU_ECOMM, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc));
    h->comm.kind = Comm::RCCL;
    return EDGPU_OK;
}
int edgpu_comm_init_callbacks(edgpu_handle h, edgpu_allgatherv_fn ag, edgpu_exchange_fn xch, edgpu_allreduce_fn ar, void *ctx)
{
    if (!h || !ag || !xch || !ar) return fail(EDGPU_EINVAL, "edgpu_comm_init_callbacks: null argument");
    h->comm.kind = Comm::CALLBACK; h->comm.ag = ag; h->comm.xch = xch; h->comm.ar = ar; h->comm.ctx = ctx;
    return EDGPU_OK;
}

int edgpu_device_count(int32_t *n)
{
    if (!n) return fail(EDGPU_EINVAL, "edgpu_device_count: null argument");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *n = c;
    return EDGPU_OK;
}
int edgpu_malloc(void **dev, size_t bytes) { if (!dev) return fail(EDGPU_EINVAL, "edgpu_malloc: null"); CHK(require_device()); HIPCHK(hipMalloc(dev, bytes)); return EDGPU_OK; }
int edgpu_free(void *dev) { if (dev) HIPCHK(hipFree(dev)); return EDGPU_OK; }
int edgpu_memcpy_h2d(void *dev, const void *host, size_t bytes) { HIPCHK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice)); return EDGPU_OK; }
int edgpu_memcpy_d2h(void *host, const void *dev, size_t bytes) { HIPCHK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost)); return EDGPU_OK; }
int edgpu_device_synchronize(void) { HIPCHK(hipDeviceSynchronize()); return EDGPU_OK; }

}  // extern "C"
