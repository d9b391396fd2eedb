// edgpu.hip -- libedgpu.so: sector operator + Lanczos loops on MI355X behind the C ABI of include/edgpu.h.
// This file holds the extern "C" entry points; the implementation is split over (included in this order)
//   runtime.hpp     error channel, RCCL loader, device buffers, per-sector state (edgpu_sector_s)
//   sector_host.hpp host-side sector/factor construction        kernels.hpp  gfx950 device kernels of the operator
//   basis_kernels.hpp  kernels of the multi-eigenpair solvers (multi-dot / update / rotation / Davidson step)
//   sector_dev.hpp  device images, halo lists, diagonal, explicit CSR, create/destroy
//   exchange.hpp    per-product exchange between ranks          product.hpp  hxv(): one sector product
//   lanczos.hpp     Lanczos ground state / tridiagonalisation / thick-restart eigensolver
// gfx950 only; no CPU compute path.
#include "runtime.hpp"
#include "sector_dev.hpp"
#include "exchange.hpp"
#include "product.hpp"
#include "lanczos.hpp"

using namespace edgpu;

// ---------------------------------------------------------------------------------------
// Sparse-matrix container with the semantics of ED_SPARSE_MATRIX.f90 (row-wise growable storage,
// duplicate-summing insert in storage order) whose product runs on the GPU.
// ---------------------------------------------------------------------------------------
struct edgpu_spmat_s {
    int64_t nrow = 0, ncol = 0;                  // rows held by this object (the local window in the MPI form)
    int64_t istart = 1, iend = 0, ishift = 0, nglob = 0; bool mpi = false;   // sp_set_mpi_matrix (ED_SPARSE_MATRIX.f90:574-585)
    std::vector<std::vector<int64_t>> cols;      // 1-based columns, insertion order (ED_SPARSE_MATRIX.f90:13-17)
    std::vector<std::vector<double>> vals;
    bool dirty = true;
    edgpu::DevBuf<int64_t> d_rp; edgpu::DevBuf<int32_t> d_cols; edgpu::DevBuf<double> d_vals, d_x, d_y;
    int64_t nnz = 0;
};

// A `state` argument of the leg / observable entry points may live on the host or already in HBM (e.g. the vector
// edgpu_lanczos_gs_device left there): device pointers are used in place, host pointers are staged once.
static bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain host memory: not known to HIP
    return a.type == hipMemoryTypeDevice;
}
static int stage_state(S *h, const double *state, const double **dev)
{
    if (is_device_ptr(state)) { *dev = state; return EDGPU_OK; }
    if (h->tmp_in.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
    HIPCHK(hipMemcpy(h->tmp_in.p, state, (size_t)h->nloc * 8, hipMemcpyHostToDevice));
    *dev = h->tmp_in.p;
    return EDGPU_OK;
}

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

int edgpu_abi_version(void) { return EDGPU_ABI_VERSION; }
const char *edgpu_last_error(void) { return g_err.c_str(); }
int edgpu_knob_count(void) { return kNumKnobs; }
int edgpu_knob_info(int32_t i, const char **name, int32_t *default_value, const char **kind, const char **doc, const char **covered_by)
{
    if (i < 0 || i >= kNumKnobs) return fail(EDGPU_EINVAL, "edgpu_knob_info: index out of range");
    if (name) *name = kKnobs[i].name;
    if (default_value) *default_value = kKnobs[i].dflt;
    if (kind) *kind = kKnobs[i].kind;
    if (doc) *doc = kKnobs[i].doc;
    if (covered_by) *covered_by = kKnobs[i].test;
    return EDGPU_OK;
}

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

// dense Hmat of build_Hv_sector(isector,Hmat) (ED_HAMILTONIAN.f90:128-137 -> ed_buildh_main/orbs with Hmat,
// ED_HAMILTONIAN_SPARSE_HxV.f90:132-195, 304-363): what the reference hands to LAPACK for every sector with
// dim <= lanc_dim_threshold (ED_DIAG.f90:105-107, 212-237).  Column-major Dim x Dim on the host.  The diagonal comes from
// the device kernel (k_build_diag, the reference's statement order); the factors are expanded as sp_dump_matrix +
// kronecker_product do.  With several ranks every rank fills its own rows and the matrix is summed over the ranks
// (the reference's mpi_sp_dump_matrix all-reduce, ED_SPARSE_MATRIX.f90:408-437), so every rank returns the full matrix.
int edgpu_sector_hmat(edgpu_handle h, double *hmat)
{
    if (!h || !hmat) return fail(EDGPU_EINVAL, "edgpu_sector_hmat: null argument");
    const SectorHost &H = h->H; const edgpu_desc &d = H.desc;
    if (H.Dim > 46340) return fail(EDGPU_EUNSUPPORTED, "edgpu_sector_hmat: the dense form is for sectors up to lanc_dim_threshold (Dim*Dim must fit an int32 count)");
    HIPCHK(hipSetDevice(h->device));
    if (!h->diag.p) CHK(build_diag(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<double> dg((size_t)h->nloc_el);
    if (h->nloc_el) HIPCHK(hipMemcpy(dg.data(), h->diag.p, dg.size() * 8, hipMemcpyDeviceToHost));
    const int64_t Dim = H.Dim, Nel = H.DimUp * H.DimDw, DimUp = H.DimUp;
    memset(hmat, 0, sizeof(double) * (size_t)Dim * (size_t)Dim);
#define HM(i, j) hmat[(size_t)(i) + (size_t)Dim * (size_t)(j)]
    const int nf = 2 * H.Ns_Ud;
    std::vector<int64_t> dims(nf), strd(nf); { int64_t st = 1; for (int k = 0; k < nf; k++) { dims[k] = H.fac[k].dim; strd[k] = st; st *= H.fac[k].dim; } }
    std::vector<int64_t> ncol; std::vector<double> nval;
    const int nI = 1 << d.Norb;
    for (int iph = 0; iph < H.DimPh; iph++) {
        const int64_t off = (int64_t)iph * Nel;
        for (int64_t il = 0; il < h->nloc_el; il++) {
            const int64_t ie = il + H.dw0 * DimUp, i = ie + off;          // global electronic row, global row
            HM(i, i) += dg[(size_t)il];                                   // spH0d
            // digits of the row
            int64_t c = ie; std::vector<int32_t> dgt(nf);
            for (int k = 0; k < nf; k++) { dgt[k] = (int32_t)(c % dims[k]); c /= dims[k]; }
            if (d.ed_total_ud && H.Jhflag) {                              // spH0nd (stored transposed, dumped as stored)
                H.nonlocal_row(dgt[0], dgt[1], ncol, nval);
                for (size_t q = 0; q < ncol.size(); q++) HM(i, ncol[q] + off) += nval[q];
            }
            for (int k = 0; k < nf; k++) {                                // I (x) ... H_k ... (x) I
                const Factor &f = H.fac[k];
                for (int64_t p = f.rowptr[dgt[k]]; p < f.rowptr[dgt[k] + 1]; p++) HM(i, ie + (int64_t)(f.cols[p] - dgt[k]) * strd[k] + off) += f.vals[p];
            }
            if (H.DimPh > 1) {                                            // stored/H_ph.f90:2-5, H_e_ph.f90:2-39
                HM(i, i) += d.w0_ph * iph;
                uint32_t mu = 0, md = 0;
                if (d.ed_total_ud) { mu = (uint32_t)H.fac[0].map[dgt[0]] & (nI - 1); md = (uint32_t)H.fac[1].map[dgt[1]] & (nI - 1); }
                else for (int a = 0; a < H.Ns_Ud; a++) { mu |= ((uint32_t)H.fac[a].map[dgt[a]] & 1u) << a; md |= ((uint32_t)H.fac[a + H.Ns_Ud].map[dgt[a + H.Ns_Ud]] & 1u) << a; }
                double g = 0.0;
                for (int a = 0; a < d.Norb; a++) g = g + d.g_ph[a] * ((double)(((mu >> a) & 1) + ((md >> a) & 1)) - 1.0);
                if (iph + 1 < H.DimPh) HM(i, ie + (int64_t)(iph + 1) * Nel) += g * sqrt((double)(iph + 1));
                if (iph > 0) HM(i, ie + (int64_t)(iph - 1) * Nel) += g * sqrt((double)iph);
            }
        }
    }
#undef HM
    if (h->multi && h->comm.size > 1) {
        const int64_t n2 = Dim * Dim;
        if (h->comm.kind == Comm::CALLBACK) { if (h->comm.ar(h->comm.ctx, hmat, (int32_t)n2)) return fail(EDGPU_ECOMM, "edgpu: allreduce callback failed"); }
        else {
            DevBuf<double> b; CHK(b.alloc((size_t)n2));
            HIPCHK(hipMemcpy(b.p, hmat, (size_t)n2 * 8, hipMemcpyHostToDevice));
            CHK(allreduce_dev(h, b.p, (int)n2));
            HIPCHK(hipStreamSynchronize(h->stream));
            HIPCHK(hipMemcpy(hmat, b.p, (size_t)n2 * 8, hipMemcpyDeviceToHost));
        }
    }
    return EDGPU_OK;
}

// rows [row0, row0+nrows) of the ELECTRONIC operator (one phonon block): merged view of the stored blocks, ascending columns
static int csr_rows_el(edgpu_handle h, int64_t row0, int64_t nrows, int64_t *row_ptr, int32_t *cols, double *vals, int64_t cap)
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
    // blocks stored lane-major per 64-row block (k_csr_il): fetch the row blocks that cover the window and undo the interleave
    auto fetch_il = [&](DevBuf<int64_t> &rp, DevBuf<int32_t> &cc, DevBuf<double> &vv, std::vector<int64_t> &hrp, std::vector<int32_t> &hc, std::vector<double> &hv) -> int {
        const int64_t DU = h->H.DimUp;
        auto blk0 = [&](int64_t r) { const int64_t q = r / DU, i = r - q * DU; return q * DU + (i / 64) * 64; };
        auto blkn = [&](int64_t r) { const int64_t q = r / DU, i = r - q * DU; return std::min<int64_t>(64, DU - (i / 64) * 64); };
        hrp.assign((size_t)nrows + 1, 0);
        if (nrows == 0) { HIPCHK(hipMemcpy(hrp.data(), rp.p + row0, 8, hipMemcpyDeviceToHost)); return EDGPU_OK; }
        const int64_t rA = blk0(row0), rl = row0 + nrows - 1, rB = blk0(rl) + blkn(rl);
        std::vector<int64_t> w((size_t)(rB - rA) + 1);
        HIPCHK(hipMemcpy(w.data(), rp.p + rA, w.size() * 8, hipMemcpyDeviceToHost));
        const int64_t m = w.back() - w[0];
        std::vector<int32_t> tc((size_t)m); std::vector<double> tv((size_t)m);
        if (m) { HIPCHK(hipMemcpy(tc.data(), cc.p + w[0], (size_t)m * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(tv.data(), vv.p + w[0], (size_t)m * 8, hipMemcpyDeviceToHost)); }
        for (int64_t r = 0; r <= nrows; r++) hrp[r] = w[row0 + r - rA];
        hc.resize((size_t)(hrp[nrows] - hrp[0])); hv.resize(hc.size());
        for (int64_t r = 0; r < nrows; r++) {
            // level e of a block holds the rows with more than e entries, packed in row order (equal lengths: e*rows + row)
            const int64_t g = row0 + r, b0 = blk0(g), nr = blkn(g), len = hrp[r + 1] - hrp[r], base = w[b0 - rA] - w[0];
            int64_t lev = base;
            for (int64_t e = 0; e < len; e++) {
                int64_t cnt = 0, before = 0;
                for (int64_t i = 0; i < nr; i++) { const bool has = w[b0 - rA + i + 1] - w[b0 - rA + i] > e; cnt += has; before += has && (b0 + i < g); }
                hc[hrp[r] - hrp[0] + e] = tc[lev + before]; hv[hrp[r] - hrp[0] + e] = tv[lev + before];
                lev += cnt;
            }
        }
        return EDGPU_OK;
    };
    // ... and stored tile-major (dw-type blocks): block (jb, idw) at jb*64*ctot + rows(jb)*cpre[idw]; one copy per block touched
    auto fetch_tm = [&](DevBuf<int64_t> &rp, DevBuf<int32_t> &cc, DevBuf<double> &vv, const std::vector<int32_t> &cnt, std::vector<int64_t> &hrp, std::vector<int32_t> &hc, std::vector<double> &hv) -> int {
        const int64_t DU = h->H.DimUp;
        std::vector<int64_t> cpre(cnt.size() + 1, 0);
        for (size_t q = 0; q < cnt.size(); q++) cpre[q + 1] = cpre[q] + cnt[q];
        const int64_t ctot = cpre.back();
        hrp.resize((size_t)nrows + 1);
        HIPCHK(hipMemcpy(hrp.data(), rp.p + row0, (size_t)(nrows + 1) * 8, hipMemcpyDeviceToHost));
        hc.resize((size_t)(hrp[nrows] - hrp[0])); hv.resize(hc.size());
        std::vector<int32_t> tc; std::vector<double> tv; int64_t have = -1;
        for (int64_t r = 0; r < nrows; r++) {
            const int64_t g = row0 + r, q = g / DU, i = g - q * DU, i0 = (i / 64) * 64, nr = std::min<int64_t>(64, DU - i0), len = cnt[q];
            const int64_t base = i0 * ctot + nr * cpre[q];
            if (base != have && len > 0) {
                tc.resize((size_t)(nr * len)); tv.resize(tc.size()); have = base;
                HIPCHK(hipMemcpy(tc.data(), cc.p + base, tc.size() * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(tv.data(), vv.p + base, tv.size() * 8, hipMemcpyDeviceToHost));
            }
            if (hrp[r + 1] - hrp[r] != len) return fail(EDGPU_EINVAL, "edgpu_csr_rows: internal: row length mismatch in a tile-major block");
            for (int64_t e = 0; e < len; e++) { hc[hrp[r] - hrp[0] + e] = tc[e * nr + (i - i0)]; hv[hrp[r] - hrp[0] + e] = tv[e * nr + (i - i0)]; }
        }
        return EDGPU_OK;
    };
    auto fetch_dw = [&](DevBuf<int64_t> &rp, DevBuf<int32_t> &cc, DevBuf<double> &vv, const std::vector<int32_t> &cnt, std::vector<int64_t> &hrp, std::vector<int32_t> &hc, std::vector<double> &hv) -> int {
        if (h->csr_tm) return fetch_tm(rp, cc, vv, cnt, hrp, hc, hv);
        if (h->csr_il_dw) return fetch_il(rp, cc, vv, hrp, hc, hv);
        return fetch(rp, cc, vv, hrp, hc, hv);
    };
    // merged view (ascending columns) of the one, two or three stored blocks
    std::vector<int64_t> rp[4]; std::vector<int32_t> cc[4]; std::vector<double> vv[4];
    int nblk = 1;
    if (h->csr_split && h->csr_il_loc) CHK(fetch_il(h->csr_rowptr, h->csr_cols, h->csr_vals, rp[0], cc[0], vv[0]));
    else CHK(fetch(h->csr_rowptr, h->csr_cols, h->csr_vals, rp[0], cc[0], vv[0]));
    if (h->csr_split) { CHK(fetch_dw(h->csrD_rowptr, h->csrD_cols, h->csrD_vals, h->h_cntD, rp[1], cc[1], vv[1])); nblk = 2; }
    if (h->csr_three) { CHK(fetch_dw(h->csrO_rowptr, h->csrO_cols, h->csrO_vals, h->h_cntO, rp[2], cc[2], vv[2])); nblk = 3; }
    const int nstored = nblk;      // blocks whose columns live in the (possibly pitched) vector space
    if (h->nd_nr > 0) {     // non-local block (compressed rows): expand the window's part into a row-pointer form
        auto at = [&](const int64_t *p, int64_t i, int64_t &o) -> int { HIPCHK(hipMemcpy(&o, p + i, 8, hipMemcpyDeviceToHost)); return EDGPU_OK; };
        auto lower = [&](int64_t key, int64_t &pos) -> int {       // first non-empty row >= key
            int64_t a = 0, b = h->nd_nr;
            while (a < b) { const int64_t m = (a + b) / 2; int64_t r; CHK(at(h->nd_rows.p, m, r)); if (r < key) a = m + 1; else b = m; }
            pos = a; return EDGPU_OK;
        };
        int64_t i0, i1; CHK(lower(row0, i0)); CHK(lower(row0 + nrows, i1));
        std::vector<int64_t> rr((size_t)(i1 - i0)), pp((size_t)(i1 - i0) + 1);
        if (i1 > i0) HIPCHK(hipMemcpy(rr.data(), h->nd_rows.p + i0, (size_t)(i1 - i0) * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(pp.data(), h->nd_ptr.p + i0, (size_t)(i1 - i0 + 1) * 8, hipMemcpyDeviceToHost));
        const int64_t ne = pp.back() - pp[0];
        cc[nblk].resize(ne); vv[nblk].resize(ne);
        if (ne) { HIPCHK(hipMemcpy(cc[nblk].data(), h->nd_cols.p + pp[0], (size_t)ne * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(vv[nblk].data(), h->nd_vals.p + pp[0], (size_t)ne * 8, hipMemcpyDeviceToHost)); }
        rp[nblk].assign((size_t)nrows + 1, 0);
        for (size_t k = 0; k < rr.size(); k++) rp[nblk][rr[k] - row0 + 1] = pp[k + 1] - pp[k];
        for (int64_t r = 0; r < nrows; r++) rp[nblk][r + 1] += rp[nblk][r];
        nblk++;
    }
    int64_t n = 0;
    row_ptr[0] = 0;
    for (int64_t r = 0; r < nrows; r++) {
        int64_t a[4], ae[4];
        for (int b = 0; b < nblk; b++) { a[b] = rp[b][r] - rp[b][0]; ae[b] = rp[b][r + 1] - rp[b][0]; }
        for (;;) {
            int pick = -1;
            for (int b = 0; b < nblk; b++) if (a[b] < ae[b] && (pick < 0 || cc[b][a[b]] < cc[pick][a[pick]])) pick = b;
            if (pick < 0) break;
            if (n >= cap) return fail(EDGPU_EINVAL, "edgpu_csr_rows: capacity too small");
            if (cols && vals) { const int64_t c = cc[pick][a[pick]]; cols[n] = pick < nstored && h->csr_ld > h->H.DimUp ? (int32_t)((c / h->csr_ld) * h->H.DimUp + c % h->csr_ld) : (int32_t)c; vals[n] = vv[pick][a[pick]]; }
            a[pick]++; n++;
        }
        row_ptr[r + 1] = n;
    }
    return EDGPU_OK;
}

int edgpu_csr_rows(edgpu_handle h, int64_t row0, int64_t nrows, int64_t *row_ptr, int32_t *cols, double *vals, int64_t cap)
{
    if (!h || !row_ptr) return fail(EDGPU_EINVAL, "edgpu_csr_rows: null argument");
    if (!h->csr_rowptr.p) return fail(EDGPU_EINVAL, "edgpu_csr_rows: sector was not created with storage=CSR");
    if (h->H.DimPh == 1) return csr_rows_el(h, row0, nrows, row_ptr, cols, vals, cap);
    // phonon sectors: row r = iel + iph*nloc_el is the electronic row iel shifted into block iph (+ w0*iph on its diagonal)
    // plus the stored e-ph entries g(iel) sqrt(.) in the neighbouring phonon blocks
    if (row0 < 0 || nrows < 0 || row0 + nrows > h->nloc) return fail(EDGPU_EINVAL, "edgpu_csr_rows: row window out of range");
    if (h->multi) return fail(EDGPU_EUNSUPPORTED, "edgpu_csr_rows: phonon sectors are exported on one rank only");
    const int64_t nel = h->nloc_el; const double w0 = h->H.desc.w0_ph;
    int64_t n = 0; row_ptr[0] = 0;
    std::vector<int32_t> ec(4096); std::vector<double> ev(4096);
    for (int64_t r = 0; r < nrows; r++) {
        const int64_t row = row0 + r, iph = row / nel, iel = row - iph * nel;
        int64_t rp2[2];
        CHK(csr_rows_el(h, iel, 1, rp2, ec.data(), ev.data(), (int64_t)ec.size()));
        double g = 0.0; HIPCHK(hipMemcpy(&g, h->csr_gel.p + iel, 8, hipMemcpyDeviceToHost));
        struct E { int64_t c; double v; }; std::vector<E> ent;
        bool has_diag = false;
        for (int64_t k = 0; k < rp2[1]; k++) { double val = ev[k]; if (ec[k] == iel) { val += w0 * iph; has_diag = true; } ent.push_back({ec[k] + iph * nel, val}); }
        if (!has_diag && w0 * iph != 0.0) ent.push_back({row, w0 * iph});
        if (iph + 1 < h->H.DimPh && g != 0.0) ent.push_back({iel + (iph + 1) * nel, g * std::sqrt((double)(iph + 1))});
        if (iph > 0 && g != 0.0) ent.push_back({iel + (iph - 1) * nel, g * std::sqrt((double)iph)});
        std::sort(ent.begin(), ent.end(), [](const E &x, const E &y) { return x.c < y.c; });
        for (auto &e : ent) {
            if (n >= cap) return fail(EDGPU_EINVAL, "edgpu_csr_rows: capacity too small");
            if (cols && vals) { cols[n] = (int32_t)e.c; vals[n] = e.v; }
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
    // explicit CSR with pitched columns: the caller's dense vectors pass through pitched copies; time the product itself on them
    const bool pit = h->H.desc.storage == EDGPU_STORE_CSR && h->csr_ld > h->H.DimUp;
    for (int i = 0; i < nwarm + (pit ? 1 : 0); i++) CHK(hxv(h, v_dev, hv_dev, false, nullptr));
    HIPCHK(hipEventRecord(h->ev_a, h->stream));
    for (int i = 0; i < nrep; i++) { if (pit) CHK(hxv(h, h->csr_xs.p, h->csr_ys.p, false, nullptr, h->csr_ld)); else CHK(hxv(h, v_dev, hv_dev, false, nullptr)); }
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
    LzF F{h->lz[1].p, h->lz[0].p, h->lz[2].p};
    const bool fused = fused_ok(h);
    const SectorHost &H = h->H;
    if (fused) {      // the start vector is generated in the dense layout and copied into the (possibly pitched) work vector
        if (h->lz[3].n != (size_t)h->nloc) CHK(h->lz[3].alloc((size_t)h->nloc));
        hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, h->stream, h->lz[3].p, H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
        HIPCHK(hipGetLastError());
        CHK(lzf_copy_in(h, F.u, h->lz[3].p));
        HIPCHK(hipMemsetAsync(F.uprev, 0, (size_t)h->vlen * 8, h->stream)); CHK(lzf_begin(h, F, nullptr));
    } else {
        if (h->lz[3].n != (size_t)h->nloc) CHK(h->lz[3].alloc((size_t)h->nloc));
        hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, h->stream, h->lz[3].p, H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
        HIPCHK(hipGetLastError());
        CHK(lzf_copy_in(h, V.ucur, h->lz[3].p));
        if (h->pad_on) HIPCHK(hipMemsetAsync(V.w, 0, (size_t)h->vlen * 8, h->stream));
        CHK(lz_start(h, V));
    }
    int k = 0;
    for (; k < nwarm; k++) { if (fused) CHK(lzf_iter(h, F, F.uprev, nullptr)); else CHK(lz_iter(h, V, k, nullptr)); }
    HIPCHK(hipEventRecord(h->ev_a, h->stream));
    for (; k < nwarm + niter; k++) { if (fused) CHK(lzf_iter(h, F, F.uprev, nullptr)); else CHK(lz_iter(h, V, k, nullptr)); }
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

int32_t edgpu_eigsh_ncv_cap(void) { return kMaxBasisBig; }
int32_t edgpu_eigsh_ncv_used(edgpu_handle h) { return h ? h->eig_ncv_used : 0; }
int32_t edgpu_lanczos_is_fused(edgpu_handle h) { return h && fused_ok(h) ? 1 : 0; }
int edgpu_sector_describe(edgpu_handle h, char *buf, int32_t cap)
{
    if (!h || !buf || cap < 1) return fail(EDGPU_EINVAL, "edgpu_sector_describe: bad argument");
    const SectorHost &H = h->H; const edgpu_desc &d = H.desc;
    std::string t = d.storage == EDGPU_STORE_CSR ? "storage=csr" : (d.storage == EDGPU_STORE_DIRECT ? "storage=direct" : "storage=kron");
    if (d.storage == EDGPU_STORE_CSR) {
        t += h->csr_three ? " blocks=local+own-dw+received-dw" : (h->csr_split ? " blocks=local+dw" : " blocks=one");
        if (h->csr_split && h->csr_uni) t += " dw-rowptr=factored";
        if (h->csr_split && (h->csr_il_loc || h->csr_il_dw)) t += std::string(" lane-major-blocks=") + (h->csr_il_loc ? "local" : "") + (h->csr_il_loc && h->csr_il_dw ? "+" : "") + (h->csr_il_dw ? (h->csr_tm ? "dw(tile-major)" : "dw") : "");
        if (h->nd_nr > 0) t += "+nonlocal(compressed rows)";
        if (H.DimPh > 1) t += "+phonon(stored g)";
    } else if (h->kron2) {
        const bool nd_pass = h->has_nd && !h->has_ph && h->nd_split;
        const bool split = h->multi || (h->kron_split && (!h->extra || nd_pass));
        t += split ? " product=two-pass" : " product=one-pass(k_kron2)";
        if (split) {
            if (h->use_lds) t += fused_form(h) == 2 ? " local=pairs" : (h->upf->packed && h->upf->width <= 16 && h->up_persistent ? " local=persistent" : " local=row-kernel");
            else t += h->upf->seg_ok && h->up_halo ? " local=segment+halo(nseg=" + std::to_string(h->upf->nseg) + ")" : " local=long-row-gather";
            if (nd_pass) t += "+nonlocal-pass";
        }
    } else t += " product=generic";
    if (h->composite) t += " composite(ed_total_ud=F)";
    t += fused_ok(h) ? " lanczos=fused" : " lanczos=three-pass";
    t += h->pad_on ? " pitch=" + std::to_string(h->vld) : " pitch=dense";
    if (h->multi) t += h->tr_on ? " exchange=transpose" : (h->halo ? " exchange=needed-rows" : " exchange=all-gather");
    snprintf(buf, (size_t)cap, "%s", t.c_str());
    return EDGPU_OK;
}

int edgpu_dvdson(edgpu_handle h, int32_t nev, int32_t maxiter, double tol, double *evals, double *evecs, int32_t *nconv, int32_t *nmatvec)
{
    if (!h || !evals) return fail(EDGPU_EINVAL, "edgpu_dvdson: null argument");
    HIPCHK(hipSetDevice(h->device));
    return dvdson(h, nev, maxiter, tol, evals, evecs, nconv, nmatvec);
}

// off-diagonal impurity density matrix of a host state: dm[(s*Norb + a)*Norb + b], s=0 up, 1 dw (diagonal left 0:
// it follows from the impurity-configuration weights).  ed_total_ud=T, single rank.
int edgpu_imp_density_offdiag(edgpu_handle h, const double *vec_host, double *dm)
{
    if (!h || !vec_host || !dm) return fail(EDGPU_EINVAL, "edgpu_imp_density_offdiag: null argument");
    if (!h->two_factor) return fail(EDGPU_EUNSUPPORTED, "edgpu_imp_density_offdiag: ed_total_ud=F has no off-diagonal density matrix (ED_OBSERVABLES.f90:283)");
    HIPCHK(hipSetDevice(h->device));
    const SectorHost &H = h->H; const int Norb = H.desc.Norb;
    for (int t = 0; t < 2 * Norb * Norb; t++) dm[t] = 0.0;
    if (Norb < 2) return EDGPU_OK;
    const double *st = nullptr;
    CHK(stage_state(h, vec_host, &st));
    const int64_t nrows = H.Qdw * H.DimPh;
    DevBuf<double> out; CHK(out.alloc((size_t)nrows * 50));
    HIPCHK(hipMemsetAsync(out.p, 0, (size_t)nrows * 50 * 8, h->stream));
    DmArgs A{}; A.DimUp = (int32_t)H.DimUp; A.DimDw = (int32_t)H.DimDw; A.Norb = Norb; A.Qdw = H.Qdw; A.dw0 = H.dw0; A.nrows = nrows;
    A.mapup = h->fac[0]->map.p; A.mapdw = h->fac[1]->map.p; A.rankup = h->fac[0]->rank.p; A.rankdw = h->fac[1]->rank.p; A.v = st; A.out = out.p;
    const bool dist = h->multi && h->comm.size > 1;
    DevBuf<double> full;
    if (dist) {   // the dw-spin partner rows belong to other ranks
        CHK(full.alloc((size_t)(H.DimUp * H.DimDw * H.DimPh)));
        CHK(gather_full(h, st, full.p, h->stream));
        A.vg = full.p;
    }
    hipLaunchKernelGGL(k_imp_dm_rows, dim3((unsigned)nrows), dim3(256), 0, h->stream, A);
    HIPCHK(hipGetLastError());
    std::vector<double> part((size_t)nrows * 50);
    HIPCHK(hipMemcpyAsync(part.data(), out.p, part.size() * 8, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
    for (int64_t r = 0; r < nrows; r++)
        for (int s2 = 0; s2 < 2; s2++) for (int a = 0; a < Norb; a++) for (int b = 0; b < Norb; b++)
            dm[(s2 * Norb + a) * Norb + b] += part[(size_t)r * 50 + s2 * 25 + a * 5 + b];
    if (dist) {
        std::vector<double> acc(dm, dm + 2 * Norb * Norb);
        DevBuf<double> d; CHK(d.upload(acc));
        CHK(allreduce_dev(h, d.p, 2 * Norb * Norb));
        HIPCHK(hipMemcpyAsync(dm, d.p, (size_t)2 * Norb * Norb * 8, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
    }
    return EDGPU_OK;
}

// cvec of the twin sector from the cvec of this sector (host vectors).  With more than one rank: vin is this
// rank's shard of sector A, vout its shard of the twin sector B (dw split of DimDw_B = DimUp_A); the state is
// all-gathered once, vout must hold DimDw_A * Qdw_B * DimPh doubles, Qdw_B = this rank's share of DimUp_A
// (same rule as ED_HAMILTONIAN.f90:96-110).
int edgpu_twin_reorder(edgpu_handle h, const double *vin, double *vout)
{
    if (!h || !vin || !vout) return fail(EDGPU_EINVAL, "edgpu_twin_reorder: null argument");
    if (vin == vout) return fail(EDGPU_EINVAL, "edgpu_twin_reorder: the reorder is out of place, vin == vout");
    HIPCHK(hipSetDevice(h->device));
    const SectorHost &H = h->H;
    const bool in_dev = is_device_ptr(vin), out_dev = is_device_ptr(vout);
    if (!(in_dev && out_dev) && h->tmp_in.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
    if (!in_dev) HIPCHK(hipMemcpy(h->tmp_in.p, vin, (size_t)h->nloc * 8, hipMemcpyHostToDevice));
    const double *src = in_dev ? vin : h->tmp_in.p;
    const bool dist = h->multi && h->comm.size > 1;
    const int64_t nelG = H.DimUp * H.DimDw;
    TwinArgs A{}; A.nud = H.Ns_Ud; A.DimPh = H.DimPh; A.nel = dist ? nelG : h->nloc_el; A.nel_out = h->nloc_el; A.row0_out = 0;
    A.vin = src; A.vout = out_dev ? vout : h->tmp_out.p;
    for (int k = 0; k < 2 * H.Ns_Ud; k++) A.dimsA[k] = H.fac[k].dim;
    DevBuf<double> full, outb;
    int64_t nout = h->nloc;
    if (dist) {
        CHK(full.alloc((size_t)(nelG * H.DimPh)));
        CHK(gather_full(h, src, full.p, h->stream));
        // shard of the twin sector: DimUp_B = DimDw_A rows long, dw columns of B = up index of A split over the ranks
        const int64_t P = h->comm.size, r = h->comm.rank, q = H.DimUp / P, rem = H.DimUp % P;
        if (H.DimUp < P) return fail(EDGPU_EUNSUPPORTED, "edgpu_twin_reorder: twin sector has fewer dw columns than ranks");
        const int64_t qb = r < rem ? q + 1 : q, b0 = r < rem ? r * (q + 1) : r * q + rem;
        A.vin = full.p; A.nel_out = H.DimDw * qb; A.row0_out = H.DimDw * b0;
        nout = A.nel_out * H.DimPh;
        if (!out_dev) { CHK(outb.alloc((size_t)nout)); A.vout = outb.p; }
    }
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(nout, 256), 8192));
    hipLaunchKernelGGL(k_twin_reorder, dim3(nb), dim3(256), 0, h->stream, A);
    HIPCHK(hipGetLastError());
    if (!out_dev) HIPCHK(hipMemcpyAsync(vout, A.vout, (size_t)nout * 8, hipMemcpyDeviceToHost, h->stream));
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
        HistGenArgs A{}; A.nfac = 2 * H.Ns_Ud; A.nI = nI; A.DimPh = H.DimPh; A.Nel = h->nloc_el; A.row0 = H.dw0 * H.DimUp; A.v = v; A.out = out.p;
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

// Local energy terms of one state (lanc_local_energy, ED_OBSERVABLES.f90:372-616), weight 1 (the host applies
// peso = exp(-beta(Ei-Egs))/Z and sums over its state list):
//   out[0] ed_Epot (interaction energy INCLUDING the Hartree shift, as the routine leaves it at 589)   out[1] ed_Ehartree
//   out[2] ed_Eknot   out[3] ed_Dust   out[4] ed_Dund   out[5] ed_Dse   out[6] ed_Dph
// The density terms are sums over the impurity-configuration weights (edgpu_imp_histogram), the off-diagonal impHloc part
// of Eknot over the impurity density matrix (edgpu_imp_density_offdiag), Dse / Dph one more pass over the state
// (k_nd_expect).  ed_total_ud=F: density terms only, as in the reference (the other blocks sit under if(ed_total_ud)).
int edgpu_local_energy(edgpu_handle h, const double *state_host, double *out7)
{
    if (!h || !state_host || !out7) return fail(EDGPU_EINVAL, "edgpu_local_energy: null argument");
    HIPCHK(hipSetDevice(h->device));
    const SectorHost &H = h->H; const edgpu_desc &d = H.desc;
    const int Norb = d.Norb, nI = 1 << Norb;
    const double *st = nullptr;
    CHK(stage_state(h, state_host, &st));
    std::vector<double> hist((size_t)nI * nI * H.DimPh);
    CHK(edgpu_imp_histogram(h, st, 1, hist.data()));
    double Epot = 0.0, Ehar = 0.0, Eknot = 0.0, Dust = 0.0, Dund = 0.0, Dse = 0.0, Dph = 0.0;
    for (int iph = 0; iph < H.DimPh; iph++)
        for (int mu = 0; mu < nI; mu++)
            for (int md = 0; md < nI; md++) {
                const double w = hist[(size_t)(mu * nI + md) + (size_t)nI * nI * iph];
                if (w == 0.0) continue;
                auto nu = [&](int a) { return (double)((mu >> a) & 1); };
                auto nd = [&](int a) { return (double)((md >> a) & 1); };
                for (int a = 0; a < Norb; a++) {
                    Eknot += H.imp(1, 1, a + 1, a + 1) * nu(a) * w + H.imp(d.Nspin, d.Nspin, a + 1, a + 1) * nd(a) * w;
                    Epot += d.Uloc[a] * nu(a) * nd(a) * w;
                    if (d.hfmode) Ehar += (-0.5 * d.Uloc[a] * (nu(a) + nd(a)) + 0.25 * d.Uloc[a]) * w;
                }
                for (int a = 0; a < Norb; a++)
                    for (int b = a + 1; b < Norb; b++) {
                        const double ust = nu(a) * nd(b) + nu(b) * nd(a), und = nu(a) * nu(b) + nd(a) * nd(b), n4 = nu(a) + nd(a) + nu(b) + nd(b);
                        Epot += d.Ust * ust * w + (d.Ust - d.Jh) * und * w;
                        Dust += ust * w; Dund += und * w;
                        if (d.hfmode) Ehar += (-0.5 * d.Ust * n4 + 0.25 * d.Ust - 0.5 * (d.Ust - d.Jh) * n4 + 0.25 * (d.Ust - d.Jh)) * w;
                    }
            }
    if (h->two_factor && Norb > 1) {
        bool offd = false;
        for (int a = 1; a <= Norb; a++) for (int b = 1; b <= Norb; b++) if (a != b && (H.imp(1, 1, a, b) != 0.0 || H.imp(d.Nspin, d.Nspin, a, b) != 0.0)) offd = true;
        if (offd) {
            std::vector<double> dm((size_t)2 * Norb * Norb);
            CHK(edgpu_imp_density_offdiag(h, st, dm.data()));
            for (int a = 0; a < Norb; a++) for (int b = 0; b < Norb; b++) if (a != b)
                Eknot += H.imp(1, 1, a + 1, b + 1) * dm[(size_t)(0 * Norb + a) * Norb + b] + H.imp(d.Nspin, d.Nspin, a + 1, b + 1) * dm[(size_t)(1 * Norb + a) * Norb + b];
        }
        if (H.Jhflag && (d.Jx != 0.0 || d.Jp != 0.0)) {
            const int64_t nrows = H.Qdw * H.DimPh;
            DevBuf<double> o; CHK(o.alloc((size_t)nrows * 2));
            NdExpArgs A{}; A.DimUp = (int32_t)H.DimUp; A.DimDw = (int32_t)H.DimDw; A.Norb = Norb; A.Qdw = H.Qdw; A.dw0 = H.dw0;
            A.mapup = h->fac[0]->map.p; A.mapdw = h->fac[1]->map.p; A.rankup = h->fac[0]->rank.p; A.rankdw = h->fac[1]->rank.p;
            A.do_se = d.Jx != 0.0; A.do_ph = d.Jp != 0.0; A.v = st; A.vfull = st; A.full_rows = H.DimDw; A.out = o.p;
            const bool dist = h->multi && h->comm.size > 1;
            DevBuf<double> full;
            if (dist) { CHK(full.alloc((size_t)(H.DimUp * H.DimDw * H.DimPh))); CHK(gather_full(h, st, full.p, h->stream)); A.vfull = full.p; }
            hipLaunchKernelGGL(k_nd_expect, dim3((unsigned)nrows), dim3(256), 0, h->stream, A);
            HIPCHK(hipGetLastError());
            std::vector<double> part((size_t)nrows * 2);
            HIPCHK(hipMemcpyAsync(part.data(), o.p, part.size() * 8, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
            double acc[2] = {0.0, 0.0};
            for (int64_t r = 0; r < nrows; r++) { acc[0] += part[(size_t)r * 2]; acc[1] += part[(size_t)r * 2 + 1]; }
            if (dist) {
                std::vector<double> a2(acc, acc + 2); DevBuf<double> dd; CHK(dd.upload(a2)); CHK(allreduce_dev(h, dd.p, 2));
                HIPCHK(hipMemcpyAsync(acc, dd.p, 16, hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream));
            }
            Dse = acc[0]; Dph = acc[1];
            Epot += d.Jx * Dse + d.Jp * Dph;
        }
    }
    out7[0] = Epot + Ehar; out7[1] = Ehar; out7[2] = Eknot; out7[3] = Dust; out7[4] = Dund; out7[5] = Dse; out7[6] = Dph;
    return EDGPU_OK;
}

// mean duration (ms) of the exchange step over the last timed batch (multi-rank; 0 on one rank): HIP events on the comm stream
int edgpu_lanczos_bench_comm_ms(edgpu_handle h, int32_t nsteps_timed, float *ms_mean)
{
    if (!h || !ms_mean) return fail(EDGPU_EINVAL, "edgpu_lanczos_bench_comm_ms: null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(hipStreamSynchronize(h->cstream));
    double tot = 0.0; int cnt = 0;
    for (int i = 0; i < nsteps_timed && 2 * i + 1 < (int)h->comm_ev.size(); i++) { float ms = 0.f; if (hipEventElapsedTime(&ms, h->comm_ev[2 * i], h->comm_ev[2 * i + 1]) == hipSuccess) { tot += ms; cnt++; } else (void)hipGetLastError(); }
    *ms_mean = cnt ? (float)(tot / cnt) : 0.f;
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
    if (from->multi != to->multi || from->comm.size != to->comm.size || from->comm.rank != to->comm.rank)
        return fail(EDGPU_EINVAL, "edgpu_apply_op: the two sectors must live on the same rank layout");
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
    DevBuf<double> full;
    if (to->multi) {
        // target rows are the rank's dw columns of the TARGET sector.  An up-spin operator leaves the dw
        // index alone and both sectors split the same DimDw the same way: the source entry is local.  A
        // dw-spin operator couples to dw columns of other ranks: all-gather the source state first.
        A.row0_to = T.dw0 * T.DimUp;
        if (fb < F.Ns_Ud) { A.src_off = F.dw0 * F.DimUp; if (F.DimDw != T.DimDw || F.dw0 != T.dw0) return fail(EDGPU_EINVAL, "edgpu_apply_op: partitions differ"); }
        else {
            CHK(full.alloc((size_t)(F.DimUp * F.DimDw * F.DimPh)));
            HIPCHK(hipSetDevice(from->device));
            CHK(gather_full(from, vin_dev, full.p, from->stream));
            A.vin = full.p; A.nel_from = F.DimUp * F.DimDw; A.src_off = 0;
        }
    }
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
    const double *st = nullptr;
    if (h->tmp_out.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
    CHK(stage_state(h, state_host, &st));
    CHK(lz_prepare(h, nlanc));
    DiagOpArgs A{};
    A.nfac = 2 * H.Ns_Ud; A.Norb = H.desc.Norb; A.total_ud = H.desc.ed_total_ud; A.Nel = h->nloc_el; A.N = h->nloc;
    for (int k = 0; k < A.nfac; k++) { A.dims[k] = H.fac[k].dim; A.maps[k] = h->fac[k]->map.p; }
    if (h->two_factor) { A.dims[1] = (int32_t)H.Qdw; A.maps[1] = h->fac[1]->map.p + H.dw0; }   // local dw columns
    else A.row0 = H.dw0 * H.DimUp;                                                              // mixed-radix digits are global
    for (int a = 0; a < EDGPU_MAX_ORB; a++) { A.cu[a] = a < A.Norb ? cup[a] : 0.0; A.cd[a] = a < A.Norb ? cdw[a] : 0.0; }
    A.vin = st; A.vout = h->tmp_out.p;
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
    const double *st = nullptr;
    if (h->tmp_out.n != (size_t)h->nloc) { CHK(h->tmp_in.alloc((size_t)h->nloc)); CHK(h->tmp_out.alloc((size_t)h->nloc)); }
    CHK(stage_state(h, state_host, &st));
    CHK(lz_prepare(h, nlanc));
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(h->nloc, 256), 8192));
    hipLaunchKernelGGL(k_apply_phonon_x, dim3(nb), dim3(256), 0, h->stream, st, h->tmp_out.p, h->nloc_el, h->H.DimPh);
    HIPCHK(hipGetLastError());
    CHK(normalise_dev(h, h->tmp_out.p, norm2));
    for (int i = 0; i < nlanc; i++) { alanc[i] = 0.0; blanc[i] = 0.0; }
    *nused = 0;
    if (*norm2 <= 0.0) return EDGPU_OK;
    return lanczos_tridiag(h, h->tmp_out.p, nlanc, threshold, alanc, blanc, nused);
}

// One Green's-function Lanczos leg entirely in HBM: O|state>, normalise, sp_lanc_tridiag.
// state_host: vecdim(from) doubles, on the host or already in HBM.  Returns norm2, alanc/blanc (nlanc) and nused.
int edgpu_gf_lanczos(edgpu_handle from, edgpu_handle to, int32_t iorb, int32_t ispin, int32_t dagger,
                     const double *state_host, int32_t nlanc, double threshold, double *norm2,
                     double *alanc, double *blanc, int32_t *nused)
{
    if (!from || !to || !state_host || !alanc || !blanc || !nused || !norm2) return fail(EDGPU_EINVAL, "edgpu_gf_lanczos: null argument");
    HIPCHK(hipSetDevice(to->device));
    const double *st = nullptr;
    CHK(stage_state(from, state_host, &st));
    if (to->tmp_in.n != (size_t)to->nloc) { CHK(to->tmp_in.alloc((size_t)to->nloc)); CHK(to->tmp_out.alloc((size_t)to->nloc)); }
    CHK(lz_prepare(to, nlanc));
    CHK(edgpu_apply_op(from, to, iorb, ispin, dagger, st, to->tmp_in.p, norm2));
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
    LzVecs V{h->bv[0], h->bv[1], h->bv[2]};
    h->bench_fused = fused_ok(h);
    if (h->bench_fused) {      // roles: bv[0] = u_{k-1}, bv[1] = u_k, bv[2] = local part
        LzF F{h->bv[1], h->bv[0], h->bv[2]};
        if (h->lz[3].n != (size_t)h->nloc) CHK(h->lz[3].alloc((size_t)h->nloc));
        hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, h->stream, h->lz[3].p, H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
        HIPCHK(hipGetLastError());
        CHK(lzf_copy_in(h, F.u, h->lz[3].p));
        HIPCHK(hipMemsetAsync(F.uprev, 0, (size_t)h->vlen * 8, h->stream));
        CHK(lzf_begin(h, F, nullptr));
    } else {
        if (h->lz[3].n != (size_t)h->nloc) CHK(h->lz[3].alloc((size_t)h->nloc));
        hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, h->stream, h->lz[3].p, H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
        HIPCHK(hipGetLastError());
        CHK(lzf_copy_in(h, h->bv[1], h->lz[3].p));
        if (h->pad_on) HIPCHK(hipMemsetAsync(h->bv[2], 0, (size_t)h->vlen * 8, h->stream));
        CHK(lz_start(h, V));
    }
    h->bench_k = 0;
    while ((int)h->bench_ev.size() < 2 * max_steps) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->bench_ev.push_back(e); }
    if (h->multi) while ((int)h->comm_ev.size() < 2 * max_steps) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->comm_ev.push_back(e); }
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
    if (h->bench_fused) {      // a step = dw half of iteration k + local half of k+1 (which holds the axpy and the norm): all of it is "product"
        LzF F{h->bv[1], h->bv[0], h->bv[2]};
        for (int i = 0; i < nsteps; i++) {
            if (timed) HIPCHK(hipEventRecord(h->bench_ev[2 * i], h->stream));
            CHK(lzf_iter(h, F, F.uprev, nullptr));
            if (timed) HIPCHK(hipEventRecord(h->bench_ev[2 * i + 1], h->stream));
            h->bench_k++;
        }
        h->bv[0] = F.uprev; h->bv[1] = F.u; h->bv[2] = F.t;
        return EDGPU_OK;
    }
    for (int i = 0; i < nsteps; i++) {
        const int k = h->bench_k;
        if (timed) {   // lz_iter with the product bracketed by events
            HIPCHK(hipEventRecord(h->bench_ev[2 * i], h->stream));
            if (!h->multi) h->fuse_mode = 0;          // as lz_iter: the alpha step rides on the product's final reduction
            h->comm_slot = i;
            const int rc = hxv(h, V.ucur, V.w, true, V.uprev, h->vld);
            h->fuse_mode = -1; h->comm_slot = -1;
            if (rc) return rc;
            HIPCHK(hipEventRecord(h->bench_ev[2 * i + 1], h->stream));
            CHK(lz_iter_tail(h, V, k, nullptr, !h->multi));
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
        std::memcpy(alphas, h->alphas.h, (size_t)ncoef * 8);      // host memory (CoefBuf), the stream was synchronised above
        std::memcpy(betas, h->betas.h + 1, (size_t)ncoef * 8);
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
    m->istart = 1; m->iend = nrow; m->ishift = 0; m->nglob = nrow;
    *out = m;
    return EDGPU_OK;
}
int edgpu_sp_init_matrix_mpi(int64_t istart, int64_t iend, int64_t ishift, int64_t n, int64_t ncol, edgpu_spmat *out)
{
    if (!out || n < 0 || istart < 1 || iend > n || iend < istart - 1 || ishift != istart - 1)
        return fail(EDGPU_EINVAL, "edgpu_sp_init_matrix_mpi: bad row window (need 1 <= istart, iend <= N, ishift = istart-1)");
    CHK(edgpu_sp_init_matrix(iend - istart + 1, ncol ? ncol : n, out));     // Nloc = iend-istart+1 rows (ED_SPARSE_MATRIX.f90:167-169)
    (*out)->istart = istart; (*out)->iend = iend; (*out)->ishift = ishift; (*out)->nglob = n; (*out)->mpi = true;
    return EDGPU_OK;
}
int edgpu_sp_load_matrix(const double *dense, int64_t nrow, int64_t ncol, edgpu_spmat *inout)
{
    if (!dense || !inout || nrow < 0 || ncol < 0) return fail(EDGPU_EINVAL, "edgpu_sp_load_matrix: bad argument");
    if (*inout) { delete *inout; *inout = nullptr; }                        // if(sparse%status)call sp_delete_matrix
    CHK(edgpu_sp_init_matrix(nrow, ncol, inout));
    for (int64_t i = 1; i <= nrow; i++)
        for (int64_t j = 1; j <= ncol; j++) { const double x = dense[(i - 1) + nrow * (j - 1)]; if (x != 0.0) CHK(edgpu_sp_insert_element(*inout, x, i, j)); }
    return EDGPU_OK;
}
int edgpu_sp_delete_matrix(edgpu_spmat m) { delete m; return EDGPU_OK; }
int edgpu_sp_insert_element(edgpu_spmat m, double value, int64_t i, int64_t j)
{
    if (!m) return fail(EDGPU_EINVAL, "sp_insert_element: matrix not allocated");
    if (i < m->istart || i > m->iend || j < 1 || j > m->ncol) return fail(EDGPU_EINVAL, "sp_insert_element: index out of range");
    i -= m->ishift;                                                          // row => sparse%row(i-sparse%Ishift)
    auto &c = m->cols[(size_t)i - 1]; auto &v = m->vals[(size_t)i - 1];
    for (size_t k = 0; k < c.size(); k++) if (c[k] == j) { v[k] += value; m->dirty = true; return EDGPU_OK; }   // column exists: sum it up
    c.push_back(j); v.push_back(value);                                                                       // new column: append
    if ((int64_t)c.size() > m->ncol) return fail(EDGPU_EINVAL, "sp_insert_element_csr ERROR: row%Size > sparse%Ncol");
    m->dirty = true;
    return EDGPU_OK;
}
int edgpu_sp_row(edgpu_spmat m, int64_t i, int32_t *size, int64_t *cols, double *vals, int32_t cap)
{
    if (!m || !size || i < m->istart || i > m->iend) return fail(EDGPU_EINVAL, "edgpu_sp_row: bad argument");
    i -= m->ishift;
    const auto &c = m->cols[(size_t)i - 1]; const auto &v = m->vals[(size_t)i - 1];
    *size = (int32_t)c.size();
    if (cols && vals) { if ((int32_t)c.size() > cap) return fail(EDGPU_EINVAL, "edgpu_sp_row: capacity too small"); for (size_t k = 0; k < c.size(); k++) { cols[k] = c[k]; vals[k] = v[k]; } }
    return EDGPU_OK;
}
int edgpu_sp_dump_matrix(edgpu_spmat m, double *dense)
{
    if (!m || !dense) return fail(EDGPU_EINVAL, "edgpu_sp_dump_matrix: null argument");
    for (int64_t i = 0; i < m->nrow; i++)
        for (size_t k = 0; k < m->cols[(size_t)i].size(); k++) dense[(i + m->ishift) + m->nglob * (m->cols[(size_t)i][k] - 1)] += m->vals[(size_t)i][k];   // ED_SPARSE_MATRIX.f90:399-405, 425-430
    return EDGPU_OK;
}
// stored entries -> device CSR (once after the last insert); the product kernels then read only device memory
static int sp_sync_device(edgpu_spmat m)
{
    if (!m->dirty) return EDGPU_OK;
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
    return EDGPU_OK;
}
static int sp_launch(edgpu_spmat m, const double *x_dev, double *y_dev, int accumulate, hipStream_t st)
{
    const bool longrows = m->nnz > 32 * m->nrow;
    if (longrows) hipLaunchKernelGGL(k_spgen_wave, dim3((unsigned)std::min<int64_t>(cdiv(m->nrow, 4), 65536)), dim3(256), 0, st, m->nrow, (const int64_t *)m->d_rp.p, (const int32_t *)m->d_cols.p, (const double *)m->d_vals.p, x_dev, y_dev, accumulate);
    else hipLaunchKernelGGL(k_spgen_lane, dim3((unsigned)std::min<int64_t>(cdiv(m->nrow, 256), 65536)), dim3(256), 0, st, m->nrow, (const int64_t *)m->d_rp.p, (const int32_t *)m->d_cols.p, (const double *)m->d_vals.p, x_dev, y_dev, accumulate);
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}
int edgpu_sp_matvec(edgpu_spmat m, const double *v_host, double *hv_host, int32_t accumulate)
{
    if (!m || !v_host || !hv_host) return fail(EDGPU_EINVAL, "edgpu_sp_matvec: null argument");
    CHK(require_device());
    CHK(sp_sync_device(m));
    if (m->nrow == 0) return EDGPU_OK;
    HIPCHK(hipMemcpy(m->d_x.p, v_host, (size_t)m->ncol * 8, hipMemcpyHostToDevice));
    if (accumulate) HIPCHK(hipMemcpy(m->d_y.p, hv_host, (size_t)m->nrow * 8, hipMemcpyHostToDevice));
    CHK(sp_launch(m, m->d_x.p, m->d_y.p, (int)accumulate, nullptr));
    HIPCHK(hipMemcpy(hv_host, m->d_y.p, (size_t)m->nrow * 8, hipMemcpyDeviceToHost));
    return EDGPU_OK;
}
int edgpu_sp_matvec_device(edgpu_spmat m, const double *v_dev, double *hv_dev, int32_t accumulate, void *hip_stream)
{
    if (!m || !v_dev || !hv_dev) return fail(EDGPU_EINVAL, "edgpu_sp_matvec_device: null argument");
    CHK(require_device());
    CHK(sp_sync_device(m));
    if (m->nrow == 0) return EDGPU_OK;
    return sp_launch(m, v_dev, hv_dev, (int)accumulate, (hipStream_t)hip_stream);
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
    h->comm.kind = Comm::RCCL; h->comm.owned = true; g_comm_created++;
    return EDGPU_OK;
}

// ---- the process-wide communicator: created once, attached to every sector -----------------------------
int edgpu_comm_create_rccl(const void *id128, int32_t rank, int32_t size, int32_t device, edgpu_comm *out)
{
    if (!id128 || !out || size < 1 || rank < 0 || rank >= size) return fail(EDGPU_EINVAL, "edgpu_comm_create_rccl: bad argument");
    *out = nullptr;
    CHK(require_device());
    if (!g_rccl.load()) return fail(EDGPU_ECOMM, "edgpu: RCCL (librccl.so) could not be loaded");
    if (device >= 0) HIPCHK(hipSetDevice(device));
    auto c = std::unique_ptr<edgpu_comm_s>(new edgpu_comm_s());
    HIPCHK(hipGetDevice(&c->device));
    NcclUid uid; memcpy(&uid, id128, sizeof(uid));
    int rc = g_rccl.CommInitRank(&c->c.nccl, size, uid, rank);
    if (rc) return fail(EDGPU_ECOMM, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc));
    c->c.kind = Comm::RCCL; c->c.rank = rank; c->c.size = size; g_comm_created++;
    *out = c.release();
    return EDGPU_OK;
}
int edgpu_comm_create_callbacks(int32_t rank, int32_t size, edgpu_allgatherv_fn ag, edgpu_exchange_fn xch, edgpu_allreduce_fn ar, void *ctx, edgpu_comm *out)
{
    if (!out || !ag || !xch || !ar || size < 1 || rank < 0 || rank >= size) return fail(EDGPU_EINVAL, "edgpu_comm_create_callbacks: bad argument");
    auto *c = new edgpu_comm_s();
    c->c.kind = Comm::CALLBACK; c->c.rank = rank; c->c.size = size; c->c.ag = ag; c->c.xch = xch; c->c.ar = ar; c->c.ctx = ctx;
    (void)hipGetDevice(&c->device); (void)hipGetLastError();
    *out = c;
    return EDGPU_OK;
}
int edgpu_comm_create_emulated(int32_t rank, int32_t size, edgpu_comm *out)
{
    if (!out || size < 1 || rank < 0 || rank >= size) return fail(EDGPU_EINVAL, "edgpu_comm_create_emulated: bad argument");
    auto *c = new edgpu_comm_s();
    c->c.kind = Comm::EMULATED; c->c.rank = rank; c->c.size = size;
    (void)hipGetDevice(&c->device); (void)hipGetLastError();
    *out = c;
    return EDGPU_OK;
}
// Sub-communicator of the first `nranks` ranks (ED_HAMILTONIAN.f90:65-93: sectors with DimDw < MpiSize run on the first
// DimDw ranks).  COLLECTIVE over the parent: every rank calls it; *is_member = 0 and *out = NULL on the ranks left out.
// The result is cached in the parent: one split per distinct size, however many sectors use it.
int edgpu_comm_sub(edgpu_comm parent, int32_t nranks, edgpu_comm *out, int32_t *is_member)
{
    if (!parent || !out || nranks < 1 || nranks > parent->c.size) return fail(EDGPU_EINVAL, "edgpu_comm_sub: bad argument");
    *out = nullptr;
    if (is_member) *is_member = parent->c.rank < nranks ? 1 : 0;
    if (nranks == parent->c.size) { *out = parent; return EDGPU_OK; }
    auto it = parent->subs.find(nranks);
    if (it != parent->subs.end()) { *out = it->second->member ? it->second : nullptr; return EDGPU_OK; }
    if (parent->c.kind != Comm::RCCL) return fail(EDGPU_EUNSUPPORTED, "edgpu_comm_sub: a callback communicator is split by the host that owns it (create one per group with edgpu_comm_create_callbacks)");
    if (!g_rccl.CommSplit) return fail(EDGPU_EUNSUPPORTED, "edgpu_comm_sub: this RCCL has no ncclCommSplit");
    HIPCHK(hipSetDevice(parent->device));
    auto *s = new edgpu_comm_s();
    s->device = parent->device; s->member = parent->c.rank < nranks;
    int rc = g_rccl.CommSplit(parent->c.nccl, s->member ? 0 : -1 /* NCCL_SPLIT_NOCOLOR */, parent->c.rank, &s->c.nccl, nullptr);
    if (rc) { delete s; return fail(EDGPU_ECOMM, std::string("ncclCommSplit: ") + g_rccl.GetErrorString(rc)); }
    s->c.kind = Comm::RCCL; s->c.rank = parent->c.rank; s->c.size = nranks;
    if (s->member) g_comm_created++;
    parent->subs[nranks] = s;
    *out = s->member ? s : nullptr;
    return EDGPU_OK;
}
int edgpu_comm_destroy(edgpu_comm c)
{
    if (!c) return EDGPU_OK;
    for (auto &kv : c->subs) { if (kv.second->c.nccl && g_rccl.ok) g_rccl.CommDestroy(kv.second->c.nccl); delete kv.second; }
    if (c->c.kind == Comm::RCCL && c->c.nccl && g_rccl.ok) g_rccl.CommDestroy(c->c.nccl);
    delete c;
    return EDGPU_OK;
}
int edgpu_comm_count(int32_t *n) { if (!n) return fail(EDGPU_EINVAL, "edgpu_comm_count: null argument"); *n = g_comm_created; return EDGPU_OK; }
// attach: the sector uses the communicator (which must outlive it); rank and size must be those of the descriptor
int edgpu_sector_attach_comm(edgpu_handle h, edgpu_comm c)
{
    if (!h || !c) return fail(EDGPU_EINVAL, "edgpu_sector_attach_comm: null argument");
    if (c->c.size != h->comm.size || c->c.rank != h->comm.rank) return fail(EDGPU_EINVAL, "edgpu_sector_attach_comm: rank/size of the communicator differ from the sector's mpi_rank/mpi_size (shrunk sectors take edgpu_comm_sub)");
    if (h->comm.kind == Comm::RCCL && h->comm.owned && h->comm.nccl && g_rccl.ok) g_rccl.CommDestroy(h->comm.nccl);
    h->comm = c->c; h->comm.owned = false;
    return EDGPU_OK;
}
// in-place sum of n host doubles over the ranks of the communicator (diagnostics: bench.py counts its ranks with it)
int edgpu_comm_allreduce_sum(edgpu_comm c, double *buf_host, int32_t n)
{
    if (!c || !buf_host || n < 1) return fail(EDGPU_EINVAL, "edgpu_comm_allreduce_sum: bad argument");
    if (c->c.kind == Comm::CALLBACK) { if (c->c.ar(c->c.ctx, buf_host, n)) return fail(EDGPU_ECOMM, "edgpu: allreduce callback failed"); return EDGPU_OK; }
    if (c->c.kind == Comm::EMULATED) return EDGPU_OK;      // the rank's own values stand for the sum
    if (c->c.kind != Comm::RCCL) return fail(EDGPU_ECOMM, "edgpu_comm_allreduce_sum: no communicator");
    HIPCHK(hipSetDevice(c->device));
    DevBuf<double> d; CHK(d.alloc((size_t)n));
    HIPCHK(hipMemcpy(d.p, buf_host, (size_t)n * 8, hipMemcpyHostToDevice));
    int rc = g_rccl.AllReduce(d.p, d.p, (size_t)n, kNcclDouble, kNcclSum, c->c.nccl, (hipStream_t)0);
    if (rc) return fail(EDGPU_ECOMM, std::string("edgpu: RCCL allreduce failed: ") + g_rccl.GetErrorString(rc));
    HIPCHK(hipStreamSynchronize((hipStream_t)0));
    HIPCHK(hipMemcpy(buf_host, d.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return EDGPU_OK;
}
// What one product exchanges: mode 0 none (one rank), 1 needed dw rows only, 2 all-gather of the vector, 3 distributed
// transpose (two all-to-alls); send_bytes[r] / recv_bytes[r] = bytes this rank sends to / receives from rank r per product.
int edgpu_sector_exchange_info(edgpu_handle h, int32_t *mode, int64_t *send_bytes, int64_t *recv_bytes)
{
    if (!h || !mode) return fail(EDGPU_EINVAL, "edgpu_sector_exchange_info: null argument");
    const SectorHost &H = h->H; const int P = h->comm.size, me = h->comm.rank;
    *mode = !h->multi ? 0 : (h->tr_on ? 3 : (h->halo ? 1 : 2));
    if (!send_bytes || !recv_bytes) return EDGPU_OK;
    auto split = [&](int64_t D, int r, int64_t &q) { const int64_t qq = D / P, rem = D % P; q = r < rem ? qq + 1 : qq; };
    for (int r = 0; r < P; r++) {
        int64_t sb = 0, rb = 0;
        if (r != me && h->multi) {
            if (h->tr_on) { int64_t qu, qd; split(H.DimUp, r, qu); split(H.DimDw, r, qd); sb = 2 * qu * H.Qdw * 8; rb = 2 * h->Qup * qd * 8; }
            else if (h->halo) { sb = h->send_rows_cnt[r] * xspace_ld(h) * H.DimPh * 8; rb = h->recv_rows_cnt[r] * xspace_ld(h) * H.DimPh * 8; }
            else { int64_t qd; split(H.DimDw, r, qd); sb = h->nloc * 8; rb = qd * H.DimUp * H.DimPh * 8; }
        }
        send_bytes[r] = sb; recv_bytes[r] = rb;
    }
    return EDGPU_OK;
}
int edgpu_sector_relay_info(edgpu_handle h, int32_t *enabled, int32_t *nchunks, int64_t *busiest_before, int64_t *busiest_after, int64_t *rows_direct, int64_t *rows_routed)
{
    if (!h) return fail(EDGPU_EINVAL, "edgpu_sector_relay_info: null handle");
    const size_t PP = (size_t)h->comm.size * h->comm.size;
    if (enabled) *enabled = h->relay_on ? 1 : 0;
    if (nchunks) *nchunks = h->relay_on ? h->relay_chunks : 0;
    int64_t mx = 0; for (int64_t v : h->xch_rows) mx = std::max(mx, v);
    if (busiest_before) *busiest_before = mx;
    if (busiest_after) *busiest_after = h->relay_on ? h->relay_after : mx;
    if (rows_direct) for (size_t i = 0; i < PP; i++) rows_direct[i] = i < h->xch_rows.size() ? h->xch_rows[i] : 0;
    if (rows_routed) for (size_t i = 0; i < PP; i++) rows_routed[i] = h->relay_on ? h->relay_link_rows[i] : (i < h->xch_rows.size() ? h->xch_rows[i] : 0);
    return EDGPU_OK;
}
int edgpu_comm_init_callbacks(edgpu_handle h, edgpu_allgatherv_fn ag, edgpu_exchange_fn xch, edgpu_allreduce_fn ar, void *ctx)
{
    if (!h || !ag || !xch || !ar) return fail(EDGPU_EINVAL, "edgpu_comm_init_callbacks: null argument");
    h->comm.kind = Comm::CALLBACK; h->comm.ag = ag; h->comm.xch = xch; h->comm.ar = ar; h->comm.ctx = ctx;
    return EDGPU_OK;
}

int edgpu_cache_release(void) { kpool_release(); g_streams.drain(); g_pinned.drain(); return EDGPU_OK; }

int edgpu_device_count(int32_t *n)
{
    if (!n) return fail(EDGPU_EINVAL, "edgpu_device_count: null argument");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *n = c;
    return EDGPU_OK;
}
int edgpu_malloc(void **dev, size_t bytes)
{
    if (!dev) return fail(EDGPU_EINVAL, "edgpu_malloc: null");
    CHK(require_device());
    hipError_t e = hipMalloc(dev, bytes);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); kpool_release(); e = hipMalloc(dev, bytes); }   // cached work vectors go first
    HIPCHK(e);
    return EDGPU_OK;
}
int edgpu_free(void *dev) { if (dev) HIPCHK(hipFree(dev)); return EDGPU_OK; }
int edgpu_memcpy_h2d(void *dev, const void *host, size_t bytes) { HIPCHK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice)); return EDGPU_OK; }
int edgpu_memcpy_d2h(void *host, const void *dev, size_t bytes) { HIPCHK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost)); return EDGPU_OK; }
int edgpu_memset(void *dev, int32_t byte_value, size_t bytes) { HIPCHK(hipMemset(dev, byte_value, bytes)); return EDGPU_OK; }
int edgpu_device_synchronize(void) { HIPCHK(hipDeviceSynchronize()); return EDGPU_OK; }

}  // extern "C"
