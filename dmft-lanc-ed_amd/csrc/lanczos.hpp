// lanczos.hpp -- device-resident Lanczos (ground state, tridiagonalisation) and the thick-restart multi-eigenpair solver
// Part of libedgpu.so (one translation unit: edgpu.hip includes these headers in order).
#pragma once
#include "tridiag_host.hpp"

namespace edgpu {

// ---------------------------------------------------------------------------------------
// Lanczos loop, device resident.  Restates SciFortran's lanczos_iteration / sp_lanc_eigh /
// sp_lanc_tridiag (module SF_SP_LINALG, not part of the reference tree; call sites
// ED_DIAG.f90:177-185, ED_GF_NORMAL.f90:232-237) with vectors kept unnormalised in HBM and the
// 1/beta scale carried in a device scalar, so one iteration is: fused product+alpha, one
// axpy+norm pass, two single-workgroup reductions -- and no host synchronisation.
// ---------------------------------------------------------------------------------------
static int lz_prepare(S *s, int nmax, int nvec = 3)
{
    for (int i = 0; i < nvec; i++) if (s->lz[i].n != (size_t)s->vlen) CHK(s->lz[i].alloc((size_t)s->vlen));      // vlen: padded length when the fused loops pad the row pitch
    if (s->nmax < nmax) {
        CHK(s->alphas.alloc((size_t)nmax + 2)); CHK(s->betas.alloc((size_t)nmax + 2)); CHK(s->zvec.alloc((size_t)nmax + 2));
        s->nmax = nmax;
    }
    HIPCHK(hipStreamSynchronize(s->stream));      // an earlier loop's kernels may still be writing the coefficient arrays (host memory)
    std::memset(s->alphas.h, 0, ((size_t)nmax + 2) * 8); std::memset(s->betas.h, 0, ((size_t)nmax + 2) * 8);
    return EDGPU_OK;
}

static int reduce_dot(S *s, const double *a, const double *b, int64_t n = -1)
{
    if (n < 0) n = s->nloc;
    int nb = (int)std::min<int64_t>(cdiv(n, 256 * 8), 2048);
    nb = std::max(nb, 1);
    hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(256), 0, s->stream, a, b, n, s->partial.p);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, s->partial.p, (int64_t)nb, s->scal.p + SC_TMP);
    HIPCHK(hipGetLastError());
    return allreduce_tmp(s);
}

struct LzVecs { double *uprev, *ucur, *w; };

// start: ucur holds the (unnormalised) start vector
static int lz_start(S *s, LzVecs &V)
{
    s->npend = false; s->sr_pending = false;
    HIPCHK(hipMemsetAsync(V.uprev, 0, (size_t)s->vlen * 8, s->stream));
    CHK(reduce_dot(s, V.ucur, V.ucur, s->vlen));
    hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, 0, 2);
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}

// iteration k (0-based): fills alphas[k], betas[k+1]; optional eigenvector accumulation
static int lz_iter_tail(S *s, LzVecs &V, int k, double *acc, bool alpha_done = false)
{
    if (!alpha_done) {
        hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, k, 0);
        if (acc) hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)s->zvec.p, k, 3);
    }
    int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(s->vlen, 256 * 4), (int64_t)s->axpy_nb));      // measured at c3: 512 blocks 0.70 ms, 1024 0.73, 4096 0.82-0.88, one pair per lane 0.76
    hipLaunchKernelGGL(k_axpy_norm, dim3(nb), dim3(256), 0, s->stream, V.w, V.ucur, acc, s->scal.p, s->vlen, s->partial.p);      // padding columns are zero and stay zero
    if (!s->multi) {   // no all-reduce between the sum and the step that uses it: one launch
        hipLaunchKernelGGL(k_reduce_final_step, dim3(1), dim3(1024), 0, s->stream, s->partial.p, (int64_t)nb, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, 1);
        HIPCHK(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, s->partial.p, (int64_t)nb, s->scal.p + SC_TMP);
        HIPCHK(hipGetLastError());
        CHK(allreduce_tmp(s));
        hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, k, 1);
        HIPCHK(hipGetLastError());
    }
    double *t = V.uprev; V.uprev = V.ucur; V.ucur = V.w; V.w = t;
    return EDGPU_OK;
}
// Several ranks, one all-reduce per iteration (EDGPU_ONE_ALLREDUCE=1; the plain loops switch it on around their iterations).
// alpha_k needs the finished product and beta_k the vector after the alpha update, so the two sums cannot be taken together as
// they stand.  Here the norm of the newest vector stays a per-rank partial sum: the next product runs on the unnormalised vector
// with neutral scales, its dot u_k.H u_k and the pending |u_k|^2 are summed in ONE 16-byte all-reduce, the scales follow, and a
// three-vector pass forms r_k = s w - alpha s u_k - beta s_prev u_{k-1} (and the next pending norm).  Same traffic as the
// two-all-reduce form (the u_prev stream moves from the product's epilogue to the update pass), alpha in the v.Hv form
// (without the -beta v.v_prev term, which is rounding noise), beta_k known one product later: callers that look at beta on the host
// either test it one iteration late (lanczos_gs) or flush the pending norm first (lz_flush).
static int lz_iter_deferred(S *s, LzVecs &V, int k, double *acc)
{
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(s->vlen, 256 * 4), (int64_t)s->axpy_nb));
    if (!s->npend) {      // scales known (first iteration, or after a flush): product with its epilogue, alpha summed by itself
        CHK(hxv(s, V.ucur, V.w, true, V.uprev, s->vld));
        hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, k, 0);
        if (acc) hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)s->zvec.p, k, 3);
        hipLaunchKernelGGL(k_axpy_norm, dim3(nb), dim3(256), 0, s->stream, V.w, V.ucur, acc, s->scal.p, s->vlen, s->partial.p);
        HIPCHK(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, k, 10);
        HIPCHK(hipGetLastError());
        s->defer_ar = true;
        const int rc = hxv(s, V.ucur, V.w, true, V.uprev, s->vld);
        s->defer_ar = false;
        if (rc) return rc;
        hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, k, 13);
        HIPCHK(hipGetLastError());
        CHK(allreduce_dev(s, s->scal.p + SC_PAIR, 2));
        hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)(acc ? s->zvec.p : nullptr), k, 11);
        hipLaunchKernelGGL(k_axpy3_norm, dim3(nb), dim3(256), 0, s->stream, V.w, V.ucur, V.uprev, acc, s->scal.p, s->vlen, s->partial.p);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, s->partial.p, (int64_t)nb, s->scal.p + SC_NPEND);
    HIPCHK(hipGetLastError());
    s->npend = true; s->npend_k = k;
    double *t = V.uprev; V.uprev = V.ucur; V.ucur = V.w; V.w = t;
    return EDGPU_OK;
}
// the pending norm becomes beta (and the scales rotate) now: one 8-byte all-reduce
static int lz_flush(S *s)
{
    if (!s->npend) return EDGPU_OK;
    CHK(allreduce_dev(s, s->scal.p + SC_NPEND, 1));
    hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, s->npend_k, 12);
    HIPCHK(hipGetLastError());
    s->npend = false;
    return EDGPU_OK;
}
struct OneArScope {      // the deferred form is on between construction and destruction of this object (every return path of the loops)
    S *s; bool was;
    explicit OneArScope(S *s_) : s(s_), was(s_->one_ar_on) { s->one_ar_on = s->multi && s->one_ar; s->npend = false; }
    ~OneArScope() { s->one_ar_on = was; s->npend = false; }
};
// the pending beta step of the self-reducing iteration, taken by a reduction kernel (before anything but its own product kernel
// looks at the scales or the coefficients)
static int lz_sr_flush(S *s)
{
    if (!s->sr_pending) return EDGPU_OK;
    hipLaunchKernelGGL(k_reduce_final_step, dim3(1), dim3(1024), 0, s->stream, (const double *)s->partial_b.p, (int64_t)s->sr_nb, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, 1);
    HIPCHK(hipGetLastError());
    s->sr_pending = false;
    return EDGPU_OK;
}
static bool lz_sr_ok(const S *s)
{
    return s->sr_on && s->sr_allow && !s->multi && s->kron2 && !s->kron_split && s->H.desc.storage != EDGPU_STORE_CSR && !s->pad_on &&
           s->nloc <= ((int64_t)1 << 22) && s->H.Qdw * s->H.DimPh <= 8192;
}
struct SrScope {      // the plain Lanczos loops allow the self-reducing iteration between construction and destruction of this object
    S *s; explicit SrScope(S *s_) : s(s_) { s->sr_allow = true; s->sr_pending = false; }
    ~SrScope() { s->sr_allow = false; s->sr_pending = false; s->sr_now = false; s->sr_host_partials = false; }
};
static int lz_iter(S *s, LzVecs &V, int k, double *acc)
{
    if (s->multi && s->one_ar_on) return lz_iter_deferred(s, V, k, acc);
    if (!acc && lz_sr_ok(s)) {      // two launches: product (takes the pending beta step, leaves alpha partials), axpy (takes alpha, leaves beta^2 partials)
        const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(s->vlen, 256 * 4), 256));
        if (s->partial_b.n < 512) CHK(s->partial_b.alloc(512));
        s->sr_now = true;
        const int rc = hxv(s, V.ucur, V.w, true, V.uprev, s->vld);
        s->sr_now = false;
        if (rc) return rc;
        hipLaunchKernelGGL(k_axpy_norm_sr, dim3(nb), dim3(256), 0, s->stream, V.w, (const double *)V.ucur, (const double *)s->scal.p, s->scal.p, (const double *)s->partial.p, s->sr_na, s->alphas.p, s->vlen, s->partial_b.p, s->sr_host_partials ? s->partial_bh.p : (double *)nullptr);
        HIPCHK(hipGetLastError());
        s->sr_pending = true; s->sr_nb = nb;
        double *t = V.uprev; V.uprev = V.ucur; V.ucur = V.w; V.w = t;
        return EDGPU_OK;
    }
    CHK(lz_sr_flush(s));
    if (!s->multi) {   // alpha step rides on the product's final reduction (device counter == k by construction)
        s->fuse_mode = 0; s->fuse_zvec = acc ? s->zvec.p : nullptr;
        int rc = hxv(s, V.ucur, V.w, true, V.uprev, s->vld);
        s->fuse_mode = -1; s->fuse_zvec = nullptr;
        if (rc) return rc;
        return lz_iter_tail(s, V, k, acc, true);
    }
    CHK(hxv(s, V.ucur, V.w, true, V.uprev, s->vld));
    return lz_iter_tail(s, V, k, acc);
}

// Small sectors are launch bound (an iteration is 4 short kernels, ~4.5 us each on the device side): what helps
// there is fewer launches (the fused reduce+step kernels above) and fewer host round trips (chunked stopping-rule
// checks in lanczos_gs).  Replaying three captured iterations as a HIP graph was measured and gave nothing over
// back-to-back asynchronous launches (200 iterations at N=4900: 3.64 ms vs 3.68 ms) -- not kept.
static bool lz_small(S *s)
{
    return !s->multi && s->nloc <= (int64_t)XKNOB("EDGPU_SMALL_MAXN", 1 << 22);
}
static int lz_iters(S *s, LzVecs &V, int k, int count, double *acc)
{
    for (const int kend = k + count; k < kend; k++) CHK(lz_iter(s, V, k, acc));
    return EDGPU_OK;
}

// ---- fused iteration (k_uppass2_ax / k_dwpass2 with uprev2; product.hpp: fused_local, fused_dw) --------------------------
// State after the local half of iteration k: u = u_k, uprev = u_{k-1}, t = local part of H u_k (unscaled), alphas[0..k-1] and
// betas[1..k] on the device.  One iteration = dw half of k (r_k, alpha_k) + local half of k+1 (u_{k+1}, beta_k): two passes
// over the vectors instead of three.
struct LzF { double *u, *uprev, *t; };
// dense vector (the caller's layout) <-> pitched vector of the fused loops; the padding columns of `dst` end up zero
static int lzf_copy_in(S *s, double *dst, const double *src)
{
    if (!s->pad_on) { HIPCHK(hipMemcpyAsync(dst, src, (size_t)s->nloc * 8, hipMemcpyDeviceToDevice, s->stream)); return EDGPU_OK; }
    HIPCHK(hipMemsetAsync(dst, 0, (size_t)s->vlen * 8, s->stream));
    HIPCHK(hipMemcpy2DAsync(dst, (size_t)s->vld * 8, src, (size_t)s->H.DimUp * 8, (size_t)s->H.DimUp * 8, (size_t)(s->nloc / s->H.DimUp), hipMemcpyDeviceToDevice, s->stream));
    return EDGPU_OK;
}
static int lzf_copy_out(S *s, double *dst, const double *src)
{
    if (!s->pad_on) { HIPCHK(hipMemcpyAsync(dst, src, (size_t)s->nloc * 8, hipMemcpyDeviceToDevice, s->stream)); return EDGPU_OK; }
    HIPCHK(hipMemcpy2DAsync(dst, (size_t)s->H.DimUp * 8, src, (size_t)s->vld * 8, (size_t)s->H.DimUp * 8, (size_t)(s->nloc / s->H.DimUp), hipMemcpyDeviceToDevice, s->stream));
    return EDGPU_OK;
}
static int lzf_zero_pads(S *s, double *x)      // a buffer that joins the elementwise passes (stored Krylov vectors) must carry zeros there
{
    if (!s->pad_on) return EDGPU_OK;
    hipLaunchKernelGGL(k_zero_pads, dim3(256), dim3(256), 0, s->stream, x, s->nloc / s->H.DimUp, (int32_t)s->H.DimUp, (int32_t)s->vld);
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}
static int lzf_begin(S *s, LzF &F, double *acc)      // F.u holds the (unnormalised) start vector, F.uprev any finite buffer
{
    HIPCHK(hipMemsetAsync(s->scal.p, 0, SC_COUNT * sizeof(double), s->stream));       // alpha_{-1} = 0, iteration counter 0
    CHK(lzf_zero_pads(s, F.t));      // odd DimUp: the dw half runs over DimUp + 1 columns and reads the padding column of t
    return fused_local(s, F.u, F.uprev, F.t, acc);                                     // u_0 unchanged, s_0 = 1/|u_0|
}
static int lzf_iter(S *s, LzF &F, double *tnew, double *acc)
{
    CHK(fused_dw(s, F.u, F.t, F.uprev));              // t <- r_k, alpha_k
    double *r = F.t;
    CHK(fused_local(s, r, F.u, tnew, acc));           // r <- u_{k+1}, beta_k ; tnew <- local part of H u_{k+1}
    F.uprev = F.u; F.u = r; F.t = tnew;
    return EDGPU_OK;
}

static int fetch_coeffs(S *s, int k, std::vector<double> &al, std::vector<double> &be)
{
    CHK(lz_sr_flush(s));
    al.resize(k + 1); be.resize(k + 2);
    HIPCHK(hipStreamSynchronize(s->stream));      // the coefficient arrays are host memory the kernels write (CoefBuf)
    std::memcpy(al.data(), s->alphas.h, (size_t)(k + 1) * 8); std::memcpy(be.data(), s->betas.h, (size_t)(k + 2) * 8);
    return EDGPU_OK;
}

// ---- whole Lanczos loop in one workgroup (k_lanczos_wg): sectors of up to ~8 000 states -----------------------------------
struct WgPlan { bool ok = false; bool a_is_up = true; int nC = 1, E = 1, emax = 8, wa = 6, wb = 6; };
static WgPlan wg_plan(const S *s)
{
    WgPlan P;
    const SectorHost &H = s->H;
    if (!s->lz_wg_on || s->multi || !s->two_factor || s->extra || H.DimPh != 1 || !s->upf || !s->dwf) return P;
    const int64_t N = H.DimUp * H.DimDw;
    if (N < 2 || N * 8 > 150 * 1024) return P;
    P.a_is_up = H.DimUp >= H.DimDw;
    const int64_t DimA = P.a_is_up ? H.DimUp : H.DimDw, DimB = P.a_is_up ? H.DimDw : H.DimUp;
    const int wA = P.a_is_up ? s->upf->width : s->dwf->width, wB = P.a_is_up ? s->dwf->width : s->upf->width;
    if (DimA > 1024 || wA > 12 || wB > 12 || wB < 1) return P;
    int nC = 1; while (nC * 64 < DimA) nC *= 2;
    P.nC = nC; P.E = (int)cdiv(DimB, 16 / nC);
    if (P.E > 12) return P;     // up to 12 elements per lane: w and v_{k-1} both stay in registers; with 16-20 the loop measured SLOWER than the per-iteration launches (33 vs 22 us per step at N = 14 400: v_{k-1} through L2, scalar registers spilled into lanes)
    P.emax = P.E <= 8 ? 8 : 12;
    if (P.emax == 12 && wA > 6) return P;      // 12 elements per lane AND 12 register-resident hop entries do not fit 128 registers
    P.wa = wA <= 6 ? 6 : 12; P.wb = wB <= 6 ? 6 : 12;
    P.ok = true;
    return P;
}
// `nsteps` iterations from iteration k0 on: v = v_k (any norm when k0 == 0), vprev = v_{k-1}; coefficients into s->alphas / s->betas;
// v_k stored at krylov + k*kstride when krylov != nullptr.  *done = iterations completed so far (after a stream synchronisation).
static int lanczos_wg_steps(S *s, const WgPlan &P, double *v, double *vprev, int k0, int nsteps, double threshold, double *krylov, int64_t kstride, int *done)
{
    const SectorHost &H = s->H; const edgpu_desc &d = H.desc;
    if (d.storage == EDGPU_STORE_DIRECT) CHK(direct_tables(s));
    if (!s->wg_done.p) CHK(s->wg_done.alloc(1));
    LzWgArgs A{};
    const DevFactor &fa = P.a_is_up ? *s->upf : *s->dwf, &fb = P.a_is_up ? *s->dwf : *s->upf;
    A.DimA = (int32_t)(P.a_is_up ? H.DimUp : H.DimDw); A.DimB = (int32_t)(P.a_is_up ? H.DimDw : H.DimUp);
    A.sA = P.a_is_up ? 1 : (int32_t)H.DimUp; A.sB = P.a_is_up ? (int32_t)H.DimUp : 1;
    A.nC = P.nC; A.E = P.E;
    // ELL images: the up factor is stored [k][iup] (transposed), the dw factor [idw][k]
    A.cA = fa.ellc.p; A.vA = fa.ellv.p; A.wA = fa.width; A.ksA = P.a_is_up ? (int32_t)H.DimUp : 1; A.asA = P.a_is_up ? 1 : fa.width;
    A.cB = fb.ellc.p; A.vB = fb.ellv.p; A.wB = fb.width; A.ksB = P.a_is_up ? 1 : (int32_t)H.DimUp; A.bsB = P.a_is_up ? fb.width : 1;
    A.EA = fa.E.p; A.EB = fb.E.p; A.mapA = fa.map.p; A.mapB = fb.map.p; A.cross = s->cross.p;
    const int nI = 1 << d.Norb;
    A.multA = P.a_is_up ? nI : 1; A.multB = P.a_is_up ? 1 : nI; A.impmask = nI - 1; A.econst = H.econst;
    A.v = v; A.vprev = vprev; A.alphas = s->alphas.p; A.betas = s->betas.p; A.k0 = k0; A.nsteps = nsteps; A.threshold = threshold;
    A.krylov = krylov; A.kstride = kstride; A.done = s->wg_done.p;
    // the record table of the wave-uniform dimension, derived on the device from the ELL images (once per sector; storage=DIRECT:
    // after the on-the-fly tables exist)
    if (!s->wg_brec.p) {
        const int recd = P.wb == 6 ? WgRec<6>::kDwords : WgRec<12>::kDwords;
        CHK(s->wg_brec.alloc((size_t)A.DimB * recd));
        if (P.wb == 6) hipLaunchKernelGGL(k_wg_pack_b<6>, dim3((unsigned)cdiv(A.DimB, 64)), dim3(64), 0, s->stream, A, s->wg_brec.p);
        else hipLaunchKernelGGL(k_wg_pack_b<12>, dim3((unsigned)cdiv(A.DimB, 64)), dim3(64), 0, s->stream, A, s->wg_brec.p);
        HIPCHK(hipGetLastError());
    }
    A.brec = s->wg_brec.p;
    const size_t lds = (size_t)(H.DimUp * H.DimDw) * 8;
#define WG_LAUNCH(E_, W_, B_, R_)                                                                                               \
    do {                                                                                                                            \
        static OncePerDevice once_;                                                                                                 \
        if (once_.first(s->device)) (void)hipFuncSetAttribute((const void *)k_lanczos_wg<E_, W_, B_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024); \
        hipLaunchKernelGGL((k_lanczos_wg<E_, W_, B_, R_>), dim3(1), dim3(1024), lds, s->stream, A);                                 \
    } while (0)
    if (P.emax == 12) { if (P.wb == 6) WG_LAUNCH(12, 6, 6, true); else WG_LAUNCH(12, 6, 12, true); }
    else if (P.wa == 6) { if (P.wb == 6) WG_LAUNCH(8, 6, 6, true); else WG_LAUNCH(8, 6, 12, true); }
    else { if (P.wb == 6) WG_LAUNCH(8, 12, 6, true); else WG_LAUNCH(8, 12, 12, true); }
#undef WG_LAUNCH
    HIPCHK(hipGetLastError());
    if (done) { HIPCHK(hipMemcpyAsync(done, s->wg_done.p, 4, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream)); }
    return EDGPU_OK;
}
// sp_lanc_eigh on such a sector: chunks of iterations per launch with the Krylov vectors kept, SciFortran's stopping rule applied to
// each iteration in order on the host (same nlanc and E0 as the per-iteration loop, a few surplus iterations), the eigenvector as
// one weighted sum over the stored vectors
static int lanczos_gs_wg(S *s, const WgPlan &P, int nitermax, double threshold, int ncheck, double *egs, double *vect_dev, int *nlanc_out)
{
    const SectorHost &H = s->H;
    const int64_t N = s->nloc;
    if ((int64_t)nitermax > H.Dim) nitermax = (int)H.Dim;
    CHK(lz_prepare(s, nitermax));
    if (s->lz[3].n != (size_t)N) CHK(s->lz[3].alloc((size_t)N));
    double *v = s->lz[0].p, *vprev = s->lz[1].p, *acc = s->lz[3].p;
    HIPCHK(hipMemcpyAsync(v, vect_dev, (size_t)N * 8, hipMemcpyDeviceToDevice, s->stream));
    CHK(reduce_dot(s, v, v));
    double nrm2;
    HIPCHK(hipMemcpyAsync(&nrm2, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
    if (nrm2 == 0.0) {      // zero input => library-generated start vector (ED_DIAG.f90:147 passes eig_basis=0)
        hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, s->stream, v, H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
        HIPCHK(hipGetLastError());
    }
    const int64_t kstride = (N + 1) & ~(int64_t)1;
    const size_t want = (size_t)(nitermax + 1) * (size_t)kstride;
    if (s->wg_slab.n < want) CHK(s->wg_slab.alloc(want));
    std::vector<double> al, be;
    int nlanc = 0, done = 0; double e0 = 0.0; bool stop = false; StopRule rule;
    const int chunk = 16;
    for (int k = 0; k < nitermax && !stop;) {
        const int n = std::min(chunk, nitermax - k);
        CHK(lanczos_wg_steps(s, P, v, vprev, k, n, threshold, s->wg_slab.p, kstride, &done));
        if (done <= k) return fail(EDGPU_EHIP, "edgpu: internal: the workgroup Lanczos kernel made no progress");
        CHK(fetch_coeffs(s, done - 1, al, be));      // a kernel that stopped early (beta below the threshold) is caught by the first test below
        for (; k < done; k++) {
            nlanc = k + 1;
            if (std::fabs(be[k + 1]) < threshold) { stop = true; break; }
            const int r = rule.check(al.data(), be.data(), nlanc, ncheck, threshold, &e0);
            if (r < 0) return fail(EDGPU_ENOTCONVERGED, "edgpu: tridiagonal QL did not converge");
            if (r > 0) { stop = true; break; }
        }
    }
    std::vector<double> z;
    if (!tridiag_lowest(al.data(), be.data(), nlanc, &e0, &z)) return fail(EDGPU_ENOTCONVERGED, "edgpu: tridiagonal QL did not converge");
    *egs = e0; *nlanc_out = nlanc;
    HIPCHK(hipMemsetAsync(acc, 0, (size_t)N * 8, s->stream));
    if (s->eig_h.n < (size_t)(kMaxBasis + 2)) CHK(s->eig_h.alloc((size_t)kMaxBasis + 2));
    const int nbk = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(N, 256 * 4), 4096));
    std::vector<double> coef(nlanc);
    for (int k = 0; k < nlanc; k++) coef[k] = -z[k];      // k_multiaxpy subtracts; the stored vectors are normalised
    for (int k0 = 0; k0 < nlanc; k0 += kMaxBasis) {
        BasisArgs B{}; B.nv = std::min(kMaxBasis, nlanc - k0);
        for (int i = 0; i < B.nv; i++) B.V[i] = s->wg_slab.p + (int64_t)(k0 + i) * kstride;
        HIPCHK(hipMemcpyAsync(s->eig_h.p, coef.data() + k0, (size_t)B.nv * 8, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        launch_multiaxpy(s->stream, nbk, B, (const double *)s->eig_h.p, acc, N);
        HIPCHK(hipGetLastError());
    }
    CHK(reduce_dot(s, acc, acc, N));
    hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, 0, 2);
    hipLaunchKernelGGL(k_scale, dim3(nbk), dim3(256), 0, s->stream, acc, s->scal.p, (int)SC_SCUR, N);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(vect_dev, acc, (size_t)N * 8, hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return EDGPU_OK;
}

static int lanczos_gs_impl(S *s, int nitermax, double threshold, int ncheck, double *egs, double *vect_dev, int *nlanc_out);
static int lanczos_gs(S *s, int nitermax, double threshold, int ncheck, double *egs, double *vect_dev, int *nlanc_out)
{
    if (nitermax >= 1) { const WgPlan P = wg_plan(s); if (P.ok) return lanczos_gs_wg(s, P, nitermax, threshold, ncheck < 1 ? 10 : ncheck, egs, vect_dev, nlanc_out); }
    // the cached Krylov vectors must survive an out-of-memory retry of any allocation made while the search runs
    g_kpool.in_use = true;
    const int rc = lanczos_gs_impl(s, nitermax, threshold, ncheck, egs, vect_dev, nlanc_out);
    g_kpool.in_use = false;
    if (rc == EDGPU_OK && !XKNOB("EDGPU_LZ_CACHE", 1)) kpool_release();      // default: keep the buffers for the next search
    return rc;
}
static int lanczos_gs_impl(S *s, int nitermax, double threshold, int ncheck, double *egs, double *vect_dev, int *nlanc_out)
{
    if (nitermax < 1) return fail(EDGPU_EINVAL, "edgpu_lanczos_gs: nitermax < 1");
    if (ncheck < 1) ncheck = 10;
    if ((int64_t)nitermax > s->H.Dim) nitermax = (int)s->H.Dim;   // ED_DIAG.f90:100 Nitermax=min(dim,lanc_niter)
    CHK(lz_prepare(s, nitermax));
    OneArScope one_ar_scope(s);
    SrScope sr_scope(s);
    const bool fused = fused_ok(s);
    const int64_t vlen = s->vlen;      // length of a work vector (row pitch padded to the 128-B line where the product kernels take one)
    if (s->lz[3].n != (size_t)s->nloc) CHK(s->lz[3].alloc((size_t)s->nloc));
    if (s->lz[4].n != (size_t)vlen) CHK(s->lz[4].alloc((size_t)vlen));
    double *start = s->lz[3].p, *acc = s->lz[4].p;      // start keeps the caller's dense layout
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
    // Krylov vectors kept in HBM when they fit (288 GB per GPU: ~90 vectors of 1.3 GB at Ns=16): the eigenvector is
    // then one weighted sum over the stored vectors instead of SciFortran's second Lanczos pass -- same nlanc, same
    // E0, same vector up to rounding, half the products.  Buffers are taken on demand; if memory runs out the run
    // simply continues with three rotating vectors and falls back to the second pass.
    const int store_max = knob("EDGPU_LZ_STORE_MAX");               // tests: cap the number of stored vectors
    bool store = knob("EDGPU_LZ_STORE") != 0 && store_max >= 4;
    int64_t budget = 0;       // bytes this call may still add to the cache (one driver query: hipMemGetInfo is slow)
    if (store) {
        if (g_kpool.device != s->device) { g_kpool.in_use = false; kpool_release(); g_kpool.in_use = true; g_kpool.device = s->device; }
        size_t fr = 0, tot = 0; HIPCHK(hipMemGetInfo(&fr, &tot));
        budget = (int64_t)fr - (int64_t)((double)tot * 0.04) - ((int64_t)2 << 30);     // head-room left to everything else
        for (size_t k = 0; k < g_kpool.cap.size(); k++) if (g_kpool.cap[k] >= (size_t)vlen) budget += (int64_t)vlen * 8;   // already ours
        // a search rarely ends before ~50 iterations: when not even that many vectors fit (Ns=18: 19 GB each) do not
        // tie the memory up for nothing, go straight to the two-pass form
        if (budget < (int64_t)std::min(nitermax, 48) * vlen * 8) store = false;
    }
    // small sectors: every vector of the search in ONE cached allocation (grown geometrically), so that a run over many
    // small sectors of growing size does not pay a hundred allocations each time
    const int64_t vstride = (vlen + 1) & ~(int64_t)1;      // even stride: every vector of the slab starts on a 16-byte boundary (the basis kernels use 16-B loads)
    const int64_t want_all = (int64_t)std::min(nitermax + 2, store_max) * vstride;
    const bool slab_mode = store && want_all * 8 <= ((int64_t)512 << 20);
    if (slab_mode && g_kpool.slab_cap < (size_t)want_all) {
        if (g_kpool.slab) { (void)hipFree(g_kpool.slab); g_kpool.slab = nullptr; }
        const size_t cap = std::max<size_t>((size_t)want_all, 2 * g_kpool.slab_cap);
        if (hipMalloc((void **)&g_kpool.slab, cap * 8) == hipSuccess) g_kpool.slab_cap = cap; else { (void)hipGetLastError(); g_kpool.slab = nullptr; g_kpool.slab_cap = 0; }
    }
    // buffer of u_k: cached from an earlier search when large enough, else (re)allocated within the budget
    auto ring_get = [&](int k) -> double * {
        if (k >= store_max) return nullptr;
        if (slab_mode) return (g_kpool.slab && (size_t)(k + 1) * (size_t)vstride <= g_kpool.slab_cap) ? g_kpool.slab + (int64_t)k * vstride : nullptr;
        while ((int)g_kpool.buf.size() <= k) { g_kpool.buf.push_back(nullptr); g_kpool.cap.push_back(0); }
        if (g_kpool.cap[k] >= (size_t)vlen) return g_kpool.buf[k];
        const int64_t need = (int64_t)vlen * 8;
        if (g_kpool.buf[k]) { (void)hipFree(g_kpool.buf[k]); budget += (int64_t)g_kpool.cap[k] * 8; g_kpool.buf[k] = nullptr; g_kpool.cap[k] = 0; }
        if (budget < need) return nullptr;
        double *q = nullptr;
        if (hipMalloc((void **)&q, (size_t)need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        budget -= need; g_kpool.buf[k] = q; g_kpool.cap[k] = (size_t)vlen;
        return q;
    };
    if (store) {
        // take what a typical search needs before the first kernel is queued (allocation is cheap on an idle device)
        const int guess = std::min(std::min(nitermax + 2, store_max), XKNOB("EDGPU_LZ_STORE_GUESS", 100));
        if (!slab_mode) for (int k = 0; k < guess; k++) if (!ring_get(k)) break;
    }
    LzVecs V{s->lz[0].p, s->lz[1].p, s->lz[2].p};
    LzF F{s->lz[0].p, s->lz[2].p, s->lz[1].p};
    if (store) {
        double *b0 = ring_get(0), *b1 = ring_get(1), *b2 = fused ? ring_get(2) : b1;
        if (b0 && b1 && b2) { V.ucur = b0; V.w = b1; F.u = b0; F.t = b1; F.uprev = b0; } else store = false;
    }
    double s0 = 0.0;                                  // scale of the start vector: v_0 = s0 * u_0
    if (fused) {
        CHK(lzf_copy_in(s, F.u, start));
        if (!store) HIPCHK(hipMemsetAsync(F.uprev, 0, (size_t)vlen * 8, s->stream));
        if (store) CHK(lzf_zero_pads(s, F.t));
        CHK(lzf_begin(s, F, nullptr));
    } else {
        CHK(lzf_copy_in(s, V.ucur, start));
        CHK(lzf_zero_pads(s, V.w));
        CHK(lz_start(s, V));
    }
    HIPCHK(hipMemcpyAsync(&s0, s->scal.p + SC_SCUR, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
    std::vector<double> al, be;
    int nlanc = 0; double e0 = 0.0; StopRule rule;
    using clk = std::chrono::steady_clock; auto tsec = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count() * 1e3; };
    const clk::time_point t_begin = clk::now(); double ms_issue = 0.0, ms_wait = 0.0, ms_rule = 0.0;      // EDGPU_DEBUG bit 3: where the wall time of a search goes
    // The stopping rule of iteration i only involves alpha/beta up to i, so iterations can be issued ahead of
    // the host check: small (launch-bound) sectors run `chunk` iterations per host round trip and then apply
    // the rule to each of them in order -- same nlanc and E0, at most chunk-1 surplus iterations.
    const int chunk = lz_small(s) ? std::max(1, XKNOB("EDGPU_LZ_CHUNK", 6)) : 1;
    bool done = false;
    // iterations [k0, k1) queued on the stream (no host round trip)
    auto issue = [&](int k0, int k1) -> int {
        for (int j = k0; j < k1; j++) {
            if (fused) {                              // dw half of j + local half of j+1: alpha_j and beta_j
                double *tnew = store ? ring_get(j + 2) : nullptr;
                if (store && !tnew) store = false;    // out of memory: from here on the oldest of the three live vectors is recycled
                if (tnew) CHK(lzf_zero_pads(s, tnew));
                else { tnew = (j == 0 && F.uprev == F.u) ? s->lz[2].p : F.uprev; if (tnew != F.uprev) CHK(lzf_zero_pads(s, tnew)); }
                CHK(lzf_iter(s, F, tnew, nullptr));
                continue;
            }
            CHK(lz_iter(s, V, j, nullptr));           // afterwards: uprev = u_j, ucur = u_{j+1}, w = recycled u_{j-1}
            if (store) {
                double *nx = ring_get(j + 2);         // u_{j+2} gets a buffer of its own, u_{j-1} stays intact
                if (nx) { V.w = nx; CHK(lzf_zero_pads(s, nx)); } else store = false; // out of memory: keep rotating through the last three
            }
        }
        return EDGPU_OK;
    };
    // the rule applied to iterations [k, kend) in order, coefficients up to kend-1 in al / be; advances k
    auto judge = [&](int &k, int kend) -> int {
        for (; k < kend; k++) {
            nlanc = k + 1;
            if (s->one_ar_on) {       // beta_{k+1} is still a pending partial sum: the invariant-subspace test looks at beta_k, one product late
                if (k > 0 && std::fabs(be[k]) < threshold) { nlanc = k; done = true; break; }
            } else
            if (std::fabs(be[k + 1]) < threshold) { done = true; break; }       // invariant subspace reached
            const int r = rule.check(al.data(), be.data(), nlanc, ncheck, threshold, &e0);
            if (r < 0) return fail(EDGPU_ENOTCONVERGED, "edgpu: tridiagonal QL did not converge");
            if (r > 0) { done = true; break; }
        }
        return EDGPU_OK;
    };
    // Launch-bound sectors, one rank: the next chunk is queued BEFORE the host waits for the one it is about to judge, so the device
    // works through it while the host evaluates the rule (the rule of a chunk costs about what its kernels do: 0.76 ms of rule, 0.93 ms of
    // waiting and 0.53 ms of issuing per 77-iteration search at 52 272 states, strictly one after the other before).  Same nlanc and E0 -- the rule
    // sees the same coefficients in the same order; up to 2*chunk-1 surplus iterations instead of chunk-1.
    const bool ahead = chunk > 1 && !fused && !s->multi && XKNOB("EDGPU_LZ_AHEAD", 1) != 0;
    if (ahead) {
        for (auto &e : s->lz_ev) if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        int issued = std::min(nitermax, chunk), c = 0;
        { const clk::time_point t0 = clk::now(); CHK(issue(0, issued)); CHK(lz_sr_flush(s)); HIPCHK(hipEventRecord(s->lz_ev[0], s->stream)); ms_issue += tsec(t0, clk::now()); }
        for (int k = 0; k < nitermax && !done; c++) {
            const int kend = std::min(nitermax, k + chunk);      // the chunk to judge now; everything up to `issued` is queued
            const clk::time_point t0 = clk::now();
            if (issued < nitermax) {
                const int k1 = std::min(nitermax, issued + chunk);
                CHK(issue(issued, k1)); CHK(lz_sr_flush(s)); HIPCHK(hipEventRecord(s->lz_ev[(c + 1) & 1], s->stream));
                issued = k1;
            }
            const clk::time_point t1 = clk::now();
            HIPCHK(hipEventSynchronize(s->lz_ev[c & 1]));      // the coefficient arrays are host memory the kernels write (CoefBuf)
            al.resize(kend); be.resize(kend + 1);
            std::memcpy(al.data(), s->alphas.h, (size_t)kend * 8); std::memcpy(be.data(), s->betas.h, (size_t)(kend + 1) * 8);
            const clk::time_point t2 = clk::now();
            CHK(judge(k, kend));
            ms_issue += tsec(t0, t1); ms_wait += tsec(t1, t2); ms_rule += tsec(t2, clk::now());
        }
    } else
    for (int k = 0; k < nitermax && !done;) {
        const int kend = std::min(nitermax, k + chunk);
        const clk::time_point t0 = clk::now();
        CHK(issue(k, kend));
        const clk::time_point t1 = clk::now();
        CHK(fetch_coeffs(s, kend - 1, al, be));
        const clk::time_point t2 = clk::now();
        CHK(judge(k, kend));
        ms_issue += tsec(t0, t1); ms_wait += tsec(t1, t2); ms_rule += tsec(t2, clk::now());
    }
    const clk::time_point t_loop = clk::now();
    std::vector<double> z;
    if (!tridiag_lowest(al.data(), be.data(), nlanc, &e0, &z)) return fail(EDGPU_ENOTCONVERGED, "edgpu: tridiagonal QL did not converge");
    if (s->dbg & 8) fprintf(stderr, "edgpu_lanczos_gs: %d iterations; loop %.3f ms (issuing %.3f, waiting for the device %.3f, stopping rule %.3f), eigenvector of T %.3f ms\n",
                            nlanc, tsec(t_begin, t_loop), ms_issue, ms_wait, ms_rule, tsec(t_loop, clk::now()));
    *egs = e0; *nlanc_out = nlanc;
    if (s->multi) {      // every rank must take the same branch below (the second pass contains collectives)
        DevBuf<double> f; std::vector<double> one(1, store ? 1.0 : 0.0);
        CHK(f.upload(one)); CHK(allreduce_dev(s, f.p, 1));
        HIPCHK(hipMemcpyAsync(one.data(), f.p, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
        if (one[0] < (double)s->comm.size - 0.5) store = false;
    }
    HIPCHK(hipMemsetAsync(acc, 0, (size_t)vlen * 8, s->stream));
    const int nbk = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(vlen, 256 * 4), 4096));
    if (store) {
        // vect = sum_k z_k v_k = sum_k (z_k s_k) u_k over the stored vectors, kMaxBasis of them per pass over acc
        std::vector<double> coef(nlanc);
        for (int k = 0; k < nlanc; k++) coef[k] = -z[k] * (k == 0 ? s0 : (be[k] > 0.0 ? 1.0 / be[k] : 0.0));   // k_multiaxpy subtracts
        if (s->eig_h.n < (size_t)(kMaxBasis + 2)) CHK(s->eig_h.alloc((size_t)kMaxBasis + 2));
        for (int k0 = 0; k0 < nlanc; k0 += kMaxBasis) {
            BasisArgs B{}; B.nv = std::min(kMaxBasis, nlanc - k0);
            for (int i = 0; i < B.nv; i++) B.V[i] = ring_get(k0 + i);
            HIPCHK(hipMemcpyAsync(s->eig_h.p, coef.data() + k0, (size_t)B.nv * 8, hipMemcpyHostToDevice, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));      // coef lives on the host stack of this call
            launch_multiaxpy(s->stream, (int)nbk, B, (const double *)s->eig_h.p, acc, vlen);
            HIPCHK(hipGetLastError());
        }
    } else {
        // second pass: regenerate the Krylov vectors and accumulate vect = sum_k Z(k,1) v_k
        if (fused) {
            std::vector<double> zs(nlanc);            // the local half of iteration k adds zs[k] * u_k ; v_k = s_k u_k
            for (int k = 0; k < nlanc; k++) zs[k] = z[k] * (k == 0 ? s0 : (be[k] > 0.0 ? 1.0 / be[k] : 0.0));
            HIPCHK(hipMemcpyAsync(s->zvec.p, zs.data(), (size_t)nlanc * 8, hipMemcpyHostToDevice, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
            F = LzF{s->lz[0].p, s->lz[2].p, s->lz[1].p};
            CHK(lzf_copy_in(s, F.u, start));
            HIPCHK(hipMemsetAsync(F.uprev, 0, (size_t)vlen * 8, s->stream));
            CHK(lzf_begin(s, F, acc));
            for (int k = 0; k + 1 < nlanc; k++) CHK(lzf_iter(s, F, F.uprev, acc));
        } else {
        V = LzVecs{s->lz[0].p, s->lz[1].p, s->lz[2].p};
        HIPCHK(hipMemcpyAsync(s->zvec.p, z.data(), (size_t)nlanc * 8, hipMemcpyHostToDevice, s->stream));
        CHK(lzf_copy_in(s, V.ucur, start));
        HIPCHK(hipMemsetAsync(V.w, 0, (size_t)vlen * 8, s->stream));      // padding columns of the third rotating vector
        CHK(lz_start(s, V));
        CHK(lz_iters(s, V, 0, nlanc, acc));
        }
    }
    CHK(reduce_dot(s, acc, acc, vlen));
    hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, 0, 2);
    hipLaunchKernelGGL(k_scale, dim3(nbk), dim3(256), 0, s->stream, acc, s->scal.p, (int)SC_SCUR, vlen);
    HIPCHK(hipGetLastError());
    CHK(lzf_copy_out(s, vect_dev, acc));
    HIPCHK(hipStreamSynchronize(s->stream));
    return EDGPU_OK;
}

static int lanczos_tridiag(S *s, const double *vin_dev, int nlanc, double threshold, double *alanc, double *blanc, int *nused)
{
    if (nlanc < 1) return fail(EDGPU_EINVAL, "edgpu_lanczos_tridiag: nlanc < 1");
    CHK(lz_prepare(s, nlanc));
    {
        const WgPlan P = wg_plan(s);
        if (P.ok) {      // the whole leg in one launch of one workgroup
            int done = 0;
            HIPCHK(hipMemcpyAsync(s->lz[0].p, vin_dev, (size_t)s->nloc * 8, hipMemcpyDeviceToDevice, s->stream));
            CHK(lanczos_wg_steps(s, P, s->lz[0].p, s->lz[1].p, 0, nlanc, threshold, nullptr, 0, &done));
            std::vector<double> al, be;
            CHK(fetch_coeffs(s, std::max(done, 1) - 1, al, be));
            for (int i = 0; i < nlanc; i++) { alanc[i] = 0.0; blanc[i] = 0.0; }
            int used = 0;
            for (int i = 0; i < done; i++) {
                alanc[i] = al[i]; used = i + 1;
                if (std::fabs(be[i + 1]) < threshold) break;
                if (i + 1 < nlanc) blanc[i + 1] = be[i + 1];
            }
            *nused = used;
            return EDGPU_OK;
        }
    }
    OneArScope one_ar_scope(s);
    SrScope sr_scope(s);
    const bool fused = fused_ok(s);
    LzVecs V{s->lz[0].p, s->lz[1].p, s->lz[2].p};
    LzF F{s->lz[0].p, s->lz[2].p, s->lz[1].p};
    if (fused) {
        CHK(lzf_copy_in(s, F.u, vin_dev));
        HIPCHK(hipMemsetAsync(F.uprev, 0, (size_t)s->vlen * 8, s->stream));
        CHK(lzf_begin(s, F, nullptr));
    } else {
        CHK(lzf_copy_in(s, V.ucur, vin_dev));
        if (s->pad_on) HIPCHK(hipMemsetAsync(V.w, 0, (size_t)s->vlen * 8, s->stream));
        CHK(lz_start(s, V));
    }
    for (int i = 0; i < nlanc; i++) { alanc[i] = 0.0; blanc[i] = 0.0; }
    std::vector<double> al, be;
    int used = 0;
    const int chunk = 8;  // look at beta on the host every `chunk` iterations
    for (int k = 0; k < nlanc;) {
        int kend = std::min(nlanc, k + chunk);
        if (fused) { for (int j = k; j < kend; j++) CHK(lzf_iter(s, F, F.uprev, nullptr)); }
        else CHK(lz_iters(s, V, k, kend - k, nullptr));
        k = kend;
        CHK(lz_flush(s));      // deferred-norm form: the last beta of the chunk
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

// in-place rotation of a basis of up to 96 vectors on the matrix cores (k_basis_rotate_mfma)
static inline int launch_basis_rotate_mfma(hipStream_t st, int device, double *const *V, int nv, int nout, const double *Ydev, int64_t n)
{
    constexpr int KU = 8;
    RotMfmaArgs R{}; R.nv = nv; R.nout = nout; R.Y = Ydev; R.nks = (int)(cdiv(cdiv(nv, 4), KU) * KU);
    for (int i = 0; i < kMaxBasisBig + 8; i++) R.V[i] = V[i < nv ? i : 0];
    const int ot = (int)cdiv(nout, 16);
    const size_t lds = (size_t)ot * R.nks * 64 * 8 + (size_t)4 * R.nks * 8;
    const int64_t ntiles = cdiv(n, 16);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(ntiles, 4 * 8), 1024));
#define ROT_(O_)                                                                                                                \
    case O_: {                                                                                                                      \
        static OncePerDevice once_;                                                                                                 \
        if (once_.first(device)) (void)hipFuncSetAttribute((const void *)k_basis_rotate_mfma<O_, KU>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
        hipLaunchKernelGGL((k_basis_rotate_mfma<O_, KU>), dim3(grid), dim3(256), lds, st, R, n);                                    \
    } break
    switch (ot) { ROT_(1); ROT_(2); ROT_(3); ROT_(4); ROT_(5); ROT_(6); default: return EDGPU_EINVAL; }
#undef ROT_
    return EDGPU_OK;
}
// Orthogonalise w against the first nv vectors of a basis by classical Gram-Schmidt in at most two passes.
// A single pass against a basis that is orthonormal to delta leaves w orthogonal only to about delta*|w|/|w'|, so the
// error is multiplied by |w|/|w'| at every step: harmless when that ratio is ~1, exponentially unstable when it is
// ~10 (H*v carries a large component along v itself).  Hence:
//   pass 1 runs over basis[first_from .. nv) -- the caller names the vectors that carry the bulk of w (for a Lanczos
//          vector the last two: the three-term recurrence), or all of them (first_from = 0);
//   pass 2 runs over the whole basis, always after a partial pass 1, else only if pass 1 removed more than half of
//          |w|^2 (the DGKS test).  After pass 2 the ratio is ~1 and the result is orthogonal to rounding.
// Every multi-dot also returns |w|^2 (w rides along as one more "basis vector") and every update pass |w'|^2, so
// no separate norm pass is needed.  hh (host, nv entries) receives the summed coefficients, wnorm the final |w|.
static int orthogonalise_cgs(S *s, double *const *basis, int nv, int first_from, double *w, int64_t n, int nb, std::vector<double> &hh, double &wnorm)
{
    std::vector<double> h2((size_t)kMaxBasis + 1);
    hh.assign((size_t)nv, 0.0);
    double before = 0.0, after = 0.0;
    for (int pass = 0; pass < 2; pass++) {
        const int from = pass == 0 ? first_from : 0, cnt = nv - from;
        // bases beyond the capacity of one kernel argument block go in groups of kMaxBasis vectors: all dots are taken
        // against the SAME w (classical Gram-Schmidt), then the groups are subtracted one after the other
        const int ngroups = std::max(1, (int)cdiv(cnt, kMaxBasis));
        std::vector<double> hpass((size_t)std::max(cnt, 1), 0.0);
        for (int g = 0; g < ngroups; g++) {
            const int g0 = g * kMaxBasis, gc = std::min(kMaxBasis, cnt - g0);
            const bool lastg = g == ngroups - 1;
            BasisArgs B{}; B.nv = gc + (lastg ? 1 : 0);
            for (int i = 0; i < gc; i++) B.V[i] = basis[from + g0 + i];
            if (lastg) B.V[gc] = w;                           // w rides along once: |w|^2 before the update
            launch_multidot(s->stream, nb, B, (const double *)w, n, s->eig_partial.p);
            hipLaunchKernelGGL(k_multidot_final, dim3(1), dim3(256), 0, s->stream, (const double *)s->eig_partial.p, nb, B.nv, s->eig_h.p + g0);
            HIPCHK(hipGetLastError());
        }
        CHK(allreduce_dev(s, s->eig_h.p, cnt + 1));
        for (int g = 0; g < ngroups; g++) {
            const int g0 = g * kMaxBasis, gc = std::min(kMaxBasis, cnt - g0);
            BasisArgs Bu{}; Bu.nv = gc;
            for (int i = 0; i < gc; i++) Bu.V[i] = basis[from + g0 + i];
            if (g == ngroups - 1) launch_multiaxpy_norm(s->stream, nb, Bu, (const double *)(s->eig_h.p + g0), w, n, s->eig_partial.p);
            else launch_multiaxpy(s->stream, nb, Bu, (const double *)(s->eig_h.p + g0), w, n);
            HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, (const double *)s->eig_partial.p, (int64_t)nb, s->scal.p + SC_TMP);
        HIPCHK(hipGetLastError());
        CHK(allreduce_tmp(s));
        std::vector<double> hall((size_t)cnt + 1);
        HIPCHK(hipMemcpyAsync(hall.data(), s->eig_h.p, (size_t)(cnt + 1) * 8, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(&after, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        for (int i = 0; i < cnt; i++) hh[from + i] += hall[i];
        before = hall[cnt];
        if (pass == 0 && first_from == 0 && after >= 0.5 * before) break;      // DGKS: one full pass was enough
    }
    wnorm = std::sqrt(after > 0.0 ? after : 0.0);
    return EDGPU_OK;
}

// ---------------------------------------------------------------------------------------
// Lowest `nev` eigenpairs: thick-restart Lanczos with full re-orthogonalisation in a basis of
// `ncv` vectors held in HBM, explicit Rayleigh-Ritz on the host.  Stands in for the ARPACK call
// sp_eigh(MatVec,eval,evec,Nblock,Nitermax,tol) of ED_DIAG.f90:153-170 (which='SA', ncv=Nblock,
// maxiter=Nitermax restarts, residual test |r_i| <= tol*max(eps^(2/3),|theta_i|)).
// ---------------------------------------------------------------------------------------
static void jacobi_eig(std::vector<double> &A, int m, std::vector<double> &w, std::vector<double> &Y)
{   // symmetric m x m matrix A (row-major); eigenvalues ascending in w, vectors = columns of Y (Y[i + m*j]).  From 6 rows on:
    // Householder + QL (tridiag_host.hpp: 26 us against 73 at m = 20, 0.9 ms against 5.1 at m = 80); below that, and should QL
    // ever fail to converge: cyclic Jacobi
    if (m >= 6) { std::vector<double> A2(A); if (sym_eig_householder(A2, m, w, Y)) return; }
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

// ---------------------------------------------------------------------------------------
// The same solver with PARTIAL re-orthogonalisation (the default; EDGPU_EIGSH_PRO=0 selects the form below that
// re-orthogonalises every vector against the whole basis).  Between restarts the basis grows by the plain three-term
// recurrence on the device-resident Lanczos engine (lz_iter: product fused with -beta*v_prev and the alpha dot, one
// axpy+norm pass, scales carried as device scalars -- no pass over the basis, no separate normalisation).  The loss of
// orthogonality is tracked on the host by the omega recurrence (H. Simon, Math. Comp. 42 (1984) 115) written for the
// general projected matrix T of a thick-restarted basis (diagonal Ritz block + arrow + tridiagonal tail):
//     beta_j w(j+1,k) = sum_i T(i,k) w(i,j) - sum_{i<=j} T(i,j) w(i,k) + (eps |T| + eta_k + eta_j)
// When estimates pass sqrt(eps) the new vector is orthogonalised -- against the kept Ritz vectors they concern when only
// those are concerned (orthogonality is lost along converging Ritz directions, which a thick-restarted basis holds as
// explicit vectors), against the whole basis otherwise; and always the first vector after a restart, whose image couples
// to every kept Ritz vector.  eta is what makes this safe across restarts: the coefficients such a pass removes are not
// part of T, so the relation H y_k = theta_k y_k + sigma_k v_l of a Ritz vector formed afterwards holds only up to
// eta_k = |F s_k| (F: the removed coefficients, kept on the host); the new Lanczos vectors overlap with the discarded
// Ritz directions, so that defect feeds the recurrence at every step.  A first form without the term had estimates of 1e-12
// while the true loss reached 1e-3 after a handful of restarts, and T produced values far below the spectrum (a 400-state
// sector and Ns=16 with 20 vectors; tests/eigsh_model.py is the numpy model of this routine, tests/test_eigsh_model_cpu.py runs it).
// A semi-orthogonal basis reproduces the Ritz values of the orthogonal one to O(eps |H|); the returned Ritz vectors are
// orthonormalised among themselves at the end.  Basis vectors stay unnormalised (v_i = sc_i * B_i), the scales are folded
// into the rotation and projection coefficients.
// ---------------------------------------------------------------------------------------
static int eigsh_full(S *s, int nev, int ncv, int maxiter, double tol, double *evals, double *evecs_host, int *nconv_out, int *nrestart_out, int *nmatvec_out);
static int eigsh_pro(S *s, int nev, int ncv, int maxiter, double tol, double *evals, double *evecs_host, int *nconv_out, int *nrestart_out, int *nmatvec_out)
{
    const int64_t Dim = s->H.Dim, n = s->nloc, vlen = s->vlen;
    if (nev < 1) return fail(EDGPU_EINVAL, "edgpu_eigsh: nev < 1");
    if ((int64_t)nev >= Dim) return fail(EDGPU_EINVAL, "edgpu_eigsh: nev must be smaller than the sector dimension (the reference switches to LAPACK there, ED_DIAG.f90:105-107)");
    if (nev >= kMaxBasisBig) return fail(EDGPU_EUNSUPPORTED, "edgpu_eigsh: nev must be below the basis capacity (edgpu_eigsh_ncv_cap)");
    if (ncv > kMaxBasisBig) ncv = kMaxBasisBig;
    if ((int64_t)ncv > Dim) ncv = (int)Dim;
    if (ncv <= nev) ncv = (int)std::min<int64_t>(std::min<int64_t>(2 * nev + 1, kMaxBasisBig), Dim);
    if (ncv <= nev) return fail(EDGPU_EINVAL, "edgpu_eigsh: ncv must exceed nev");
    // Nblock is a workspace size, not part of the result: with HBM to spare a basis of 32 vectors needs fewer restarts and fewer
    // products than the reference's default of 20 (Ns=16, two pairs: 167 products / 0.81 s against 182 / 0.94 s); taken only
    // when the extra vectors fit comfortably (a quarter of the free memory), reported through edgpu_eigsh_ncv_used
    // (one rank only: the free memory differs from rank to rank, and every rank must run the same basis)
    if (ncv < 32 && Dim >= (1 << 20) && !s->multi && XKNOB("EDGPU_EIGSH_GROW", 1)) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && (size_t)(32 + 5) * (size_t)vlen * 8 < fr / 4) ncv = 32; else (void)hipGetLastError();
    }
    // Memory: the basis, one buffer ahead for the fused iteration, a zero/previous vector and a dense scratch vector.  When the
    // requested basis does not fit (Ns=18 on one GPU: 18.9 GB per vector, 14 fit) the search runs with the vectors that do --
    // more restarts, same pairs -- and says so through edgpu_eigsh_ncv_used; below nev + 2 vectors it refuses.  Several ranks
    // must run the same basis: there the request either fits everywhere or every rank returns EDGPU_ENOMEM.
    {
        const bool fused0 = fused_ok(s) && knob("EDGPU_EIGSH_FUSED") != 0;
        const int extra = 3 + (fused0 ? 1 : 0);      // B(m), the buffer ahead, lz[0], lz[3]
        const int memcap_mb = knob("EDGPU_EIGSH_MEM_MB");
        auto fit = [&]() -> int64_t {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return 1 << 20; }
            int64_t held = 0;
            for (auto &b : s->eigV) if (b && b->n >= (size_t)vlen) held++;
            if (s->lz[0].n == (size_t)vlen) held++;
            if (s->lz[3].n == (size_t)n) held++;
            int64_t avail = (int64_t)fr - ((int64_t)2 << 30) - (int64_t)((double)tot * 0.02);
            if (memcap_mb > 0) avail = std::min<int64_t>(avail, (int64_t)memcap_mb << 20);      // tests: pretend the device is this full
            return avail / (vlen * 8) + held - extra;
        };
        int64_t mfit = fit();
        if (mfit < ncv) {      // give back what other loops cached: Krylov vectors of earlier ground-state searches, the two rotating vectors this routine does not use
            kpool_release(); s->lz[1].release(); s->lz[2].release();
            for (auto &b : s->eigV) if (b && b->n && b->n < (size_t)vlen) b->release();
            mfit = fit();
        }
        int fits = mfit >= ncv ? 1 : 0;
        if (s->multi) {
            DevBuf<double> f; std::vector<double> one(1, (double)fits);
            CHK(f.upload(one)); CHK(allreduce_dev(s, f.p, 1));
            HIPCHK(hipMemcpyAsync(one.data(), f.p, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
            if (one[0] < (double)s->comm.size - 0.5)
                return fail(EDGPU_ENOMEM, "edgpu_eigsh: a basis of " + std::to_string(ncv) + " vectors of " + std::to_string(vlen * 8 >> 20) + " MiB does not fit on every rank (" + std::to_string(mfit) + " fit here)");
        } else if (!fits) {
            if (mfit < nev + 2)
                return fail(EDGPU_ENOMEM, "edgpu_eigsh: only " + std::to_string(std::max<int64_t>(mfit, 0)) + " basis vectors of " + std::to_string(vlen * 8 >> 20) + " MiB fit in the free device memory; " + std::to_string(nev + 2) + " are the minimum for " + std::to_string(nev) + " pairs (use more ranks)");
            ncv = (int)mfit;
        }
    }
    s->eig_ncv_used = ncv;
    if (maxiter < 1) maxiter = 1;
    const double eps = 2.220446049250313e-16, tol_eff = std::max(tol, 1e-12), eps23 = std::pow(eps, 2.0 / 3.0);
    const double tau = std::sqrt(eps) * (double)std::max(1, knob("EDGPU_EIGSH_OMEGA_PCT")) / 100.0;
    const double sel = (double)std::max(0, XKNOB("EDGPU_EIGSH_SEL_PPM", 100000)) * 1e-6;      // a pass over kept Ritz vectors takes those whose estimate exceeds sel*tau
    const int m = ncv, M = m + 2;
    CHK(lz_prepare(s, m + 4, 1));      // of the three rotating vectors of the plain loops only the first is used here
    if (s->lz[3].n != (size_t)n) CHK(s->lz[3].alloc((size_t)n));
    // the fused iteration (stored factors / on the fly, one rank) forms the local part of H u_{j+1} one buffer ahead
    const bool fused = fused_ok(s) && knob("EDGPU_EIGSH_FUSED") != 0;
    const int nbuf = m + 1 + (fused ? 1 : 0);
    while ((int)s->eigV.size() < nbuf) s->eigV.push_back(std::make_unique<DevBuf<double>>());
    for (int i = 0; i < nbuf; i++) {
        if (s->eigV[i]->n < (size_t)vlen && s->eigV[i]->alloc((size_t)vlen) != EDGPU_OK) {
            for (auto &b : s->eigV) if (b) b->release();      // leave nothing behind that would starve the next call
            return fail(EDGPU_ENOMEM, "edgpu_eigsh: out of device memory for basis vector " + std::to_string(i) + " of " + std::to_string(nbuf));
        }
        if (s->pad_on) HIPCHK(hipMemsetAsync(s->eigV[i]->p, 0, (size_t)vlen * 8, s->stream));      // padding columns must hold zeros
    }
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(vlen, 256 * 8), 1024));
    const int gs = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(vlen, 256 * 4), 4096));
    if (s->eig_h.n < (size_t)(m + 3)) CHK(s->eig_h.alloc((size_t)m + 3));
    if (s->eig_partial.n < (size_t)nb * (kMaxBasis + 2)) CHK(s->eig_partial.alloc((size_t)nb * (kMaxBasis + 2)));
    if (s->eig_y.n < (size_t)m * m) CHK(s->eig_y.alloc((size_t)m * m));
    auto B = [&](int i) { return s->eigV[i]->p; };
    const SectorHost &H = s->H;
    std::vector<double> sc((size_t)M, 1.0), T((size_t)M * M, 0.0), W((size_t)M * M, 0.0), F((size_t)M * M, 0.0), eta((size_t)M, 0.0), theta, Y, coef((size_t)M);
    auto Tm = [&](int i, int j) -> double & { return T[(size_t)i * M + j]; };
    auto Wm = [&](int i, int j) -> double & { return W[(size_t)i * M + j]; };
    auto Fm = [&](int i, int j) -> double & { return F[(size_t)i * M + j]; };
    auto rotate = [&](int nv, int nout, const std::vector<double> &Yc) -> int {      // B_a <- sum_k B_k Yc(k,a), a < nout (in place)
        HIPCHK(hipMemcpyAsync(s->eig_y.p, Yc.data(), (size_t)nv * nout * 8, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));      // Yc may be a temporary of the caller
        const int g2 = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(vlen, 256 * 2 * 16), 2048));
        const int rot2 = knob("EDGPU_ROT2");
        const bool rot_mfma = knob("EDGPU_ROT_MFMA") != 0;      // 0: the one- and two-element kernels that hold the basis of an element in registers
        if (rot_mfma) {      // tall-skinny GEMM on the matrix cores, any basis size (Ns=16: 80 -> 40 vectors 212 ms -> ~35 ms, 20 -> 2 slightly faster as well)
            std::vector<double *> vb((size_t)nv); for (int i = 0; i < nv; i++) vb[i] = B(i);
            CHK(launch_basis_rotate_mfma(s->stream, s->device, vb.data(), nv, nout, s->eig_y.p, vlen)); }
        else if (nv <= 20 && rot2) { RotArgs2<5> R{}; R.nv = nv; R.nout = nout; R.Y = s->eig_y.p; for (int i = 0; i < nv; i++) R.V[i] = B(i);
            hipLaunchKernelGGL(k_basis_rotate2<5>, dim3(g2), dim3(256), (size_t)nv * nout * 8, s->stream, R, vlen); }
        else if (nv <= 32 && rot2) { RotArgs2<8> R{}; R.nv = nv; R.nout = nout; R.Y = s->eig_y.p; for (int i = 0; i < nv; i++) R.V[i] = B(i);
            hipLaunchKernelGGL(k_basis_rotate2<8>, dim3(g2), dim3(256), (size_t)nv * nout * 8, s->stream, R, vlen); }
        else if (nv <= kMaxBasis) { RotArgs R{}; R.nv = nv; R.nout = nout; R.Y = s->eig_y.p; for (int i = 0; i < nv; i++) R.V[i] = B(i);
            hipLaunchKernelGGL(k_basis_rotate<kMaxBasis>, dim3(gs), dim3(256), (size_t)nv * nout * 8, s->stream, R, vlen); }
        else { RotArgsT<kMaxBasisBig> R{}; R.nv = nv; R.nout = nout; R.Y = s->eig_y.p; for (int i = 0; i < nv; i++) R.V[i] = B(i);
            static OncePerDevice once;
            if (once.first(s->device)) HIPCHK(hipFuncSetAttribute((const void *)k_basis_rotate<kMaxBasisBig>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 96 * 8));
            hipLaunchKernelGGL(k_basis_rotate<kMaxBasisBig>, dim3(gs), dim3(256), (size_t)nv * nout * 8, s->stream, R, vlen); }
        HIPCHK(hipGetLastError());
        return EDGPU_OK;
    };
    // w <- w - sum_{i in idx} <v_i,w> v_i with v_i = sc[i]*B_i (classical Gram-Schmidt, a second pass when the first one
    // removed more than half of |w|^2); |w|^2 afterwards in scal[TMP] and in nrm2; removed[t] = coefficient of v_idx[t]
    std::vector<double> removed;
    // One rank: what these passes hand to the host (dots, norms) and take from it (coefficients) goes through pinned host memory
    // the kernels write and read themselves -- a synchronisation where there was a copy call and a synchronisation (a copy call is
    // ~20 us of host time; a launch-bound sector pays ~30 such events per search: profiles/r04/eigsh_small_sector_where.txt).
    // Layout of eig_hp: [0, kHp) results of k_multidot_final, [kHp, 2 kHp) coefficients for the update kernels, [2 kHp] the last norm.
    constexpr int kHp = kMaxBasis + 8;
    const bool pin = !s->multi && XKNOB("EDGPU_EIGSH_PINNED", 1) != 0;
    if (pin && s->eig_hp.n < (size_t)(2 * kHp + 8)) CHK(s->eig_hp.alloc((size_t)2 * kHp + 8));
    double *const res_dev = pin ? s->eig_hp.p : s->eig_h.p;                 // device view of where the dots land
    double *const cf_dev = pin ? s->eig_hp.p + kHp : s->eig_h.p;            // ... and of where the coefficients are read from
    double *const nrm_dev = pin ? s->eig_hp.p + 2 * kHp : nullptr;
    auto fetch_norm = [&](double &out) -> int {      // after k_reduce_final2(..., scal + SC_TMP, nrm_dev)
        if (pin) { HIPCHK(hipStreamSynchronize(s->stream)); out = s->eig_hp.h[2 * kHp]; }
        else { HIPCHK(hipMemcpyAsync(&out, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream)); }
        return EDGPU_OK;
    };
    auto project_out = [&](const std::vector<int> &idx, const double *scl, double *w, double &nrm2) -> int {
        const int nv = (int)idx.size();
        removed.assign((size_t)nv, 0.0);
        double before = -1.0;
        const bool pinp = pin && nv + 1 <= kHp;      // larger index sets (bases beyond one kernel argument block) keep the copies
        for (int pass = 0; pass < 2; pass++) {
            const int ngroups = std::max(1, (int)cdiv(nv, kMaxBasis));
            for (int g = 0; g < ngroups; g++) {
                const int g0 = g * kMaxBasis, gc = std::min(kMaxBasis, nv - g0); const bool lastg = g == ngroups - 1;
                BasisArgs Ba{}; Ba.nv = gc + (lastg ? 1 : 0);
                for (int i = 0; i < gc; i++) Ba.V[i] = B(idx[g0 + i]);
                if (lastg) Ba.V[gc] = w;                      // |w|^2 before the update rides along
                launch_multidot(s->stream, nb, Ba, (const double *)w, vlen, s->eig_partial.p);
                hipLaunchKernelGGL(k_multidot_final, dim3(1), dim3(256), 0, s->stream, (const double *)s->eig_partial.p, nb, Ba.nv, (pinp ? res_dev : s->eig_h.p) + g0);
                HIPCHK(hipGetLastError());
            }
            if (pinp) { HIPCHK(hipStreamSynchronize(s->stream)); std::memcpy(coef.data(), s->eig_hp.h, (size_t)(nv + 1) * 8); }
            else {
                CHK(allreduce_dev(s, s->eig_h.p, nv + 1));
                HIPCHK(hipMemcpyAsync(coef.data(), s->eig_h.p, (size_t)(nv + 1) * 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
            }
            before = coef[nv];
            for (int i = 0; i < nv; i++) { removed[i] += coef[i] * scl[i]; coef[i] *= scl[i] * scl[i]; }
            if (pinp) std::memcpy(s->eig_hp.h + kHp, coef.data(), (size_t)nv * 8);
            else HIPCHK(hipMemcpyAsync(s->eig_h.p, coef.data(), (size_t)nv * 8, hipMemcpyHostToDevice, s->stream));
            const double *cfp = pinp ? cf_dev : s->eig_h.p;
            for (int g = 0; g < ngroups; g++) {
                const int g0 = g * kMaxBasis, gc = std::min(kMaxBasis, nv - g0);
                BasisArgs Bu{}; Bu.nv = gc; for (int i = 0; i < gc; i++) Bu.V[i] = B(idx[g0 + i]);
                if (g == ngroups - 1) launch_multiaxpy_norm(s->stream, nb, Bu, cfp + g0, w, vlen, s->eig_partial.p);
                else launch_multiaxpy(s->stream, nb, Bu, cfp + g0, w, vlen);
                HIPCHK(hipGetLastError());
            }
            hipLaunchKernelGGL(k_reduce_final2, dim3(1), dim3(1024), 0, s->stream, (const double *)s->eig_partial.p, (int64_t)nb, s->scal.p + SC_TMP, nrm_dev);
            HIPCHK(hipGetLastError());
            CHK(allreduce_tmp(s));
            CHK(fetch_norm(nrm2));
            if (nrm2 >= 0.5 * before) break;
        }
        return EDGPU_OK;
    };
    auto scalar_mode = [&](int mode) -> int {
        hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, -1, mode);
        HIPCHK(hipGetLastError());
        return EDGPU_OK;
    };
    auto all_upto = [](int j) { std::vector<int> v((size_t)j + 1); for (int i = 0; i <= j; i++) v[i] = i; return v; };
    double *dense = s->lz[3].p;
    hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, s->stream, dense, H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
    HIPCHK(hipGetLastError());
    CHK(lzf_copy_in(s, B(0), dense));
    LzVecs V{s->lz[0].p, B(0), B(1)};
    LzF Fz{B(0), s->lz[0].p, B(1)};                       // fused: u = u_j, uprev = u_{j-1}, t = local part of H u_j (unscaled), one buffer ahead
    if (fused) { HIPCHK(hipMemsetAsync(Fz.uprev, 0, (size_t)vlen * 8, s->stream)); CHK(lzf_begin(s, Fz, nullptr)); }
    else CHK(lz_start(s, V));                             // zeroes the dummy previous vector, scale of u_0 -> scal[SCUR]
    HIPCHK(hipMemcpyAsync(&sc[0], s->scal.p + SC_SCUR, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
    // after u_{j+1} = B(j+1) was changed in place: its norm / scale / beta_j follow; the fused form also redoes the local half
    // known_b2 >= 0: the caller already holds |u_{j+1}|^2 = scal[TMP] (project_out returned it): nothing to fetch on the plain path
    auto renorm = [&](int j, double &beta, double known_b2 = -1.0) -> int {
        if (fused) CHK(fused_local(s, B(j + 1), B(j), B(j + 2), nullptr, true));
        else CHK(scalar_mode(6));
        double b2 = known_b2;
        if (fused || !(known_b2 >= 0.0)) { HIPCHK(hipMemcpyAsync(&b2, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream)); }
        beta = std::sqrt(std::max(b2, 0.0));
        return EDGPU_OK;
    };
    Wm(0, 0) = 1.0;
    // launch-bound sectors: the two-launch (self-reducing) iteration, beta summed on the host from a pinned copy of the partial sums
    SrScope sr_scope(s);
    if (!fused && lz_sr_ok(s)) { if (s->partial_bh.n < 512) CHK(s->partial_bh.alloc(512)); s->sr_host_partials = true; } else s->sr_allow = false;
    int l = 0, j = 0, kloc = 0, nconv = 0, it = 0, nmv = 0, meff = m, n_full = 0, n_sel = 0; int64_t nreads = 0;
    using clk = std::chrono::steady_clock; auto tsec = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count() * 1e3; };
    const clk::time_point t_begin = clk::now();      // EDGPU_DEBUG bit 3: where the wall time of a search goes
    double ms_issue = 0.0, ms_wait = 0.0, ms_step_host = 0.0, ms_eig = 0.0, ms_restart_host = 0.0, ms_restart_dev = 0.0;
    bool exact = false; double normT = 0.0, beta_last = 0.0;
    uint64_t seed = 7654321;
    for (it = 0; it < maxiter; it++) {
        meff = m; exact = false;
        for (; j < m; j++) {
            const clk::time_point t0 = clk::now();
            if (fused) { Fz.u = B(j); Fz.t = B(j + 1); CHK(lzf_iter(s, Fz, B(j + 2), nullptr)); }      // afterwards Fz.u = B(j+1), Fz.uprev = B(j), Fz.t = B(j+2)
            else { V.ucur = B(j); V.w = B(j + 1); CHK(lz_iter(s, V, kloc, nullptr)); }              // afterwards V.uprev = B(j), V.ucur = B(j+1)
            nmv++;
            const clk::time_point t1 = clk::now();
            HIPCHK(hipStreamSynchronize(s->stream));
            const clk::time_point t2 = clk::now();
            ms_issue += tsec(t0, t1); ms_wait += tsec(t1, t2);
            struct StepTimer { const clk::time_point a; double &acc; ~StepTimer() { acc += std::chrono::duration<double>(clk::now() - a).count() * 1e3; } } step_timer{t2, ms_step_host};
            double alpha = s->alphas.h[kloc], beta;          // written by the step kernels into pinned host memory: no copy call
            if (s->sr_pending) {
                // self-reducing iteration (two launches): the beta step is still pending on the device (the next product takes it);
                // the host sums the partial sums of |w|^2 itself.  Whatever else touches the scales or the new vector commits it first.
                double b2 = 0.0; for (int i = 0; i < s->sr_nb; i++) b2 += s->partial_bh.h[i];
                beta = std::sqrt(std::max(b2, 0.0));
                if ((l > 0 && j == l) || j == m - 1) CHK(lz_sr_flush(s));
            } else beta = s->betas.h[kloc + 1];
            kloc++;
            Tm(j, j) = alpha;
            normT = std::max(normT, std::fabs(alpha) + beta + (j > 0 ? std::fabs(Tm(j, j - 1)) : 0.0));
            const bool arrow = l > 0 && j == l;            // H v_l couples to every kept Ritz vector
            std::vector<int> idx;
            if (arrow && beta > 0.0) {
                // first step after a restart: H v_l = sum_i sigma_i y_i + alpha v_l + beta v_{l+1} with the KNOWN sigma_i = T(i,l):
                // they are subtracted in one pass over the kept Ritz vectors (no dots); how orthogonal the result is to them is
                // what the recurrence below estimates from the defects eta_i -- a pass with measured coefficients follows only
                // if that estimate asks for it
                for (int i = 0; i < l; i++) coef[i] = Tm(i, l) * sc[i];
                const bool pina = pin && l <= kHp;
                if (pina) std::memcpy(s->eig_hp.h + kHp, coef.data(), (size_t)l * 8);      // the device is idle: this step has been waited for
                else { HIPCHK(hipMemcpyAsync(s->eig_h.p, coef.data(), (size_t)l * 8, hipMemcpyHostToDevice, s->stream)); HIPCHK(hipStreamSynchronize(s->stream)); }
                const double *cfp = pina ? cf_dev : s->eig_h.p;
                const int ngroups = std::max(1, (int)cdiv(l, kMaxBasis));
                for (int g = 0; g < ngroups; g++) {
                    const int g0 = g * kMaxBasis, gc = std::min(kMaxBasis, l - g0);
                    BasisArgs Bu{}; Bu.nv = gc; for (int i = 0; i < gc; i++) Bu.V[i] = B(g0 + i);
                    if (g == ngroups - 1) launch_multiaxpy_norm(s->stream, nb, Bu, cfp + g0, B(j + 1), vlen, s->eig_partial.p);
                    else launch_multiaxpy(s->stream, nb, Bu, cfp + g0, B(j + 1), vlen);
                    HIPCHK(hipGetLastError());
                }
                hipLaunchKernelGGL(k_reduce_final2, dim3(1), dim3(1024), 0, s->stream, (const double *)s->eig_partial.p, (int64_t)nb, s->scal.p + SC_TMP, nrm_dev);
                HIPCHK(hipGetLastError());
                CHK(allreduce_tmp(s));
                if (pin && !fused) { CHK(scalar_mode(6)); double b2 = 0.0; CHK(fetch_norm(b2)); beta = std::sqrt(std::max(b2, 0.0)); }      // = renorm(), the norm read from pinned memory
                else CHK(renorm(j, beta));
                nreads += l;
            }
            {
                // omega recurrence: estimates w(j+1,k), k < j
                const double psi = 2.0 * eps * normT;
                bool any = false, beyond = false;
                for (int k = 0; k < j; k++) {
                    double num = 0.0;
                    for (int i = 0; i <= j; i++) { const double t1 = Tm(i, k), t2 = Tm(i, j); if (t1 != 0.0) num += t1 * Wm(i, j); if (t2 != 0.0) num -= t2 * Wm(i, k); }
                    const double w = (num + std::copysign(psi + eta[k] + eta[j], num)) / std::max(beta, 1e-300);
                    Wm(j + 1, k) = Wm(k, j + 1) = w;
                    if (!(std::fabs(w) <= tau)) { any = true; if (k >= l) beyond = true; }      // also catches NaN
                }
                Wm(j + 1, j) = Wm(j, j + 1) = eps * normT / std::max(beta, 1e-300); Wm(j + 1, j + 1) = 1.0;
                if (any && beyond) { idx = all_upto(j); n_full++; }
                else if (any) { for (int k = 0; k < l; k++) if (!(std::fabs(Wm(j + 1, k)) <= sel * tau)) idx.push_back(k); n_sel++; }
            }
            if (!idx.empty() && beta > 0.0) {
                double nrm2 = 0.0;
                CHK(lz_sr_flush(s));
                std::vector<double> scl(idx.size()); for (size_t t = 0; t < idx.size(); t++) scl[t] = sc[idx[t]];
                CHK(project_out(idx, scl.data(), B(j + 1), nrm2)); nreads += 2 * (int64_t)idx.size();
                CHK(renorm(j, beta, nrm2));                // beta_j and the scale of u_{j+1} follow the new norm
                for (size_t t = 0; t < idx.size(); t++) {
                    Wm(j + 1, idx[t]) = Wm(idx[t], j + 1) = eps;
                    Fm(idx[t], j) += removed[t];                 // not part of T: feeds the defect of the Ritz vectors formed at the next restart
                }
            }
            Tm(j + 1, j) = Tm(j, j + 1) = beta; beta_last = beta;
            sc[j + 1] = beta > 0.0 ? 1.0 / beta : 0.0;
            if (beta <= 1e-13 * std::max(1.0, normT)) {
                // invariant subspace: every Ritz pair of T(0..j) is exact.  Enough of them -> done; else continue with a fresh
                // random direction orthogonal to the basis (finds further copies of degenerate levels in tiny sectors)
                Tm(j + 1, j) = Tm(j, j + 1) = 0.0; beta_last = 0.0;
                CHK(lz_sr_flush(s));
                if (j + 1 >= nev || (int64_t)(j + 1) >= Dim) { meff = j + 1; exact = true; break; }
                hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, s->stream, dense, H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, seed++);
                HIPCHK(hipGetLastError());
                CHK(lzf_copy_in(s, B(j + 1), dense));
                double nrm2 = 0.0; const std::vector<int> all = all_upto(j);
                std::vector<double> scl(sc.begin(), sc.begin() + j + 1);
                CHK(project_out(all, scl.data(), B(j + 1), nrm2));
                CHK(project_out(all, scl.data(), B(j + 1), nrm2));
                if (!(nrm2 > 0.0)) { meff = j + 1; exact = true; break; }
                { double bb = 0.0; CHK(renorm(j, bb)); }
                CHK(scalar_mode(7));
                sc[j + 1] = 1.0 / std::sqrt(nrm2); kloc = 0; V.uprev = B(j); Fz.uprev = B(j);
                for (int k = 0; k <= j; k++) Wm(j + 1, k) = Wm(k, j + 1) = eps;
                Wm(j + 1, j + 1) = 1.0;
            }
        }
        std::vector<double> Hs((size_t)meff * meff);
        for (int a = 0; a < meff; a++) for (int b = 0; b < meff; b++) Hs[(size_t)a * meff + b] = Tm(a, b);
        { const clk::time_point a = clk::now(); jacobi_eig(Hs, meff, theta, Y); ms_eig += tsec(a, clk::now()); }
        nconv = 0;
        const int want = std::min(nev, meff);
        for (int i = 0; i < want; i++) {
            const double resid = std::fabs(beta_last * Y[(meff - 1) + (size_t)meff * i]);
            if (exact || resid <= tol_eff * std::max(eps23, std::fabs(theta[i]))) nconv++;
        }
        if (nconv >= want || exact || it + 1 >= maxiter) break;
        // thick restart: lowest l Ritz vectors (scales folded into the rotation), then the residual direction u_m
        const int lnew = std::min(m - 1, nev + std::max(1, (m - nev) / 2));
        std::vector<double> Yc((size_t)m * lnew);
        for (int a = 0; a < lnew; a++) for (int k = 0; k < m; k++) Yc[k + (size_t)m * a] = sc[k] * Y[k + (size_t)m * a];
        { const clk::time_point a = clk::now(); CHK(rotate(m, lnew, Yc)); nreads += m + lnew; ms_restart_dev += tsec(a, clk::now()); }
        const clk::time_point t_rh = clk::now();
        std::swap(s->eigV[lnew], s->eigV[m]);            // B(lnew) <- u_m, no copy
        if (fused) std::swap(s->eigV[lnew + 1], s->eigV[m + 1]);      // ... and the local part of H u_m that is already there
        std::vector<double> Wn((size_t)M * M, 0.0), etan((size_t)M, 0.0);
        for (int a = 0; a < lnew; a++) {
            for (int b = 0; b <= a; b++) {
                double x = 0.0;
                for (int p = 0; p < m; p++) { double r = 0.0; for (int q = 0; q < m; q++) r += Wm(p, q) * Y[q + (size_t)m * b]; x += Y[p + (size_t)m * a] * r; }
                Wn[(size_t)a * M + b] = Wn[(size_t)b * M + a] = a == b ? 1.0 : x;
            }
            double x = 0.0, f2 = 0.0, e1 = 0.0;
            for (int p = 0; p < m; p++) {
                x += Wm(m, p) * Y[p + (size_t)m * a];
                double r = 0.0; for (int q = 0; q < m; q++) r += Fm(p, q) * Y[q + (size_t)m * a];
                f2 += r * r; e1 += std::fabs(Y[p + (size_t)m * a]) * eta[p];
            }
            Wn[(size_t)lnew * M + a] = Wn[(size_t)a * M + lnew] = x;
            etan[a] = std::sqrt(f2) + e1;                 // defect of H y_a = theta_a y_a + sigma_a v_l
        }
        Wn[(size_t)lnew * M + lnew] = 1.0;
        W.swap(Wn); eta.swap(etan); std::fill(F.begin(), F.end(), 0.0);
        const double scm = sc[m];
        std::fill(T.begin(), T.end(), 0.0);
        for (int a = 0; a < lnew; a++) { Tm(a, a) = theta[a]; Tm(a, lnew) = Tm(lnew, a) = beta_last * Y[(m - 1) + (size_t)m * a]; sc[a] = 1.0; }
        sc[lnew] = scm;
        CHK(scalar_mode(7));                              // scal[SCUR] already is the scale of u_m
        l = lnew; j = l; kloc = 0;
        V.uprev = B(l - 1); Fz.uprev = B(l - 1);           // any finite vector: its coefficient is zero
        ms_restart_host += tsec(t_rh, clk::now());
    }
    const clk::time_point t_loop = clk::now();
    const int nout = std::min(nev, meff);
    {
        std::vector<double> Yc((size_t)meff * nout);
        for (int a = 0; a < nout; a++) for (int k = 0; k < meff; k++) Yc[k + (size_t)meff * a] = sc[k] * Y[k + (size_t)meff * a];
        CHK(rotate(meff, nout, Yc));
    }
    // the Ritz vectors of a semi-orthogonal basis are orthonormal to sqrt(eps) only: Gram-Schmidt among the nout of them
    for (int i = 0; i < nout; i++) {
        double nrm2 = 0.0;
        std::vector<double> ones((size_t)std::max(i, 1), 1.0);
        std::vector<int> prev((size_t)i); for (int t = 0; t < i; t++) prev[t] = t;
        if (i > 0) CHK(project_out(prev, ones.data(), B(i), nrm2));
        else { CHK(reduce_dot(s, B(0), B(0), vlen)); HIPCHK(hipMemcpyAsync(&nrm2, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream)); }
        hipLaunchKernelGGL(k_scale_copy, dim3(gs), dim3(256), 0, s->stream, B(i), (const double *)B(i), nrm2 > 0.0 ? 1.0 / std::sqrt(nrm2) : 0.0, vlen);
        HIPCHK(hipGetLastError());
    }
    // Safety net: the TRUE residual |H x - theta x| of every returned pair (one product and one pass each).  The Lanczos estimate
    // beta |s_m| that declared them converged assumes a basis orthogonal to sqrt(eps); should that assumption ever fail the
    // residual shows it, and the call is answered by the whole-basis form instead (EDGPU_EIGSH_VERIFY=0 skips the check).
    if (nconv >= nev && nout < m && XKNOB("EDGPU_EIGSH_VERIFY", 1)) {
        double worst = 0.0;
        for (int i = 0; i < nout; i++) {
            double *w = B(m);                              // stale basis vector: free after the final rotation
            CHK(hxv(s, B(i), w, false, nullptr, s->vld));
            coef[0] = theta[i];
            if (pin) s->eig_hp.h[kHp] = theta[i];      // nothing on the device reads the coefficient area now: the last pass that did has been waited for
            else { HIPCHK(hipMemcpyAsync(s->eig_h.p, coef.data(), 8, hipMemcpyHostToDevice, s->stream)); HIPCHK(hipStreamSynchronize(s->stream)); }
            BasisArgs Bu{}; Bu.nv = 1; Bu.V[0] = B(i);
            launch_multiaxpy_norm(s->stream, nb, Bu, (const double *)(pin ? cf_dev : s->eig_h.p), w, vlen, s->eig_partial.p);
            hipLaunchKernelGGL(k_reduce_final2, dim3(1), dim3(1024), 0, s->stream, (const double *)s->eig_partial.p, (int64_t)nb, s->scal.p + SC_TMP, nrm_dev);
            HIPCHK(hipGetLastError());
            CHK(allreduce_tmp(s));
            double r2 = 0.0;
            CHK(fetch_norm(r2));
            worst = std::max(worst, std::sqrt(std::max(r2, 0.0)) / std::max(1.0, std::fabs(theta[i])));
        }
        if (s->dbg & 8) fprintf(stderr, "edgpu_eigsh: largest true residual / max(1,|theta|) of the returned pairs: %.3e\n", worst);
        if (!(worst <= 1e-6)) {
            if (s->dbg & 8) fprintf(stderr, "edgpu_eigsh: residual check failed, repeating with whole-basis re-orthogonalisation\n");
            return eigsh_full(s, nev, ncv, maxiter, tol, evals, evecs_host, nconv_out, nrestart_out, nmatvec_out);
        }
    }
    for (int i = 0; i < nev; i++) evals[i] = i < nout ? theta[i] : 0.0;
    if (evecs_host)
        for (int i = 0; i < nout; i++) {
            CHK(lzf_copy_out(s, dense, B(i)));
            HIPCHK(hipMemcpyAsync(evecs_host + (int64_t)i * n, dense, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
        }
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->dbg & 8) fprintf(stderr, "edgpu_eigsh: %d products, %d restarts, basis %d; re-orthogonalisations: %d whole basis, %d kept Ritz vectors only; %.1f basis-vector reads per product\n",
                            nmv, it + 1, m, n_full, n_sel, nmv ? (double)nreads / nmv : 0.0);
    if (s->dbg & 8) fprintf(stderr, "edgpu_eigsh: loop %.3f ms (issuing steps %.3f, waiting for the device %.3f, host work per step incl. re-orthogonalisation passes %.3f, eigenproblems of T %.3f, restart rotations %.3f, restart bookkeeping %.3f), Ritz vectors + checks %.3f ms\n",
                            tsec(t_begin, t_loop), ms_issue, ms_wait, ms_step_host, ms_eig, ms_restart_dev, ms_restart_host, tsec(t_loop, clk::now()));
    if (nconv_out) *nconv_out = nconv;
    if (nrestart_out) *nrestart_out = it + 1;
    if (nmatvec_out) *nmatvec_out = nmv;
    if (nconv < nev) return fail(EDGPU_ENOTCONVERGED, "edgpu_eigsh: " + std::to_string(nconv) + " of " + std::to_string(nev) + " pairs converged within maxiter restarts");
    return EDGPU_OK;
}

static int eigsh_full(S *s, int nev, int ncv, int maxiter, double tol, double *evals, double *evecs_host, int *nconv_out, int *nrestart_out, int *nmatvec_out);
static int eigsh(S *s, int nev, int ncv, int maxiter, double tol, double *evals, double *evecs_host, int *nconv_out, int *nrestart_out, int *nmatvec_out)
{
    if (knob("EDGPU_EIGSH_PRO")) return eigsh_pro(s, nev, ncv, maxiter, tol, evals, evecs_host, nconv_out, nrestart_out, nmatvec_out);
    return eigsh_full(s, nev, ncv, maxiter, tol, evals, evecs_host, nconv_out, nrestart_out, nmatvec_out);
}

static int eigsh_full(S *s, int nev, int ncv, int maxiter, double tol, double *evals, double *evecs_host, int *nconv_out, int *nrestart_out, int *nmatvec_out)
{
    const int64_t Dim = s->H.Dim, n = s->nloc;
    if (nev < 1) return fail(EDGPU_EINVAL, "edgpu_eigsh: nev < 1");
    if ((int64_t)nev >= Dim) return fail(EDGPU_EINVAL, "edgpu_eigsh: nev must be smaller than the sector dimension (the reference switches to LAPACK there, ED_DIAG.f90:105-107)");
    if (nev >= kMaxBasisBig) return fail(EDGPU_EUNSUPPORTED, "edgpu_eigsh: nev must be below the basis capacity (edgpu_eigsh_ncv_cap)");
    if (ncv > kMaxBasisBig) ncv = kMaxBasisBig;         // documented cap (edgpu_eigsh_ncv_cap): a smaller thick-restart basis costs restarts, not accuracy
    if ((int64_t)ncv > Dim) ncv = (int)Dim;
    if (ncv <= nev) ncv = (int)std::min<int64_t>(std::min<int64_t>(2 * nev + 1, kMaxBasisBig), Dim);
    if (ncv <= nev) return fail(EDGPU_EINVAL, "edgpu_eigsh: ncv must exceed nev");
    s->eig_ncv_used = ncv;
    if (maxiter < 1) maxiter = 1;
    const double tol_eff = std::max(tol, 1e-12), eps23 = std::pow(2.220446049250313e-16, 2.0 / 3.0);
    const int m = ncv;
    while ((int)s->eigV.size() < m + 1) s->eigV.push_back(std::make_unique<DevBuf<double>>());
    for (int i = 0; i <= m; i++)
        if (s->eigV[i]->n < (size_t)n && s->eigV[i]->alloc((size_t)n) != EDGPU_OK) {
            for (auto &b : s->eigV) if (b) b->release();      // leave nothing behind that would starve the next call
            return fail(EDGPU_ENOMEM, "edgpu_eigsh: out of device memory for basis vector " + std::to_string(i) + " of " + std::to_string(m + 1) + " (" + std::to_string(n * 8 >> 20) + " MiB each)");
        }
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256 * 8), 1024));
    if (s->eig_h.n < (size_t)(m + 3)) CHK(s->eig_h.alloc((size_t)m + 3));
    if (s->eig_partial.n < (size_t)nb * (kMaxBasis + 2)) CHK(s->eig_partial.alloc((size_t)nb * (kMaxBasis + 2)));
    if (s->eig_y.n < (size_t)m * m) CHK(s->eig_y.alloc((size_t)m * m));
    auto Vp = [&](int i) { return s->eigV[i]->p; };
    const SectorHost &H = s->H;
    const int gsr = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256 * 4), 4096));
    // V_j <- sum_i V_i Y_ij in place; bases above 48 vectors take the 96-capacity instantiation
    auto rotate = [&](int nv, int nout) -> int {
        if (knob("EDGPU_ROT_MFMA") != 0) {
            std::vector<double *> vb((size_t)nv); for (int i = 0; i < nv; i++) vb[i] = Vp(i);
            CHK(launch_basis_rotate_mfma(s->stream, s->device, vb.data(), nv, nout, s->eig_y.p, n)); }
        else if (nv <= kMaxBasis) { RotArgs R{}; R.nv = nv; R.nout = nout; R.Y = s->eig_y.p; for (int i = 0; i < nv; i++) R.V[i] = Vp(i);
            hipLaunchKernelGGL(k_basis_rotate<kMaxBasis>, dim3(gsr), dim3(256), (size_t)nv * nout * 8, s->stream, R, n); }
        else { RotArgsT<kMaxBasisBig> R{}; R.nv = nv; R.nout = nout; R.Y = s->eig_y.p; for (int i = 0; i < nv; i++) R.V[i] = Vp(i);
            static OncePerDevice once;
            if (once.first(s->device)) HIPCHK(hipFuncSetAttribute((const void *)k_basis_rotate<kMaxBasisBig>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 96 * 8));
            hipLaunchKernelGGL(k_basis_rotate<kMaxBasisBig>, dim3(gsr), dim3(256), (size_t)nv * nout * 8, s->stream, R, n); }
        HIPCHK(hipGetLastError());
        return EDGPU_OK;
    };
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
            std::vector<double *> bas(j + 1); for (int i = 0; i <= j; i++) bas[i] = Vp(i);
            std::vector<double> hj; double beta;
            // a Lanczos vector: the bulk of H v_j lies along v_j and v_{j-1}; right after a restart it couples to every kept Ritz vector
            CHK(orthogonalise_cgs(s, bas.data(), j + 1, j > l ? std::max(j - 1, 0) : 0, w, n, nb, hj, beta));
            for (int i = 0; i <= j; i++) { hh[i] = hj[i]; Hm[(size_t)i * m + j] = hh[i]; Hm[(size_t)j * m + i] = hh[i]; }
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
            if (meff < m || resid <= tol_eff * std::max(eps23, std::fabs(theta[i]))) nconv++;      // invariant subspace: every Ritz pair is exact
        }
        if (nconv >= want || meff < m || it + 1 >= maxiter) break;
        // thick restart: keep the lowest l Ritz vectors, append the residual direction
        l = std::min(m - 1, nev + std::max(1, (m - nev) / 2));
        HIPCHK(hipMemcpyAsync(s->eig_y.p, Y.data(), (size_t)m * l * 8, hipMemcpyHostToDevice, s->stream));
        CHK(rotate(m, l));
        HIPCHK(hipMemcpyAsync(Vp(l), Vp(m), (size_t)n * 8, hipMemcpyDeviceToDevice, s->stream));
        std::fill(Hm.begin(), Hm.end(), 0.0);
        for (int i = 0; i < l; i++) Hm[(size_t)i * m + i] = theta[i];
    }
    const int nout = std::min(nev, meff);
    HIPCHK(hipMemcpyAsync(s->eig_y.p, Y.data(), (size_t)meff * nout * 8, hipMemcpyHostToDevice, s->stream));
    CHK(rotate(meff, nout));
    for (int i = 0; i < nev; i++) evals[i] = i < nout ? theta[i] : 0.0;
    if (evecs_host)
        for (int i = 0; i < nout; i++) HIPCHK(hipMemcpyAsync(evecs_host + (int64_t)i * n, Vp(i), (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (nconv_out) *nconv_out = nconv;
    if (nrestart_out) *nrestart_out = it + 1;
    if (nmatvec_out) *nmatvec_out = nmv;
    // fewer converged pairs than asked for is an ERROR the caller must see (evals / evecs / nconv are filled all the same):
    // sp_eigh's callers take every returned pair for an eigenpair (ED_DIAG.f90:240-280)
    if (nconv < nev) return fail(EDGPU_ENOTCONVERGED, "edgpu_eigsh: " + std::to_string(nconv) + " of " + std::to_string(nev) + " pairs converged within maxiter restarts");
    return EDGPU_OK;
}

// ---------------------------------------------------------------------------------------
// Davidson with the diagonal preconditioner: stands in for sp_dvdson_eigh(MatVec,eval,evec,nitermax,tol) of
// ED_DIAG.f90:189-203 (lanc_method="dvdson"; the routine lives in SciFortran, not in the reference tree, so only
// its contract -- lowest Neigen pairs to a residual tolerance -- is restated).  Search space V and its image W=HV
// are kept in HBM (2 x mmax vectors); Rayleigh-Ritz on the host; one new direction per product, aimed at the lowest
// pair that has not converged yet; thick restart onto the lowest Ritz vectors when the space is full.
// ---------------------------------------------------------------------------------------
static int dvdson(S *s, int nev, int maxiter, double tol, double *evals, double *evecs_host, int *nconv_out, int *nmatvec_out)
{
    const int64_t Dim = s->H.Dim, n = s->nloc;
    if (nev < 1) return fail(EDGPU_EINVAL, "edgpu_dvdson: nev < 1");
    if ((int64_t)nev >= Dim) return fail(EDGPU_EINVAL, "edgpu_dvdson: nev must be smaller than the sector dimension (the reference switches to LAPACK there, ED_DIAG.f90:105-107)");
    // Search space: 2*nev + 4 vectors (8 for the two pairs ED_DIAG.f90:189-203 asks for) with GD+1 restarts -- the restarted space
    // keeps the lowest nev + 1 Ritz vectors AND the previous iterate of the target pair (Stathopoulos & Saad's "+k" vectors: the
    // restarted method then converges like the unrestarted one), so a small space loses no products while every pass over the
    // space is short: c3, two pairs, 162-165 products for every size between 8 and 12 and 1 or 2 extra Ritz vectors kept
    // (profiles/r04/dvdson_c3.txt; rounds 2-3 ran 16 vectors, 3.0 s; 10 vectors 2.35 s; this form 1.49 s).
    int mmax = std::min<int64_t>(std::min<int64_t>(kMaxBasis, Dim), std::max(2 * nev + 4, 8));
    { const int mx = XKNOB("EDGPU_DVD_MMAX", 0); if (mx > nev + 2) mmax = (int)std::min<int64_t>(std::min<int64_t>(kMaxBasis, Dim), mx); }      // experiments build: size of the search space
    const bool small = mmax <= 12 && s->dvd_fused_on;      // k_dvd_step: residual, correction and the Gram-Schmidt dots in one pass
    if (mmax <= nev) return fail(EDGPU_EINVAL, "edgpu_dvdson: too many eigenpairs requested for the search space");
    if (maxiter < 1) maxiter = 1;
    const double tol_eff = std::max(tol, 1e-11);
    // the diagonal of H: stored formats hold it, the on-the-fly format derives it once here
    if (s->diag.n < (size_t)s->nloc_el) CHK(build_diag(s));
    while ((int)s->eigV.size() < 2 * mmax + 1) s->eigV.push_back(std::make_unique<DevBuf<double>>());
    for (int i = 0; i <= 2 * mmax; i++)
        if (s->eigV[i]->n < (size_t)n && s->eigV[i]->alloc((size_t)n) != EDGPU_OK) {
            for (auto &b : s->eigV) if (b) b->release();
            return fail(EDGPU_ENOMEM, "edgpu_dvdson: out of device memory for vector " + std::to_string(i) + " of " + std::to_string(2 * mmax + 1) + " (search space and its image, " + std::to_string(n * 8 >> 20) + " MiB each)");
        }
    auto Vp = [&](int i) { return s->eigV[i]->p; };
    auto Wp = [&](int i) { return s->eigV[mmax + i]->p; };
    double *t = s->eigV[2 * mmax]->p;
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256 * 8), 1024));
    const int gs = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256 * 4), 4096));
    if (s->eig_h.n < (size_t)(2 * mmax + 3)) CHK(s->eig_h.alloc((size_t)2 * mmax + 3));
    if (s->eig_partial.n < (size_t)nb * (2 * mmax + 3)) CHK(s->eig_partial.alloc((size_t)nb * (2 * mmax + 3)));
    if (s->eig_y.n < (size_t)mmax * mmax) CHK(s->eig_y.alloc((size_t)mmax * mmax));
    const SectorHost &H = s->H;
    const bool rot_mfma = knob("EDGPU_ROT_MFMA") != 0;
    auto norm_of = [&](const double *x, double &out) -> int {
        CHK(reduce_dot(s, x, x));
        HIPCHK(hipMemcpyAsync(&out, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
        out = std::sqrt(out);
        return EDGPU_OK;
    };
    // dots of `w` with the first nv basis vectors -> host (one pass over w)
    std::vector<double> h2(mmax + 1);
    auto dots = [&](int nv, const double *w, bool with_W) -> int {
        BasisArgs B{}; B.nv = nv; for (int i = 0; i < nv; i++) B.V[i] = with_W ? Wp(i) : Vp(i);
        launch_multidot(s->stream, nb, B, w, n, s->eig_partial.p);
        hipLaunchKernelGGL(k_multidot_final, dim3(1), dim3(256), 0, s->stream, (const double *)s->eig_partial.p, nb, nv, s->eig_h.p);
        HIPCHK(hipGetLastError());
        CHK(allreduce_dev(s, s->eig_h.p, nv));
        HIPCHK(hipMemcpyAsync(h2.data(), s->eig_h.p, (size_t)nv * 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
        return EDGPU_OK;
    };
    hipLaunchKernelGGL(k_random_start, dim3(2048), dim3(256), 0, s->stream, Vp(0), H.DimUp, H.Qdw, H.dw0, H.DimDw, H.DimPh, (uint64_t)1234567);
    HIPCHK(hipGetLastError());
    double nrm; CHK(norm_of(Vp(0), nrm));
    hipLaunchKernelGGL(k_scale_copy, dim3(gs), dim3(256), 0, s->stream, Vp(0), (const double *)Vp(0), 1.0 / nrm, n);
    std::vector<double> G((size_t)mmax * mmax, 0.0), theta, Y, yprev;
    int m = 0, nmv = 0, nconv = 0, it = 0;
    // append V[m] (already orthonormal to the space): W[m] = H V[m], new column of G
    // gcol: the off-diagonal part of that column when it is already known (V_i . H v_new = W_i . v_new, summed inside k_dvd_step):
    // only the diagonal entry then needs the new image, one dot of two vectors instead of a pass over the whole space
    auto extend = [&](const double *gcol) -> int {
        CHK(hxv(s, Vp(m), Wp(m), false, nullptr)); nmv++;
        if (gcol) {
            double gmm = 0.0;
            CHK(reduce_dot(s, Vp(m), Wp(m), n));
            HIPCHK(hipMemcpyAsync(&gmm, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
            for (int i = 0; i < m; i++) { G[(size_t)i * mmax + m] = gcol[i]; G[(size_t)m * mmax + i] = gcol[i]; }
            G[(size_t)m * mmax + m] = gmm;
        } else {
            CHK(dots(m + 1, Wp(m), false));
            for (int i = 0; i <= m; i++) { G[(size_t)i * mmax + m] = h2[i]; G[(size_t)m * mmax + i] = h2[i]; }
        }
        m++;
        return EDGPU_OK;
    };
    const bool gcol_on = XKNOB("EDGPU_DVD_GCOL", 1) != 0;      // experiments build: 0 = every new column of the projected matrix by a pass over the space
    int n_fast = 0, n_slow = 0;
    const bool rot_fused = XKNOB("EDGPU_DVD_ROTFUSED", 1) != 0;      // experiments build: 0 = restarts by the rotation kernels
    // Thick restart (GD+1): the lowest `keep` Ritz vectors + the previous iterate of the target pair, V and W rotated alike.
    // plan_restart: the m x nout coefficient matrix from the current Ritz vectors Y (column major, ld m); adopt_restart: the host
    // state after the bases have been rotated (by k_dvd_step<.., true> or by the rotation kernels).
    struct RestartPlan { std::vector<double> Cm; int keep = 0, nout = 0; bool plus = false; double gpp = 0.0; };
    auto plan_restart = [&]() -> RestartPlan {
        RestartPlan R;
        R.keep = std::min(mmax - 2, nev + XKNOB("EDGPU_DVD_KEEP", 1));
        const int keep = R.keep;
        R.Cm.assign((size_t)m * (keep + 1), 0.0);
        for (int j = 0; j < keep; j++) for (int i = 0; i < m; i++) R.Cm[(size_t)m * j + i] = Y[(size_t)m * j + i];
        if (!yprev.empty() && (int)yprev.size() <= m) {
            std::vector<double> pv((size_t)m, 0.0);
            for (size_t i = 0; i < yprev.size(); i++) pv[i] = yprev[i];
            for (int pass = 0; pass < 2; pass++)
                for (int j = 0; j < keep; j++) { double d = 0.0; for (int i = 0; i < m; i++) d += Y[(size_t)m * j + i] * pv[i]; for (int i = 0; i < m; i++) pv[i] -= d * Y[(size_t)m * j + i]; }
            double nn = 0.0; for (int i = 0; i < m; i++) nn += pv[i] * pv[i];
            if (nn > 1e-16) {
                nn = 1.0 / std::sqrt(nn);
                for (int i = 0; i < m; i++) { pv[i] *= nn; R.Cm[(size_t)m * keep + i] = pv[i]; }
                for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) R.gpp += pv[i] * G[(size_t)i * mmax + j] * pv[j];
                R.plus = true;
            }
        }
        R.nout = keep + (R.plus ? 1 : 0);
        return R;
    };
    auto adopt_restart = [&](const RestartPlan &R, std::vector<double> &hfirst, double t2, double r2) -> int {
        std::fill(G.begin(), G.end(), 0.0);
        for (int i = 0; i < R.keep; i++) G[(size_t)i * mmax + i] = theta[i];
        if (R.plus) G[(size_t)R.keep * mmax + R.keep] = R.gpp;     // the Ritz vectors are G-orthogonal to everything orthogonal to them
        if (small && !hfirst.empty()) {
            // the correction was computed against the old basis and is still a valid new direction; its dots with the
            // rotated bases follow from the old ones (V' = V C, W' = W C  =>  V'^T t = C^T V^T t): no pass over the space
            const int nout = R.nout;
            std::vector<double> hr((size_t)2 * nout + 2, 0.0);
            for (int j = 0; j < nout; j++) for (int i = 0; i < m; i++) { hr[j] += R.Cm[(size_t)m * j + i] * hfirst[i]; hr[nout + j] += R.Cm[(size_t)m * j + i] * hfirst[m + i]; }
            hr[2 * nout] = t2; hr[2 * nout + 1] = r2;
            hfirst.swap(hr);
            HIPCHK(hipMemcpyAsync(s->eig_h.p, hfirst.data(), (size_t)nout * 8, hipMemcpyHostToDevice, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
        }
        m = R.nout;
        yprev.clear();
        return EDGPU_OK;
    };
    CHK(extend(nullptr));
    for (it = 0; it < maxiter; it++) {
        std::vector<double> Gs((size_t)m * m);
        for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Gs[(size_t)i * m + j] = G[(size_t)i * mmax + j];
        jacobi_eig(Gs, m, theta, Y);
        const int k = std::min(nconv, m - 1);                 // target pair
        // residual of pair k and the preconditioned correction, one pass over V and W
        HIPCHK(hipMemcpyAsync(s->eig_y.p, Y.data() + (size_t)m * k, (size_t)m * 8, hipMemcpyHostToDevice, s->stream));
        DvdArgs D{}; D.nv = m; for (int i = 0; i < m; i++) { D.V[i] = Vp(i); D.W[i] = Wp(i); }
        D.y = s->eig_y.p; D.theta = theta[k]; D.diag = s->diag.p; D.nel = s->nloc_el; D.w0 = H.DimPh > 1 ? H.desc.w0_ph : 0.0;
        double r2 = 0.0, t2 = 0.0; std::vector<double> hfirst;
        bool restarted = false;
        if (small) {      // ... and the dots V_i . t of the first Gram-Schmidt pass, W_i . t, |t|^2, |r|^2 on the way
            // a full space is restarted by this very pass (k_dvd_step<.., true>): what the restart keeps depends on the Ritz
            // vectors only, not on the residual that is about to be computed
            RestartPlan R;
            bool rot = false;
            if (m == mmax && rot_fused) { R = plan_restart(); rot = R.nout <= kDvdRotOut; }
            if (rot) {
                HIPCHK(hipMemcpyAsync(s->eig_y.p + mmax, R.Cm.data(), (size_t)m * R.nout * 8, hipMemcpyHostToDevice, s->stream));
            }
            launch_dvd_step(s->stream, nb, D, t, n, s->eig_partial.p, rot ? (const double *)(s->eig_y.p + mmax) : nullptr, rot ? R.nout : 0);
            hipLaunchKernelGGL(k_multidot_final, dim3(1), dim3(256), 0, s->stream, (const double *)s->eig_partial.p, nb, 2 * m + 2, s->eig_h.p);
            HIPCHK(hipGetLastError());
            CHK(allreduce_dev(s, s->eig_h.p, 2 * m + 2));
            hfirst.resize((size_t)2 * m + 2);      // [0,m): V_i.t   [m,2m): W_i.t   2m: |t|^2   2m+1: |r|^2
            HIPCHK(hipMemcpyAsync(hfirst.data(), s->eig_h.p, (size_t)(2 * m + 2) * 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
            t2 = hfirst[2 * m]; r2 = hfirst[2 * m + 1];
            if (rot) { CHK(adopt_restart(R, hfirst, t2, r2)); restarted = true; }      // the bases are the rotated ones from here on, converged or not
        } else {
            hipLaunchKernelGGL(k_dvd_resid, dim3(nb), dim3(256), 0, s->stream, D, t, n, s->eig_partial.p);
            hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, (const double *)s->eig_partial.p, (int64_t)nb, s->scal.p + SC_TMP);
            HIPCHK(hipGetLastError());
            CHK(allreduce_tmp(s));
            HIPCHK(hipMemcpyAsync(&r2, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
        }
        if (std::sqrt(r2) <= tol_eff * std::max(1.0, std::fabs(theta[k]))) {
            nconv++;
            yprev.clear();
            if (nconv >= nev || nconv >= m) break;
            continue;                                          // next pair, same space
        }
        if (m == mmax && !restarted) {     // thick restart by the rotation kernels (large spaces, or the fused form switched off)
            RestartPlan R = plan_restart();
            HIPCHK(hipMemcpyAsync(s->eig_y.p, R.Cm.data(), (size_t)m * R.nout * 8, hipMemcpyHostToDevice, s->stream));
            for (int half = 0; half < 2; half++) {
                if (rot_mfma) { std::vector<double *> vb((size_t)m); for (int i = 0; i < m; i++) vb[i] = half ? Wp(i) : Vp(i);
                    CHK(launch_basis_rotate_mfma(s->stream, s->device, vb.data(), m, R.nout, s->eig_y.p, n)); continue; }
                RotArgs Rt{}; Rt.nv = m; Rt.nout = R.nout; Rt.Y = s->eig_y.p; for (int i = 0; i < m; i++) Rt.V[i] = half ? Wp(i) : Vp(i);
                hipLaunchKernelGGL(k_basis_rotate<kMaxBasis>, dim3(gs), dim3(256), (size_t)m * R.nout * 8, s->stream, Rt, n);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(s->stream));          // R.Cm lives on this frame
            CHK(adopt_restart(R, hfirst, t2, r2));
            restarted = true;
        } else if (!restarted) {
            yprev.assign(Y.begin() + (size_t)m * k, Y.begin() + (size_t)m * (k + 1));      // this iterate, in the basis of this iteration (the next one has one vector more: coefficient 0)
        }
        // orthogonalise t against the space, normalise, append
        double tn = 0.0;
        bool done_orth = false;
        if (small && gcol_on) {
            // |t - sum h_i V_i|^2 = |t|^2 - sum h_i^2 for an orthonormal space; when that difference keeps at least half of |t|^2
            // it is no cancellation (relative error a few ulp) AND one Gram-Schmidt pass is enough (DGKS): the new basis vector
            // is written normalised in the same pass, and its row of the projected matrix follows from the W_i . t.
            double sh = 0.0; for (int i = 0; i < m; i++) sh += hfirst[i] * hfirst[i];
            const double est = t2 - sh;
            if (est >= 0.5 * t2 && est > 1e-28) {
                const double sc = 1.0 / std::sqrt(est);
                BasisArgs Bu{}; Bu.nv = m; for (int i = 0; i < m; i++) Bu.V[i] = Vp(i);
                launch_dvd_orth(s->stream, nb, Bu, (const double *)s->eig_h.p, (const double *)t, Vp(m), sc, n, s->eig_partial.p);
                hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, (const double *)s->eig_partial.p, (int64_t)nb, s->scal.p + SC_TMP);
                HIPCHK(hipGetLastError());
                CHK(allreduce_tmp(s));
                double after = 0.0;
                HIPCHK(hipMemcpyAsync(&after, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
                if (after >= 0.5 * t2 && std::fabs(after - est) <= 1e-8 * t2) {      // what was summed agrees with what was predicted: |v_new| = 1 to rounding
                    std::vector<double> gcol((size_t)m);
                    for (int i = 0; i < m; i++) { double a = hfirst[m + i]; for (int j = 0; j < m; j++) a -= hfirst[j] * G[(size_t)i * mmax + j]; gcol[i] = a * sc; }
                    n_fast++;
                    CHK(extend(gcol.data()));
                    continue;
                }
                // the space has lost orthogonality beyond what the prediction tolerates: orthogonalise v_new itself, the long way
                std::vector<double *> bas(m); for (int i = 0; i < m; i++) bas[i] = Vp(i);
                std::vector<double> hj; double vn = 0.0;
                CHK(orthogonalise_cgs(s, bas.data(), m, 0, Vp(m), n, nb, hj, vn));
                if (!(vn > 1e-14)) break;
                hipLaunchKernelGGL(k_scale_copy, dim3(gs), dim3(256), 0, s->stream, t, (const double *)Vp(m), 1.0 / vn, n);
                hipLaunchKernelGGL(k_scale_copy, dim3(gs), dim3(256), 0, s->stream, Vp(m), (const double *)t, 1.0, n);
                HIPCHK(hipGetLastError());
                n_slow++;
                CHK(extend(nullptr));
                continue;
            }
        }
        n_slow++;
        if (small && !restarted) {      // first pass with the dots that came with the correction; DGKS: a second, full pass only when the vector shrank
            BasisArgs Bu{}; Bu.nv = m; for (int i = 0; i < m; i++) Bu.V[i] = Vp(i);
            launch_multiaxpy_norm(s->stream, nb, Bu, (const double *)s->eig_h.p, t, n, s->eig_partial.p);
            hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, (const double *)s->eig_partial.p, (int64_t)nb, s->scal.p + SC_TMP);
            HIPCHK(hipGetLastError());
            CHK(allreduce_tmp(s));
            double after = 0.0;
            HIPCHK(hipMemcpyAsync(&after, s->scal.p + SC_TMP, 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
            if (after >= 0.5 * t2) { tn = std::sqrt(after > 0.0 ? after : 0.0); done_orth = true; }
        }
        if (!done_orth) {
            std::vector<double *> bas(m); for (int i = 0; i < m; i++) bas[i] = Vp(i);
            std::vector<double> hj;
            CHK(orthogonalise_cgs(s, bas.data(), m, 0, t, n, nb, hj, tn));
        }
        if (!(tn > 1e-14)) break;                              // no new direction: the space is invariant
        hipLaunchKernelGGL(k_scale_copy, dim3(gs), dim3(256), 0, s->stream, Vp(m), (const double *)t, 1.0 / tn, n);
        HIPCHK(hipGetLastError());
        CHK(extend(nullptr));
    }
    if (knob("EDGPU_DEBUG") & 8) fprintf(stderr, "edgpu_dvdson: %d products, %d steps in two passes over the space (column of the projected matrix from W.t), %d the long way\n", nmv, n_fast, n_slow);
    // Ritz vectors of the lowest nev pairs
    {
        std::vector<double> Gs((size_t)m * m);
        for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Gs[(size_t)i * m + j] = G[(size_t)i * mmax + j];
        jacobi_eig(Gs, m, theta, Y);
    }
    const int nout = std::min(nev, m);
    HIPCHK(hipMemcpyAsync(s->eig_y.p, Y.data(), (size_t)m * nout * 8, hipMemcpyHostToDevice, s->stream));
    if (rot_mfma) { std::vector<double *> vb((size_t)m); for (int i = 0; i < m; i++) vb[i] = Vp(i);
        CHK(launch_basis_rotate_mfma(s->stream, s->device, vb.data(), m, nout, s->eig_y.p, n)); }
    else { RotArgs Rt{}; Rt.nv = m; Rt.nout = nout; Rt.Y = s->eig_y.p; for (int i = 0; i < m; i++) Rt.V[i] = Vp(i);
        hipLaunchKernelGGL(k_basis_rotate<kMaxBasis>, dim3(gs), dim3(256), (size_t)m * nout * 8, s->stream, Rt, n); }
    HIPCHK(hipGetLastError());
    for (int i = 0; i < nev; i++) evals[i] = i < nout ? theta[i] : 0.0;
    if (evecs_host)
        for (int i = 0; i < nout; i++) HIPCHK(hipMemcpyAsync(evecs_host + (int64_t)i * n, Vp(i), (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (nconv_out) *nconv_out = std::min(nconv, nout);
    if (nmatvec_out) *nmatvec_out = nmv;
    if (std::min(nconv, nout) < nev) return fail(EDGPU_ENOTCONVERGED, "edgpu_dvdson: " + std::to_string(std::min(nconv, nout)) + " of " + std::to_string(nev) + " pairs converged within maxiter products");
    return EDGPU_OK;
}

}  // namespace edgpu
