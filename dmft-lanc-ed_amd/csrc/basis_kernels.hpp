// basis_kernels.hpp -- gfx950 kernels of the multi-eigenpair solvers (thick-restart Lanczos, Davidson) and of the stored-Krylov
// eigenvector sum: multi-dot / multi-axpy over a basis held in HBM, basis rotations (plain and on the f64 matrix cores), the
// Davidson step, start vectors.  Split from kernels.hpp (the sector operator and the Lanczos vector updates) so that the digest
// bench.py stamps the counter summaries with (PRODUCT_SOURCES) only moves when a product kernel does.
#pragma once

namespace edgpu {

// ---------------------------------------------------------------------------------------------
// Krylov-basis kernels of the multi-eigenpair solver (thick-restart Lanczos, eigsh() in edgpu.hip)
// ---------------------------------------------------------------------------------------------
constexpr int kMaxBasis = 48;
struct BasisArgs { const double *V[kMaxBasis + 1]; int32_t nv; };

// partial[b*nv + i] = sum over the block's elements of V_i . w     (one pass over w)
// Two elements per lane (16-B loads) and the basis in groups of four vectors whose loads are issued together: a guard per
// vector would put every load in its own basic block and serialise them (the first form of this kernel ran at 2.6-3.6 TB/s).
// NG: groups compiled in (nv <= 4*NG): small bases get a small register footprint, i.e. more resident waves -- a 3-vector
// dot through the 13-group instance keeps only 24 KB per CU in flight.
constexpr int kBasisGroups = (kMaxBasis + 1 + 3) / 4;
template <int NG>
__global__ void __launch_bounds__(256) k_multidot(BasisArgs B, const double *__restrict__ w, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[4];
    double acc[NG * 4];
#pragma unroll
    for (int i = 0; i < NG * 4; i++) acc[i] = 0.0;
    const int nv = B.nv;
    const int64_t n2 = n >> 1;
    const dbl2_t *w2 = reinterpret_cast<const dbl2_t *>(w);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n2; p += (int64_t)gridDim.x * blockDim.x) {
        const dbl2_t x = w2[p];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int lim = (kMaxBasis + 1 - g * 4) < 4 ? (kMaxBasis + 1 - g * 4) : 4;      // compile time after unrolling
            if (lim == 4 && (g + 1) * 4 <= nv) {
                dbl2_t v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = reinterpret_cast<const dbl2_t *>(B.V[g * 4 + j < kMaxBasis + 1 ? g * 4 + j : 0])[p];
#pragma unroll
                for (int j = 0; j < 4; j++) acc[g * 4 + j] += v[j].x * x.x + v[j].y * x.y;
            } else if (g * 4 < nv) {
#pragma unroll
                for (int j = 0; j < 4; j++) if (j < lim && g * 4 + j < nv) { const dbl2_t v = reinterpret_cast<const dbl2_t *>(B.V[g * 4 + j < kMaxBasis + 1 ? g * 4 + j : 0])[p]; acc[g * 4 + j] += v.x * x.x + v.y * x.y; }
            }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {     // odd length: the last element
        const double x = w[n - 1];
#pragma unroll
        for (int i = 0; i < NG * 4 && i <= kMaxBasis; i++) if (i < nv) acc[i] += B.V[i][n - 1] * x;
    }
#pragma unroll
    for (int i = 0; i < NG * 4 && i <= kMaxBasis; i++) {     // unrolled with a uniform guard: acc[] stays in registers (no scratch)
        if (i < nv) {
            double r = block_sum(acc[i], red);
            if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * nv + i] = r;
        }
    }
}
// out[i] = sum_b partial[b*nv+i], single workgroup, fixed order
__global__ void __launch_bounds__(256) k_multidot_final(const double *__restrict__ partial, int nb, int nv, double *__restrict__ out)
{
    __shared__ double red[4];
    for (int i = 0; i < nv; i++) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nb; b += blockDim.x) s += partial[(int64_t)b * nv + i];
        double r = block_sum(s, red);
        if (threadIdx.x == 0) out[i] = r;
        __syncthreads();
    }
}
// w -= sum_i h[i] V_i      (NORM: |w_new|^2 accumulated on the way, block partials in fixed order: saves the separate norm pass)
// out[0] = out2[0] = sum(in[0..n)) in the order of k_reduce_final: the device keeps the value where its scalar steps read it, the host
// gets it in pinned memory without a copy call (out2 may be null)
__global__ void __launch_bounds__(1024) k_reduce_final2(const double *__restrict__ in, int64_t n, double *__restrict__ out, double *__restrict__ out2)
{
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    double r = block_sum(s, red);
    if (threadIdx.x == 0) { *out = r; if (out2) *out2 = r; }
}
// Same shape as k_multidot: two elements per lane, groups of four vectors loaded together; coefficients from LDS (a uniform
// read is a broadcast) instead of ~100 scalar registers.
template <bool NORM, int NG>
__device__ __forceinline__ void multiaxpy_body(const BasisArgs &B, const double *__restrict__ h, double *__restrict__ w, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[4];
    __shared__ double cs[kBasisGroups * 4];
    const int nv = B.nv;
    for (int i = threadIdx.x; i < kBasisGroups * 4; i += blockDim.x) cs[i] = i < nv ? h[i] : 0.0;
    __syncthreads();
    const int64_t n2 = n >> 1;
    dbl2_t *w2 = reinterpret_cast<dbl2_t *>(w);
    double ss = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n2; p += (int64_t)gridDim.x * blockDim.x) {
        dbl2_t x = w2[p];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int lim = (kMaxBasis + 1 - g * 4) < 4 ? (kMaxBasis + 1 - g * 4) : 4;
            if (lim == 4 && (g + 1) * 4 <= nv) {
                dbl2_t v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = reinterpret_cast<const dbl2_t *>(B.V[g * 4 + j < kMaxBasis + 1 ? g * 4 + j : 0])[p];
#pragma unroll
                for (int j = 0; j < 4; j++) { const double c = cs[g * 4 + j]; x.x -= c * v[j].x; x.y -= c * v[j].y; }
            } else if (g * 4 < nv) {
#pragma unroll
                for (int j = 0; j < 4; j++) if (j < lim && g * 4 + j < nv) { const dbl2_t v = reinterpret_cast<const dbl2_t *>(B.V[g * 4 + j < kMaxBasis + 1 ? g * 4 + j : 0])[p]; const double c = cs[g * 4 + j]; x.x -= c * v.x; x.y -= c * v.y; }
            }
        }
        w2[p] = x;
        if (NORM) ss += x.x * x.x + x.y * x.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double x = w[n - 1];
#pragma unroll
        for (int i = 0; i <= kMaxBasis; i++) if (i < nv) x -= cs[i] * B.V[i][n - 1];
        w[n - 1] = x;
        if (NORM) ss += x * x;
    }
    if (NORM) { double r = block_sum(ss, red); if (threadIdx.x == 0) partial[blockIdx.x] = r; }
}
template <int NG>
__global__ void __launch_bounds__(256) k_multiaxpy(BasisArgs B, const double *__restrict__ h, double *__restrict__ w, int64_t n)
{
    multiaxpy_body<false, NG>(B, h, w, n, nullptr);
}
template <int NG>
__global__ void __launch_bounds__(256) k_multiaxpy_norm(BasisArgs B, const double *__restrict__ h, double *__restrict__ w, int64_t n, double *__restrict__ partial)
{
    multiaxpy_body<true, NG>(B, h, w, n, partial);
}
// host-side dispatch on the number of vectors
#define EDGPU_NG_DISPATCH(nv, CALL) do { const int ng_ = ((nv) + 3) / 4; if (ng_ <= 1) { CALL(1); } else if (ng_ <= 2) { CALL(2); } else if (ng_ <= 4) { CALL(4); } else if (ng_ <= 6) { CALL(6); } else { CALL(kBasisGroups); } } while (0)
static inline void launch_multidot(hipStream_t st, int nb, const BasisArgs &B, const double *w, int64_t n, double *partial)
{
#define C_(NG) hipLaunchKernelGGL((k_multidot<NG>), dim3(nb), dim3(256), 0, st, B, w, n, partial)
    EDGPU_NG_DISPATCH(B.nv, C_);
#undef C_
}
static inline void launch_multiaxpy(hipStream_t st, int nb, const BasisArgs &B, const double *h, double *w, int64_t n)
{
#define C_(NG) hipLaunchKernelGGL((k_multiaxpy<NG>), dim3(nb), dim3(256), 0, st, B, h, w, n)
    EDGPU_NG_DISPATCH(B.nv, C_);
#undef C_
}
static inline void launch_multiaxpy_norm(hipStream_t st, int nb, const BasisArgs &B, const double *h, double *w, int64_t n, double *partial)
{
#define C_(NG) hipLaunchKernelGGL((k_multiaxpy_norm<NG>), dim3(nb), dim3(256), 0, st, B, h, w, n, partial)
    EDGPU_NG_DISPATCH(B.nv, C_);
#undef C_
}
// in-place rotation of the basis: V_j <- sum_i V_i Y[i + nv*j], j < nout  (row by row: all nv values of an element sit in
// registers, so the capacity is a template parameter: 48 for the usual bases, 96 for the large ones of finite-T runs)
constexpr int kMaxBasisBig = 96;
// Rotation of a basis of up to 4*NG vectors with two elements per lane (16-byte accesses) and every workgroup on one contiguous
// piece of the vectors; coefficients broadcast from LDS as in the general form below (read through the scalar cache instead they
// made the kernel slower: 1.13 against 0.92 s for the Ns=16 eigensolver call).
template <int NG> struct RotArgs2 { double *V[4 * NG]; int32_t nv, nout; const double *Y; };
template <int NG>
__global__ void __launch_bounds__(256) k_basis_rotate2(RotArgs2<NG> R, int64_t n)
{
    extern __shared__ double ysh[];     // nv*nout
    for (int t = threadIdx.x; t < R.nv * R.nout; t += blockDim.x) ysh[t] = R.Y[t];
    __syncthreads();
    const int64_t n2 = n >> 1;
    const int64_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const int64_t e1 = (int64_t)(blockIdx.x + 1) * per < n2 ? (int64_t)(blockIdx.x + 1) * per : n2;
    for (int64_t e = (int64_t)blockIdx.x * per + threadIdx.x; e < e1; e += blockDim.x) {
        dbl2_t x[4 * NG];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            if ((g + 1) * 4 <= R.nv) {
#pragma unroll
                for (int j = 0; j < 4; j++) x[g * 4 + j] = reinterpret_cast<const dbl2_t *>(R.V[g * 4 + j])[e];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) { dbl2_t z = {0.0, 0.0}; x[g * 4 + j] = (g * 4 + j < R.nv) ? reinterpret_cast<const dbl2_t *>(R.V[g * 4 + j])[e] : z; }
            }
        }
        for (int j = 0; j < R.nout; j++) {
            const double *yj = ysh + R.nv * j;
            dbl2_t acc = {0.0, 0.0};
#pragma unroll
            for (int i = 0; i < 4 * NG; i++) if (i < R.nv) { const double c = yj[i]; acc.x += x[i].x * c; acc.y += x[i].y * c; }
            reinterpret_cast<dbl2_t *>(R.V[j])[e] = acc;
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {      // odd length: the last element
        double x[4 * NG];
        for (int i = 0; i < R.nv; i++) x[i] = R.V[i][n - 1];
        for (int j = 0; j < R.nout; j++) { double a = 0.0; for (int i = 0; i < R.nv; i++) a += x[i] * ysh[i + R.nv * j]; R.V[j][n - 1] = a; }
    }
}

template <int MAXB> struct RotArgsT { double *V[MAXB + 1]; int32_t nv, nout; const double *Y; };
using RotArgs = RotArgsT<kMaxBasis>;
template <int MAXB>
__global__ void __launch_bounds__(256) k_basis_rotate(RotArgsT<MAXB> R, int64_t n)
{
    extern __shared__ double ysh[];     // nv*nout
    for (int t = threadIdx.x; t < R.nv * R.nout; t += blockDim.x) ysh[t] = R.Y[t];
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        double x[((MAXB + 1 + 3) / 4) * 4];
        // the basis in groups of four vectors whose loads are issued together (a guard per vector serialises the loads)
#pragma unroll
        for (int g = 0; g < (MAXB + 1 + 3) / 4; g++) {
            const int lim = (MAXB + 1 - g * 4) < 4 ? (MAXB + 1 - g * 4) : 4;
            if (lim == 4 && (g + 1) * 4 <= R.nv) {
#pragma unroll
                for (int j = 0; j < 4; j++) x[g * 4 + j] = R.V[g * 4 + j <= MAXB ? g * 4 + j : 0][e];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) x[g * 4 + j] = (j < lim && g * 4 + j < R.nv) ? R.V[g * 4 + j <= MAXB ? g * 4 + j : 0][e] : 0.0;
            }
        }
        for (int j = 0; j < R.nout; j++) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i <= MAXB; i++) if (i < R.nv) s += x[i] * ysh[i + R.nv * j];
            R.V[j][e] = s;
        }
    }
}
// Rotation of a large basis on the matrix cores: V_j <- sum_i V_i Y[i + nv*j] is a tall-skinny GEMM (n x nv times nv x nout,
// n ~ 1e8), in place.  v_mfma_f64_16x16x4_f64 with rows = 16 outputs, columns = 16 consecutive elements, k = 4 inputs:
//   A lane l = Y(input 4ks + (l>>4), output 16ot + (l&15))   -- staged in LDS once per workgroup, fragment-major (64 consecutive
//                                                                doubles per (ot, ks): conflict-free 512-byte reads)
//   B lane l = V[4ks + (l>>4)][e0 + (l&15)]                   -- one 8-byte load per lane and k-step, 128-byte segments
//   D lane l, register r = output 16ot + (l>>4) + 4r, element e0 + (l&15)
// A wave owns a 16-element tile: it reads all nv inputs of the tile (KU k-steps of loads in flight), keeps the OT output tiles
// in 8*OT accumulator registers and only then writes, so the rotation can overwrite its own inputs.  The one-element kernels
// above hold the whole basis of an element in registers (200 VGPRs at 96 vectors: 212 ms per rotation of 80 -> 40 vectors at
// Ns=16, 7x the time the 160 GB take); this one needs 8*OT + 2*KU registers whatever the basis size.  No branch around a load:
// k-steps and pointers are padded (zero coefficients, a valid address), elements beyond n load the last one.
typedef double dbl4_t __attribute__((ext_vector_type(4)));
struct RotMfmaArgs { double *V[kMaxBasisBig + 8]; int32_t nv, nout, nks; const double *Y; };      // nks: k-steps, padded to a multiple of KU
template <int OT, int KU>
__global__ void __launch_bounds__(256) k_basis_rotate_mfma(RotMfmaArgs R, int64_t n)
{
    extern __shared__ double ysh[];     // [OT][nks][64] coefficient fragments, then the pointer table (4*nks entries)
    const int nks = R.nks, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double **vp = reinterpret_cast<double **>(ysh + (size_t)OT * nks * 64);
    for (int t = threadIdx.x; t < OT * nks * 64; t += blockDim.x) {
        const int l = t & 63, ks = (t >> 6) % nks, ot = (t >> 6) / nks;
        const int i = 4 * ks + (l >> 4), j = 16 * ot + (l & 15);
        ysh[t] = (i < R.nv && j < R.nout) ? R.Y[i + R.nv * j] : 0.0;
    }
    for (int t = threadIdx.x; t < 4 * nks; t += blockDim.x) vp[t] = R.V[t < R.nv ? t : 0];
    __syncthreads();
    const int64_t ntiles = (n + 15) >> 4, nw = (int64_t)gridDim.x * 4;
    const int64_t per = (ntiles + nw - 1) / nw, w = (int64_t)blockIdx.x * 4 + wave;      // each wave walks one contiguous piece
    const int64_t t1 = (w + 1) * per < ntiles ? (w + 1) * per : ntiles;
    for (int64_t tile = w * per; tile < t1; tile++) {
        const int64_t e = tile * 16 + (lane & 15), ec = e < n ? e : n - 1;
        dbl4_t acc[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ot++) acc[ot] = dbl4_t{0.0, 0.0, 0.0, 0.0};
        for (int ks0 = 0; ks0 < nks; ks0 += KU) {
            double x[KU];
#pragma unroll
            for (int u = 0; u < KU; u++) x[u] = vp[4 * (ks0 + u) + (lane >> 4)][ec];
#pragma unroll
            for (int u = 0; u < KU; u++) {
#pragma unroll
                for (int ot = 0; ot < OT; ot++)
                    acc[ot] = __builtin_amdgcn_mfma_f64_16x16x4f64(ysh[((size_t)ot * nks + ks0 + u) * 64 + lane], x[u], acc[ot], 0, 0, 0);
            }
        }
        if (e < n) {
#pragma unroll
            for (int ot = 0; ot < OT; ot++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int j = 16 * ot + (lane >> 4) + 4 * r;
                    if (j < R.nout) vp[j][e] = acc[ot][r];      // nout <= nv: the table holds that pointer
                }
            }
        }
    }
}
// y = a*x
// Davidson step (dvdson() in lanczos.hpp): residual of the Ritz pair (theta, V y) and the diagonally preconditioned
// correction in one pass over the two bases:
//   r_e = sum_j y_j (W_j[e] - theta V_j[e]) ,  t_e = r_e / (theta - d_e)  (|theta - d_e| kept away from zero),
//   partial[b] = sum r_e^2 over the block.    d_e = diag[e mod nel] + w0 * (e div nel)  (phonon blocks share the electronic diagonal)
struct DvdArgs { double *V[kMaxBasis + 1]; double *W[kMaxBasis + 1]; int32_t nv;      /* V, W: read by every step, written in place by the restart rotation of k_dvd_step<.., true> */
                  const double *y; double theta; const double *diag; int64_t nel; double w0; };
__global__ void __launch_bounds__(256) k_dvd_resid(DvdArgs A, double *__restrict__ t, int64_t n, double *__restrict__ partial)
{
    // two elements per lane, the two bases in groups of four vector pairs loaded together, coefficients from LDS (as k_multidot)
    __shared__ double red[4];
    __shared__ double cs[kBasisGroups * 4];
    const int nv = A.nv;
    for (int i = threadIdx.x; i < kBasisGroups * 4; i += blockDim.x) cs[i] = i < nv ? A.y[i] : 0.0;
    __syncthreads();
    const double theta = A.theta;
    auto finish = [&](int64_t e, double r) -> double {
        const int64_t iph = A.nel >= n ? 0 : e / A.nel;
        double den = theta - (A.diag[e - iph * A.nel] + A.w0 * (double)iph);
        if (fabs(den) < 1e-8) den = den < 0.0 ? -1e-8 : 1e-8;
        return r / den;
    };
    double rr = 0.0;
    const int64_t n2 = n >> 1;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n2; p += (int64_t)gridDim.x * blockDim.x) {
        dbl2_t r; r.x = 0.0; r.y = 0.0;
#pragma unroll 1      // unrolled over all 13 groups the compiler hoists every load of the element: 359 registers, one wave per SIMD
        for (int g = 0; g < kBasisGroups; g++) {
            if (g * 4 >= nv) break;
            const int lim = (kMaxBasis + 1 - g * 4) < 4 ? (kMaxBasis + 1 - g * 4) : 4;
            if (lim == 4 && (g + 1) * 4 <= nv) {
                dbl2_t wv[4], vv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { const int i = g * 4 + j <= kMaxBasis ? g * 4 + j : 0; wv[j] = reinterpret_cast<const dbl2_t *>(A.W[i])[p]; vv[j] = reinterpret_cast<const dbl2_t *>(A.V[i])[p]; }
#pragma unroll
                for (int j = 0; j < 4; j++) { const double c = cs[g * 4 + j]; r.x += c * (wv[j].x - theta * vv[j].x); r.y += c * (wv[j].y - theta * vv[j].y); }
            } else if (g * 4 < nv) {
#pragma unroll
                for (int j = 0; j < 4; j++) if (j < lim && g * 4 + j < nv) {
                    const int i = g * 4 + j <= kMaxBasis ? g * 4 + j : 0;
                    const dbl2_t wv = reinterpret_cast<const dbl2_t *>(A.W[i])[p], vv = reinterpret_cast<const dbl2_t *>(A.V[i])[p];
                    const double c = cs[g * 4 + j]; r.x += c * (wv.x - theta * vv.x); r.y += c * (wv.y - theta * vv.y);
                }
            }
        }
        dbl2_t o; o.x = finish(2 * p, r.x); o.y = finish(2 * p + 1, r.y);
        reinterpret_cast<dbl2_t *>(t)[p] = o;
        rr += r.x * r.x + r.y * r.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t e = n - 1; double r = 0.0;
#pragma unroll
        for (int i = 0; i <= kMaxBasis; i++) if (i < nv) r += cs[i] * (A.W[i][e] - theta * A.V[i][e]);
        t[e] = finish(e, r); rr += r * r;
    }
    double s = block_sum(rr, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// Davidson step for SMALL search spaces (nv <= NV <= 12; round 4): residual, preconditioned correction AND the first Gram-Schmidt
// dots in ONE pass over V and W.  The V values of an element stay in registers until its correction t is known, so that
// h_i = V_i . t (and |t|^2, |r|^2) are summed on the way: the separate multi-dot pass over the basis (nv more vector reads per
// iteration) is gone.  The W values are in registers as well, so g_i = W_i . t rides along: with H symmetric,
// V_i . (H v_new) = W_i . v_new, and v_new is a known combination of t and the V_j -- the new column of the projected matrix
// follows on the host without the pass over V that used to compute it (dvdson(), lanczos.hpp).
// partial[b*(2nv+2) + i]: i < nv: V_i.t ; nv <= i < 2nv: W_(i-nv).t ; 2nv: |t|^2 ; 2nv+1: |r|^2  (layout of k_multidot_final).
// ROT (a restart is due): every value of V and W at an element is in registers here, so the thick-restart rotation
// V'_j = sum_i V_i C[i + nv*j], W'_j likewise (j < nout <= 6), is written from them in place -- the restart costs its 2*nout vector
// writes instead of two more passes over both bases (k_basis_rotate* read 2*nv and write 2*nout vectors).  The dots come out
// against the OLD basis; the host rotates them (V'^T t = C^T V^T t).
constexpr int kDvdRotOut = 6;
template <int NV, bool ROT>
__global__ void __launch_bounds__(256) k_dvd_step(DvdArgs A, double *__restrict__ t, int64_t n, double *__restrict__ partial, const double *__restrict__ C, int nout)
{
    __shared__ double red[4];
    __shared__ double cs[NV];
    __shared__ double cr[ROT ? NV * kDvdRotOut : 1];
    constexpr int nv = NV;      // instances are compiled for the exact size of the space (launch_dvd_step): no guards, base pointers stay scalar
    for (int i = threadIdx.x; i < NV; i += blockDim.x) cs[i] = i < nv ? A.y[i] : 0.0;
    if (ROT) for (int q = threadIdx.x; q < NV * kDvdRotOut; q += blockDim.x) { const int j = q / NV, i = q - j * NV; cr[q] = (i < nv && j < nout) ? C[i + nv * j] : 0.0; }
    __syncthreads();
    const double theta = A.theta;
    auto finish = [&](int64_t e, double r) -> double {
        const int64_t iph = A.nel >= n ? 0 : e / A.nel;
        double den = theta - (A.diag[e - iph * A.nel] + A.w0 * (double)iph);
        if (fabs(den) < 1e-8) den = den < 0.0 ? -1e-8 : 1e-8;
        return r / den;
    };
    double hs[NV], gw[NV], tt = 0.0, rr = 0.0;
#pragma unroll
    for (int i = 0; i < NV; i++) { hs[i] = 0.0; gw[i] = 0.0; }
    const int64_t n2 = n >> 1;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n2; p += (int64_t)gridDim.x * blockDim.x) {
        dbl2_t vv[NV], wv[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) { const int ic = i < nv ? i : 0; vv[i] = reinterpret_cast<const dbl2_t *>(A.V[ic])[p]; wv[i] = reinterpret_cast<const dbl2_t *>(A.W[ic])[p]; }
        dbl2_t r; r.x = 0.0; r.y = 0.0;
#pragma unroll
        for (int i = 0; i < NV; i++) { const double c = cs[i]; r.x += c * (wv[i].x - theta * vv[i].x); r.y += c * (wv[i].y - theta * vv[i].y); }
        dbl2_t o; o.x = finish(2 * p, r.x); o.y = finish(2 * p + 1, r.y);
        reinterpret_cast<dbl2_t *>(t)[p] = o;
        rr += r.x * r.x + r.y * r.y; tt += o.x * o.x + o.y * o.y;
#pragma unroll
        for (int i = 0; i < NV; i++) if (i < nv) { hs[i] += vv[i].x * o.x + vv[i].y * o.y; gw[i] += wv[i].x * o.x + wv[i].y * o.y; }
        if (ROT) {
#pragma unroll
            for (int j = 0; j < kDvdRotOut; j++) if (j < nout) {
                dbl2_t a, b; a.x = 0.0; a.y = 0.0; b.x = 0.0; b.y = 0.0;
#pragma unroll
                for (int i = 0; i < NV; i++) { const double c = cr[j * NV + i]; a.x += c * vv[i].x; a.y += c * vv[i].y; b.x += c * wv[i].x; b.y += c * wv[i].y; }
                reinterpret_cast<dbl2_t *>(A.V[j])[p] = a;
                reinterpret_cast<dbl2_t *>(A.W[j])[p] = b;
            }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t e = n - 1; double r = 0.0;
#pragma unroll
        for (int i = 0; i < NV; i++) if (i < nv) r += cs[i] * (A.W[i][e] - theta * A.V[i][e]);
        const double o = finish(e, r);
        t[e] = o; rr += r * r; tt += o * o;
#pragma unroll
        for (int i = 0; i < NV; i++) if (i < nv) { hs[i] += A.V[i][e] * o; gw[i] += A.W[i][e] * o; }
        if (ROT) {
            double ve[NV], we[NV];
#pragma unroll
            for (int i = 0; i < NV; i++) { ve[i] = A.V[i < nv ? i : 0][e]; we[i] = A.W[i < nv ? i : 0][e]; }
#pragma unroll
            for (int j = 0; j < kDvdRotOut; j++) if (j < nout) {
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int i = 0; i < NV; i++) { a += cr[j * NV + i] * ve[i]; b += cr[j * NV + i] * we[i]; }
                A.V[j][e] = a; A.W[j][e] = b;
            }
        }
    }
    const int64_t pb = (int64_t)blockIdx.x * (2 * nv + 2);
#pragma unroll
    for (int i = 0; i < NV; i++) { if (i < nv) { const double sI = block_sum(hs[i], red); if (threadIdx.x == 0) partial[pb + i] = sI; } }
#pragma unroll
    for (int i = 0; i < NV; i++) { if (i < nv) { const double sI = block_sum(gw[i], red); if (threadIdx.x == 0) partial[pb + nv + i] = sI; } }
    { const double sT = block_sum(tt, red); if (threadIdx.x == 0) partial[pb + 2 * nv] = sT; }
    { const double sR = block_sum(rr, red); if (threadIdx.x == 0) partial[pb + 2 * nv + 1] = sR; }
}

static inline void launch_dvd_step(hipStream_t st, int nb, const DvdArgs &D, double *t, int64_t n, double *partial, const double *C, int nout)
{
    const int m = D.nv;
#define C_(NV) do { if (C) hipLaunchKernelGGL((k_dvd_step<NV, true>), dim3(nb), dim3(256), 0, st, D, t, n, partial, C, nout); \
                    else hipLaunchKernelGGL((k_dvd_step<NV, false>), dim3(nb), dim3(256), 0, st, D, t, n, partial, C, nout); } while (0)
    switch (m) { case 1: C_(1); break; case 2: C_(2); break; case 3: C_(3); break; case 4: C_(4); break; case 5: C_(5); break; case 6: C_(6); break;
                 case 7: C_(7); break; case 8: C_(8); break; case 9: C_(9); break; case 10: C_(10); break; case 11: C_(11); break; default: C_(12); }
#undef C_
}

// Second half of the small-space Davidson step: v_new = (t - sum_i h_i V_i) * scale written straight into the next basis slot
// (the norm of the orthogonalised vector is known beforehand from |t|^2 - sum h_i^2 whenever that difference is not a
// cancellation, which is also the DGKS condition for one Gram-Schmidt pass being enough), |t - sum h_i V_i|^2 summed on the
// way so that the host can check both.  Replaces multiaxpy_norm in place + k_scale_copy (two vector reads and one write less).
template <int NV>
__global__ void __launch_bounds__(256) k_dvd_orth(BasisArgs B, const double *__restrict__ h, const double *__restrict__ t, double *__restrict__ out, double scale, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[4];
    __shared__ double cs[NV];
    constexpr int nv = NV;
    for (int i = threadIdx.x; i < NV; i += blockDim.x) cs[i] = i < nv ? h[i] : 0.0;
    __syncthreads();
    double ss = 0.0;
    const int64_t n2 = n >> 1;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n2; p += (int64_t)gridDim.x * blockDim.x) {
        dbl2_t vv[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) vv[i] = reinterpret_cast<const dbl2_t *>(B.V[i < nv ? i : 0])[p];
        dbl2_t x = reinterpret_cast<const dbl2_t *>(t)[p];
#pragma unroll
        for (int i = 0; i < NV; i++) { const double c = cs[i]; x.x -= c * vv[i].x; x.y -= c * vv[i].y; }
        ss += x.x * x.x + x.y * x.y;
        x.x *= scale; x.y *= scale;
        reinterpret_cast<dbl2_t *>(out)[p] = x;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double x = t[n - 1];
#pragma unroll
        for (int i = 0; i < NV; i++) if (i < nv) x -= cs[i] * B.V[i][n - 1];
        ss += x * x; out[n - 1] = x * scale;
    }
    const double r = block_sum(ss, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

static inline void launch_dvd_orth(hipStream_t st, int nb, const BasisArgs &B, const double *h, const double *t, double *out, double scale, int64_t n, double *partial)
{
#define C_(NV) hipLaunchKernelGGL((k_dvd_orth<NV>), dim3(nb), dim3(256), 0, st, B, h, t, out, scale, n, partial)
    switch (B.nv) { case 1: C_(1); break; case 2: C_(2); break; case 3: C_(3); break; case 4: C_(4); break; case 5: C_(5); break; case 6: C_(6); break;
                    case 7: C_(7); break; case 8: C_(8); break; case 9: C_(9); break; case 10: C_(10); break; case 11: C_(11); break; default: C_(12); }
#undef C_
}

__global__ void __launch_bounds__(256) k_scale_copy(double *__restrict__ y, const double *__restrict__ x, double a, int64_t n)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) y[e] = a * x[e];
}

// deterministic pseudo-random start vector in (0,1): counter-based hash of the GLOBAL index,
// so the same vector is produced for any rank count (the library seeds a fixed-seed RNG,
// SciFortran lanczos_eigh; the stream itself is compiler specific and not a parity quantity)
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) k_random_start(double *__restrict__ x, int64_t DimUp, int64_t Qdw, int64_t dw0, int64_t DimDw, int DimPh, uint64_t seed)
{
    const int64_t n = DimUp * Qdw * DimPh;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t iph = i / (DimUp * Qdw), r = i - iph * DimUp * Qdw;
        const int64_t g = r + dw0 * DimUp + iph * DimUp * DimDw;
        x[i] = ((double)(splitmix64((uint64_t)g + seed) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    }
}

}  // namespace edgpu
