// kernels.hpp -- gfx950 device kernels of the sector operator and the Lanczos vector updates.
//
// Layout in HBM (all fp64 vectors): v(iup, idw, iph) with iup fastest,
//   i = iup + idw*DimUp + iph*DimUp*Qdw        (ED_HAMILTONIAN_SPARSE_HxV.f90:417, 0-based here)
// A rank owns the dw columns [dw0, dw0+Qdw) (ED_HAMILTONIAN.f90:96-110).
//
// No MFMA anywhere: these are HBM-bound gather/stream kernels; what matters is coalescing
// along iup, LDS staging of the row that the up-hops gather from, and keeping the per-spin
// factor tables (KBs..MBs) cache resident.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace edgpu {

constexpr int kMaxTerms = 256;

// scalars of the Lanczos recurrence, resident in HBM so the loop needs no host round trip
enum { SC_SCUR = 0, SC_SPREV = 1, SC_BETA = 2, SC_ALPHA = 3, SC_TMP = 4, SC_ZK = 5, SC_K = 6,
       SC_NPEND = 7, SC_SV_SCUR = 8, SC_SV_BETA = 9, SC_PAIR = 10 /* and 11 */,
       SC_N_SCUR = 12, SC_N_SPREV = 13, SC_N_BETA = 14, SC_N_K = 15 /* self-reducing iteration: scales as the running kernel pair sees them */, SC_COUNT = 16 };      // 7-11: deferred-norm form of the loop for several ranks (one all-reduce per iteration)   // SC_K: iteration counter kept on the device (fused reduce+step kernels)

struct DiagArgs {  // bit-exact diagonal (stored/H_local.f90:13-71, stored/Orbs/H_local.f90:1-80)
    int32_t Norb, Nspin, Nbath, Nfoo, hfmode, total_ud, Ns_Ud;
    int32_t dims[2 * EDGPU_MAX_ORB];
    const int32_t *maps[2 * EDGPU_MAX_ORB];
    int64_t nloc, shift;  // local electronic rows, global offset of the first one
    double Uloc[EDGPU_MAX_ORB], Ust, Jh, xmu;
    double imp_up[EDGPU_MAX_ORB], imp_dw[EDGPU_MAX_ORB];
    const double *bd_up, *bd_dw;  // [Nfoo*Nbath]: bath_diag(1,a,k), bath_diag(Nspin,a,k), index (a-1)+Nfoo*(k-1)
    const int32_t *stride;        // [Norb*Nbath]: getBathStride(a,k), index (a-1)+Norb*(k-1)
};

// HIP's __dadd_rn/__dmul_rn are plain + and * which the compiler contracts to FMA; the
// reference's CPU path rounds after every multiply and every add, so pin the instructions.
__device__ __forceinline__ double mul_rn(double a, double b) { double r; asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double add_rn(double a, double b) { double r; asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double sub_rn(double a, double b) { double r; asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__global__ void __launch_bounds__(256) k_build_diag(DiagArgs A, double *__restrict__ d)
{
    int64_t il = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (il >= A.nloc) return;
    int64_t ig = il + A.shift;  // 0-based global electronic index
    uint32_t mu[EDGPU_MAX_ORB], md[EDGPU_MAX_ORB];
    {
        int64_t c = ig;
        for (int k = 0; k < A.Ns_Ud; k++) { mu[k] = (uint32_t)A.maps[k][c % A.dims[k]]; c /= A.dims[k]; }
        for (int k = 0; k < A.Ns_Ud; k++) { md[k] = (uint32_t)A.maps[k + A.Ns_Ud][c % A.dims[k + A.Ns_Ud]]; c /= A.dims[k + A.Ns_Ud]; }
    }
    auto nimp = [&](const uint32_t *m, int a) -> int { return A.total_ud ? (int)((m[0] >> (a - 1)) & 1u) : (int)(m[a - 1] & 1u); };
    auto nbath = [&](const uint32_t *m, int a, int kp) -> int {
        return A.total_ud ? (int)((m[0] >> (A.stride[(a - 1) + A.Norb * (kp - 1)] - 1)) & 1u) : (int)((m[a - 1] >> kp) & 1u);
    };
    // Non-fused IEEE ops in the reference's statement order => same bits as the CPU path.
    double h = 0.0;
    for (int a = 1; a <= A.Norb; a++) {
        int nu = nimp(mu, a), nd = nimp(md, a);
        h = add_rn(h, mul_rn(A.imp_up[a - 1], (double)nu));
        h = add_rn(h, mul_rn(A.imp_dw[a - 1], (double)nd));
        h = sub_rn(h, mul_rn(A.xmu, (double)(nu + nd)));
    }
    for (int a = 1; a <= A.Norb; a++)
        h = add_rn(h, mul_rn(mul_rn(A.Uloc[a - 1], (double)nimp(mu, a)), (double)nimp(md, a)));
    if (A.Norb > 1) {
        for (int a = 1; a <= A.Norb; a++)
            for (int b = a + 1; b <= A.Norb; b++)
                h = add_rn(h, mul_rn(A.Ust, (double)(nimp(mu, a) * nimp(md, b) + nimp(mu, b) * nimp(md, a))));
        double ustjh = sub_rn(A.Ust, A.Jh);
        for (int a = 1; a <= A.Norb; a++)
            for (int b = a + 1; b <= A.Norb; b++)
                h = add_rn(h, mul_rn(ustjh, (double)(nimp(mu, a) * nimp(mu, b) + nimp(md, a) * nimp(md, b))));
    }
    if (A.hfmode) {
        for (int a = 1; a <= A.Norb; a++) {
            double t1 = mul_rn(mul_rn(0.5, A.Uloc[a - 1]), (double)(nimp(mu, a) + nimp(md, a)));
            h = add_rn(sub_rn(h, t1), mul_rn(0.25, A.Uloc[a - 1]));
        }
        if (A.Norb > 1) {
            double ustjh = sub_rn(A.Ust, A.Jh);
            for (int a = 1; a <= A.Norb; a++)
                for (int b = a + 1; b <= A.Norb; b++) {
                    double n4 = (double)(nimp(mu, a) + nimp(md, a) + nimp(mu, b) + nimp(md, b));
                    h = add_rn(sub_rn(h, mul_rn(mul_rn(0.5, A.Ust), n4)), mul_rn(0.25, A.Ust));
                    h = add_rn(sub_rn(h, mul_rn(mul_rn(0.5, ustjh), n4)), mul_rn(0.25, ustjh));
                }
        }
    }
    for (int a = 1; a <= A.Nfoo; a++)
        for (int kp = 1; kp <= A.Nbath; kp++) {
            int ix = (a - 1) + A.Nfoo * (kp - 1);
            h = add_rn(h, mul_rn(A.bd_up[ix], (double)nbath(mu, a, kp)));
            h = add_rn(h, mul_rn(A.bd_dw[ix], (double)nbath(md, a, kp)));
        }
    d[il] = h;
}

// ---------------------------------------------------------------------------------------------
// block-wide deterministic sum (fixed shuffle tree, then fixed order over waves)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}
// the same sum through the DPP network (no LDS crossbar round trips): quad permutes, row shifts, row broadcasts; the total ends up
// in lane 63 and is broadcast from there.  Every lane of the wave must be active.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_mov_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double x)
{
    x += dpp_mov_f64<0xb1>(x);             // quad_perm [1,0,3,2]
    x += dpp_mov_f64<0x4e>(x);             // quad_perm [2,3,0,1]
    x += dpp_mov_f64<0x114>(x);            // row_shr 4   (lanes 4.. of a row; lanes 0-3 add 0: old value 0, bound_ctrl off -> keeps old = 0)
    x += dpp_mov_f64<0x118>(x);            // row_shr 8
    x += dpp_mov_f64<0x142, 0xa>(x);       // row_bcast 15 -> rows 1 and 3
    x += dpp_mov_f64<0x143, 0xc>(x);       // row_bcast 31 -> rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double block_sum(double x, double *scratch /* >= blockDim/64 doubles */)
{
    x = wave_sum(x);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = x;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < nw; i++) r += scratch[i];
    return r;  // valid on thread 0
}

// ---------------------------------------------------------------------------------------------
// The fused two-factor operator (ed_total_ud=T):
//   Hv(iup,idw) = d*v + sum_k Hdw[idw,k] v(iup,jdw_k) + sum_k Hup[iup,k] v(jup_k,idw) [+ph] [+Jx/Jp]
// replaces the five sweeps of spMatVec_main (ED_HAMILTONIAN_SPARSE_HxV.f90:391-485) and the
// scatter loops of directMatVec_main (ED_HAMILTONIAN_DIRECT_HxV.f90:21-95) by one gather pass.
// One workgroup per (idw, iph) row of DimUp contiguous doubles; the row is staged in LDS
// because every up-hop gathers from it; dw-hops read whole neighbour rows, coalesced in iup.
// ---------------------------------------------------------------------------------------------
struct Kron2Args {
    int32_t DimUp, DimDw, DimPh, Norb;
    int64_t Qdw, dw0;
    const int32_t *upc; const double *upv; int32_t upw;  // ELL, [upw][DimUp]
    const int32_t *dwc; const double *dwv; int32_t dww;  // ELL, [DimDw][dww]
    const double *diag;                                   // [Qdw*DimUp] (KRON) or nullptr (DIRECT)
    const double *Eup, *Edw, *cross;                     // DIRECT: per-spin tables + impurity coupling
    const int32_t *mapup, *mapdw, *rankup, *rankdw;
    int32_t impmask, nI;
    double econst, w0, gsum, Jx, Jp;
    const double *gup, *gdw;
    const double *v, *vfull;  // local shard / gathered vector (same pointer on one GPU)
    double *out;
    const double *uprev; const double *scal; double *partial;  // Lanczos epilogue
    int32_t rsplit;   // >1 (rows too long for LDS, local-only pass): `rsplit` workgroups of one XCD share a row
    // KF_SR (self-reducing Lanczos iteration, launch-bound sectors): the beta^2 partial sums the previous axpy pass left behind
    const double *sr_part; int32_t sr_n; double *scal_w; double *betas_w;
};

enum { KF_DIRECT = 1, KF_LDS = 2, KF_EPI = 4, KF_PH = 8, KF_ND = 16, KF_NOLOCAL = 32, KF_NODW = 64, KF_SR = 128 };

// Self-reducing Lanczos iteration (round 4).  A launch-bound sector spends an iteration in four launches: product (alpha partial sums),
// a one-workgroup reduction + scalar step, the axpy/norm pass (beta^2 partial sums), another reduction + step -- ~4.5 us each on the
// device side whatever they do.  Here EVERY workgroup of the consuming kernel sums the partials of the producing one itself (a few
// hundred numbers, the same fixed order everywhere, so every workgroup holds the same bits) and applies the scalar step in registers;
// workgroup 0 also writes the coefficient and the scales for whoever reads them later.  Scales come in two sets: the committed ones
// (SC_SCUR, SC_SPREV, SC_BETA, SC_K: what the host-side loops and the other kernels read) and the ones of the iteration in flight
// (SC_N_*): the product kernel reads the committed set and writes the in-flight one, the axpy kernel reads the in-flight set and
// commits it -- no workgroup reads a slot that another workgroup of the same launch writes.  Two launches per iteration.
__device__ __forceinline__ double sr_sum(const double *__restrict__ part, int n, double *red /* >= 17 doubles */)
{
    double x = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) x += part[i];
    x = wave_sum(x);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = x;
    __syncthreads();
    if (threadIdx.x == 0) { double r = 0.0; for (int i = 0; i < nw; i++) r += red[i]; red[16] = r; }
    __syncthreads();
    return red[16];
}

template <int F>
__global__ void __launch_bounds__(1024) k_kron2(Kron2Args A)
{
    extern __shared__ double lds[];
    const int DimUp = A.DimUp;
    double *red = lds + ((F & KF_LDS) ? DimUp : 0);
    // rsplit > 1: blocks b, b+8, b+16, ... (same blockIdx % 8, hence the same XCD under round-robin dispatch)
    // take consecutive parts of one row, so that row -- which every up hop gathers from -- is fetched into
    // one L2 once and at most 8 rows per XCD are live.  Placement only affects speed.
    int64_t row = blockIdx.x; int part = 0;
    if (A.rsplit > 1) {
        const int64_t g = blockIdx.x / (8 * A.rsplit); const int w = (int)(blockIdx.x - g * 8 * A.rsplit);
        row = g * 8 + (w & 7); part = w >> 3;
    }
    if (row >= A.Qdw * A.DimPh) return;                                // never true for a well-formed launch
    const int iph = (int)(row / A.Qdw);
    const int64_t idwl = row - (int64_t)iph * A.Qdw, idw = A.dw0 + idwl;
    const int64_t rbase = ((int64_t)iph * A.Qdw + idwl) * DimUp;       // local row base
    const int64_t fbase = (int64_t)iph * A.DimDw * DimUp;              // phonon block base in vfull
    const double *vrow = A.v + rbase;
    if (F & KF_LDS) {
        for (int i = threadIdx.x; i < DimUp; i += blockDim.x) lds[i] = vrow[i];
        __syncthreads();
    }
    const int dww = (F & KF_NODW) ? 0 : A.dww;
    const int32_t *dwc = A.dwc + idw * A.dww;
    const double *dwv = A.dwv + idw * A.dww;
    uint32_t mdw = 0; double edw = 0.0;
    if (F & (KF_DIRECT | KF_PH | KF_ND)) mdw = (uint32_t)A.mapdw[idw];
    if (F & KF_DIRECT) edw = A.Edw[idw] + A.econst;
    double s_cur = 1.0, c_prev = 0.0, asum = 0.0;
    if (F & KF_EPI) { s_cur = A.scal[SC_SCUR]; c_prev = A.scal[SC_BETA] * A.scal[SC_SPREV]; }
    if ((F & KF_EPI) && (F & KF_SR)) {      // the beta step of the previous iteration, taken here by every workgroup (see sr_sum)
        double scur = A.scal[SC_SCUR], sprev = A.scal[SC_SPREV], beta = A.scal[SC_BETA], K = A.scal[SC_K];
        if (A.sr_n > 0) { const double b = sqrt(sr_sum(A.sr_part, A.sr_n, red)); beta = b; sprev = scur; scur = b > 0.0 ? 1.0 / b : 0.0; K += 1.0; }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            A.scal_w[SC_N_SCUR] = scur; A.scal_w[SC_N_SPREV] = sprev; A.scal_w[SC_N_BETA] = beta; A.scal_w[SC_N_K] = K;
            if (A.sr_n > 0) A.betas_w[(int)K] = beta;
        }
        s_cur = scur; c_prev = beta * sprev;
    }

    int iup_lo = 0, iup_hi = DimUp;
    if (A.rsplit > 1) { const int chunk = (DimUp + A.rsplit - 1) / A.rsplit; iup_lo = part * chunk; iup_hi = iup_lo + chunk < DimUp ? iup_lo + chunk : DimUp; }
    for (int iup = iup_lo + threadIdx.x; iup < iup_hi; iup += blockDim.x) {
        const double x = (F & KF_LDS) ? lds[iup] : vrow[iup];
        double acc = 0.0;
        uint32_t mup = 0;
        if (F & (KF_DIRECT | KF_PH | KF_ND)) mup = (uint32_t)A.mapup[iup];
        if (!(F & KF_NOLOCAL)) {
            double dval;
            if (F & KF_DIRECT) dval = edw + A.Eup[iup] + A.cross[(mup & A.impmask) * A.nI + (mdw & A.impmask)];
            else dval = A.diag[idwl * DimUp + iup];
            if (F & KF_PH) dval += A.w0 * iph;
            acc = dval * x;
            // up hops: gather inside the staged row
            for (int k = 0; k < A.upw; k++) {
                int32_t j = A.upc[(int64_t)k * DimUp + iup];
                double t = A.upv[(int64_t)k * DimUp + iup];
                acc += t * ((F & KF_LDS) ? lds[j] : vrow[j]);
            }
            if (F & KF_PH) {  // stored/H_e_ph.f90:12-39 : g(n-1) (x) (b + b^+)
                double g = A.gup[mup & A.impmask] + A.gdw[mdw & A.impmask] - A.gsum;
                const int64_t pstride = A.Qdw * DimUp;
                if (iph + 1 < A.DimPh) acc += g * sqrt((double)(iph + 1)) * A.v[rbase + pstride + iup];
                if (iph > 0) acc += g * sqrt((double)iph) * A.v[rbase - pstride + iup];
            }
        }
        // dw hops: neighbour rows, coalesced in iup
        for (int k = 0; k < dww; k++) acc += dwv[k] * A.vfull[fbase + (int64_t)dwc[k] * DimUp + iup];
        if ((F & KF_ND) && !(F & KF_NODW)) {  // stored/H_non_local.f90:21-83 evaluated on the fly
            for (int a = 0; a < A.Norb; a++)
                for (int b = 0; b < A.Norb; b++) {
                    const uint32_t ba = 1u << a, bb = 1u << b;
                    if (A.Jx != 0.0 && a != b && (mup & bb) && (mdw & ba) && !(mdw & bb) && !(mup & ba)) {
                        uint32_t k1 = mdw & ~ba, k2 = k1 | bb, k3 = mup & ~bb, k4 = k3 | ba;
                        int sg = __popc(mdw & (ba - 1)) + __popc(k1 & (bb - 1)) + __popc(mup & (bb - 1)) + __popc(k3 & (ba - 1));
                        double t = (sg & 1) ? -A.Jx : A.Jx;
                        acc += t * A.vfull[fbase + (int64_t)A.rankdw[k2] * DimUp + A.rankup[k4]];
                    }
                    if (A.Jp != 0.0 && (mup & bb) && (mdw & bb) && !(mdw & ba) && !(mup & ba)) {
                        uint32_t k1 = mdw & ~bb, k2 = k1 | ba, k3 = mup & ~bb, k4 = k3 | ba;
                        int sg = __popc(mdw & (bb - 1)) + __popc(k1 & (ba - 1)) + __popc(mup & (bb - 1)) + __popc(k3 & (ba - 1));
                        double t = (sg & 1) ? -A.Jp : A.Jp;
                        acc += t * A.vfull[fbase + (int64_t)A.rankdw[k2] * DimUp + A.rankup[k4]];
                    }
                }
        }
        if (F & KF_EPI) {
            double w;
            if (F & KF_NOLOCAL) w = A.out[rbase + iup] + s_cur * acc;        // second half of a split product
            else w = s_cur * acc - c_prev * A.uprev[rbase + iup];
            A.out[rbase + iup] = w;
            if (!(F & KF_NODW)) asum += (s_cur * x) * w;
        } else {
            if (F & KF_NOLOCAL) A.out[rbase + iup] += acc; else A.out[rbase + iup] = acc;
        }
    }
    if ((F & KF_EPI) && !(F & KF_NODW)) {
        double r = block_sum(asum, red);
        if (threadIdx.x == 0) A.partial[blockIdx.x] = r;
    }
}

// ---------------------------------------------------------------------------------------------
typedef double dbl2v __attribute__((ext_vector_type(2)));
// Local pass (diagonal + up hops) of the split product, persistent form: one 1024-thread
// workgroup per CU walks rows b, b+grid, ...  The row is staged in LDS for the up-hop gathers;
// the NEXT row is already on its way into registers while the current one is computed, so the
// HBM stream never waits for the barrier.  H_up is read in a packed image (16-bit column +
// 8-bit term code per hop: 3 B instead of 12 B per entry, amplitudes from a tiny LDS table)
// because every workgroup re-reads the whole factor for every row it owns.
// ---------------------------------------------------------------------------------------------
struct UpArgs {
    int32_t DimUp, DimPh; int64_t Qdw, dw0, nrows;
    const uint32_t *pcols, *pcodes; int32_t upw;      // packed ELL: [ceil(upw/2)][DimUp], [ceil(upw/4)][DimUp]
    const double *tval; int32_t nterms;
    const double *diag, *Eup, *Edw, *cross; const int32_t *mapup, *mapdw; int32_t impmask, nI; double econst;
    const double *v; double *out; const double *uprev; const double *scal;
    int32_t ld;              // row pitch of v / out / uprev in doubles (0: DimUp); the stored diagonal keeps pitch DimUp
    const double *crowtab;   // DIRECT: [idw][nI] row constants (k_crow_table)
    const uint4 *pelem16;    // element-major image, 12 dwords per configuration (8 of 16-bit columns, 4 of 8-bit codes); nullptr: the slot-major image above
};

template <bool DIRECT, bool EPI, int NT>
__global__ void __launch_bounds__(NT) k_uppass(UpArgs A)
{
    extern __shared__ double lds[];
    constexpr int MAXQ = 13312 / NT;               // DimUp <= 13312 (Ns <= 16 at half filling)
    constexpr int MAXC = 4;                        // <= 16 hops per up configuration
    const int DimUp = A.DimUp, tid = threadIdx.x;
    const int64_t ld = A.ld ? A.ld : DimUp;
    double *tv = lds + ((DimUp + 1) & ~1);
    double *crow = tv + 128;                                      // DIRECT: see k_uppass2
    uint8_t *imp8 = reinterpret_cast<uint8_t *>(crow + 32);
    for (int t = tid; t < A.nterms; t += NT) tv[t] = A.tval[t];
    if (DIRECT) for (int t = tid; t < DimUp; t += NT) imp8[t] = (uint8_t)((uint32_t)A.mapup[t] & A.impmask);
    const int nq = (DimUp + NT - 1) / NT;
    const int ncw = (A.upw + 1) >> 1, ndw = (A.upw + 3) >> 2;
    // Two software pipelines.  Over rows: the NEXT row of v is requested into registers while
    // the current row is computed out of LDS.  Inside a row: the operands of element q+1 (stored
    // diagonal, u_prev, packed hops) are requested before element q is computed.
    double xr[MAXQ];
    int64_t row = blockIdx.x;
    double crn = 0.0;
    if (DIRECT) crn = A.crowtab[(A.dw0 + (row < A.nrows ? row : 0) % A.Qdw) * A.nI + (tid < A.nI ? tid : 0)];
    if (row < A.nrows) {
#pragma unroll
        for (int q = 0; q < MAXQ; q++) { const int i = tid + q * NT; xr[q] = (q < nq && i < DimUp) ? A.v[row * ld + i] : 0.0; }
    }
    double s_cur = 1.0, c_prev = 0.0;
    if (EPI) { s_cur = A.scal[SC_SCUR]; c_prev = A.scal[SC_BETA] * A.scal[SC_SPREV]; }
    struct Ops { double d, u; uint32_t codes[MAXC], cols[2 * MAXC]; };
    while (row < A.nrows) {
        const int64_t idwl = row % A.Qdw;
        auto load_ops = [&](int i, Ops &o) {
            o.d = DIRECT ? A.Eup[i] : A.diag[idwl * DimUp + i];
            if (EPI) o.u = A.uprev[row * ld + i];
            if (A.pelem16) {      // three 16-byte loads: columns 0-7, 8-15, codes (padding slots: the row's own position, code 127)
                const uint4 *pe = A.pelem16 + 3 * (int64_t)i;
                const uint4 a = pe[0], b = pe[1], c4 = pe[2];
                o.cols[0] = a.x; o.cols[1] = a.y; o.cols[2] = a.z; o.cols[3] = a.w; o.cols[4] = b.x; o.cols[5] = b.y; o.cols[6] = b.z; o.cols[7] = b.w;
                o.codes[0] = c4.x; o.codes[1] = c4.y; o.codes[2] = c4.z; o.codes[3] = c4.w;
            } else {
#pragma unroll
            for (int c = 0; c < MAXC; c++) if (c < ndw) o.codes[c] = A.pcodes[(int64_t)c * DimUp + i];
#pragma unroll
            for (int c = 0; c < 2 * MAXC; c++) o.cols[c] = c < ncw ? A.pcols[(int64_t)c * DimUp + i] : ((uint32_t)i | ((uint32_t)i << 16));      // slots beyond the image (odd widths): the row's own position, like any padding slot -- never an uninitialised LDS index
            }
        };
        Ops cur, nxt;
        if (tid < DimUp) load_ops(tid, cur);
#pragma unroll
        for (int q = 0; q < MAXQ; q++) { const int i = tid + q * NT; if (q < nq && i < DimUp) lds[i] = xr[q]; }
        if (DIRECT && tid < A.nI) crow[tid] = crn;
        __syncthreads();
        const int64_t next = row + gridDim.x;
        if (DIRECT) crn = A.crowtab[(A.dw0 + (next < A.nrows ? next : row) % A.Qdw) * A.nI + (tid < A.nI ? tid : 0)];
        if (next < A.nrows) {
#pragma unroll
            for (int q = 0; q < MAXQ; q++) { const int i = tid + q * NT; if (q < nq && i < DimUp) xr[q] = A.v[next * ld + i]; }
        }
        for (int i = tid; i < DimUp; i += NT) {
            if (i + NT < DimUp) load_ops(i + NT, nxt);
            const double x = lds[i];
            const double dval = DIRECT ? cur.d + crow[imp8[i]] : cur.d;
            double acc = dval * x;
#pragma unroll
            for (int c = 0; c < MAXC; c++)
                if (c < ndw && __any(cur.codes[c] != 0x7f7f7f7fu)) {   // ragged factors: a group of 4 slots that is padding in all 64 lanes is skipped
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const uint32_t code = (cur.codes[c] >> (8 * e)) & 0xffu;
                        const uint32_t col = (cur.cols[2 * c + (e >> 1)] >> (16 * (e & 1))) & 0xffffu;
                        double t = tv[code & 0x7fu];
                        t = (code & 0x80u) ? -t : t;
                        acc += t * lds[col];
                    }
                }
            if (EPI) A.out[row * ld + i] = s_cur * acc - c_prev * cur.u;
            else A.out[row * ld + i] = acc;
            cur = nxt;
        }
        __syncthreads();
        row = next;
    }
}

// Pair form of the persistent local pass (needs DimUp even and <= 8 hops per up configuration):
// a lane owns two consecutive iup, so the row, the stored diagonal, u_prev and the result move as
// 16-byte accesses and the packed H_up image of the pair (2 x 24 B, element-major) as three 16-byte
// loads: about a third of the vector-memory instructions of k_uppass.
struct UpArgs2 {
    int32_t DimUp, DimPh; int64_t Qdw, dw0, nrows;
    const uint4 *packed;     // [iup/2][3] uint4: element e -> dwords {c01,c23,c45,c67,codes0123,codes4567}
    const double *tval; int32_t nterms;
    const double *diag, *Eup, *Edw, *cross; const int32_t *mapup, *mapdw; int32_t impmask, nI; double econst;
    const double *v; double *out; const double *uprev; const double *scal;
    int32_t ld;              // row pitch of v / out / uprev in doubles (0: DimUp)
    const double *crowtab;   // DIRECT: [idw][nI] row constants (k_crow_table)
    int32_t dtrue;           // true DimUp when DimUp above is the row length rounded up to even (odd rows on pitched vectors: the last pair ends in the zero padding column); 0: same
};

// NH: hop slots per up configuration.  8: the 6-dword element image (3 x 16 bytes per pair); 10: the 8-dword image pelem8 (4 x 16 bytes
// per pair) -- sectors one or two electrons off half filling (9-10 up hops: the Green's-function target sectors), which otherwise
// fall to the one-configuration-per-lane kernel with its 12-dword image.
template <bool DIRECT, bool EPI, int NT, int NH = 8>
__global__ void __launch_bounds__(NT) k_uppass2(UpArgs2 A)
{
    extern __shared__ double lds[];
    constexpr int MAXQ = (13312 / 2 + NT - 1) / NT;      // pairs per lane
    const int DimUp = A.DimUp, npair = DimUp >> 1, tid = threadIdx.x;
    const int64_t ld = A.ld ? A.ld : DimUp;
    const int dtrue = A.dtrue ? A.dtrue : DimUp;      // pitch of the stored diagonal, length of mapup
    double *tv = lds + DimUp;
    // DIRECT: the diagonal is Eup(iup) [an operand like the stored diagonal] + crow(up impurity word), with crow = Edw + econst +
    // cross(., dw impurity word) of the row taken from the per-product table a row ahead (k_crow_table) and the impurity word of
    // every up configuration staged in LDS once per launch -- the first form loaded mapup, Eup and cross one after the other
    // inside every step (2.37 ms per product at Ns=16 against 2.17 with the stored diagonal)
    double *crow = tv + 128;
    uint8_t *imp8 = reinterpret_cast<uint8_t *>(crow + 32);
    for (int t = tid; t < A.nterms; t += NT) tv[t] = A.tval[t];
    if (DIRECT) for (int t = tid; t < DimUp; t += NT) imp8[t] = t < dtrue ? (uint8_t)((uint32_t)A.mapup[t] & A.impmask) : (uint8_t)0;
    const int nq = (npair + NT - 1) / NT;
    dbl2v xr[MAXQ];
    int64_t row = blockIdx.x;
    double crn = 0.0;
    if (DIRECT) crn = A.crowtab[(A.dw0 + (row < A.nrows ? row : 0) % A.Qdw) * A.nI + (tid < A.nI ? tid : 0)];
    if (row < A.nrows) {
#pragma unroll
        for (int q = 0; q < MAXQ; q++) { const int p = tid + q * NT; if (q < nq && p < npair) xr[q] = *reinterpret_cast<const dbl2v *>(A.v + row * ld + 2 * p); }
    }
    double s_cur = 1.0, c_prev = 0.0;
    if (EPI) { s_cur = A.scal[SC_SCUR]; c_prev = A.scal[SC_BETA] * A.scal[SC_SPREV]; }
    struct Ops { dbl2v d, u; uint4 w0, w1, w2, w3; };
    while (row < A.nrows) {
        const int64_t idwl = row % A.Qdw;
        auto load_ops = [&](int p, Ops &o) {
            if (!DIRECT) { if (A.dtrue) { const double *dp = A.diag + idwl * dtrue + 2 * p; o.d.x = dp[0]; o.d.y = dp[1]; }      // odd rows: the dense diagonal is not 16-byte aligned row by row
                           else o.d = *reinterpret_cast<const dbl2v *>(A.diag + idwl * dtrue + 2 * p); }
            else o.d = *reinterpret_cast<const dbl2v *>(A.Eup + 2 * p);
            if (EPI) o.u = *reinterpret_cast<const dbl2v *>(A.uprev + row * ld + 2 * p);
            if (NH == 8) { o.w0 = A.packed[3 * p]; o.w1 = A.packed[3 * p + 1]; o.w2 = A.packed[3 * p + 2]; }
            else { o.w0 = A.packed[4 * p]; o.w1 = A.packed[4 * p + 1]; o.w2 = A.packed[4 * p + 2]; o.w3 = A.packed[4 * p + 3]; }
        };
        Ops cur, nxt;
        if (tid < npair) load_ops(tid, cur);
#pragma unroll
        for (int q = 0; q < MAXQ; q++) { const int p = tid + q * NT; if (q < nq && p < npair) *reinterpret_cast<dbl2v *>(lds + 2 * p) = xr[q]; }
        if (DIRECT && tid < A.nI) crow[tid] = crn;
        __syncthreads();
        const int64_t next = row + gridDim.x;
        if (DIRECT) crn = A.crowtab[(A.dw0 + (next < A.nrows ? next : row) % A.Qdw) * A.nI + (tid < A.nI ? tid : 0)];
        if (next < A.nrows) {
#pragma unroll
            for (int q = 0; q < MAXQ; q++) { const int p = tid + q * NT; if (q < nq && p < npair) xr[q] = *reinterpret_cast<const dbl2v *>(A.v + next * ld + 2 * p); }
        }
        for (int p = tid; p < npair; p += NT) {
            if (p + NT < npair) load_ops(p + NT, nxt);
            const dbl2v x = *reinterpret_cast<const dbl2v *>(lds + 2 * p);
            dbl2v dval;
            if (DIRECT) {
                const uint32_t im = *reinterpret_cast<const uint16_t *>(imp8 + 2 * p);
                dval.x = cur.d.x + crow[im & 0xffu];
                dval.y = cur.d.y + crow[im >> 8];
            } else dval = cur.d;
            // NH = 8: element 0 of the pair: dwords w0.x w0.y w0.z w0.w (cols) w1.x w1.y (codes); element 1: w1.z w1.w w2.x w2.y, w2.z w2.w
            // NH = 10: element 0 = (w0, w1), element 1 = (w2, w3): five dwords of columns, three of codes each
            const uint32_t c0[5] = {cur.w0.x, cur.w0.y, cur.w0.z, cur.w0.w, NH == 8 ? 0u : cur.w1.x}, k0[3] = {NH == 8 ? cur.w1.x : cur.w1.y, NH == 8 ? cur.w1.y : cur.w1.z, NH == 8 ? 0u : cur.w1.w};
            const uint32_t c1[5] = {NH == 8 ? cur.w1.z : cur.w2.x, NH == 8 ? cur.w1.w : cur.w2.y, NH == 8 ? cur.w2.x : cur.w2.z, NH == 8 ? cur.w2.y : cur.w2.w, NH == 8 ? 0u : cur.w3.x},
                           k1[3] = {NH == 8 ? cur.w2.z : cur.w3.y, NH == 8 ? cur.w2.w : cur.w3.z, NH == 8 ? 0u : cur.w3.w};
            double a0 = dval.x * x.x, a1 = dval.y * x.y;
#pragma unroll
            for (int e = 0; e < NH; e++) {
                const uint32_t code0 = (k0[e >> 2] >> (8 * (e & 3))) & 0xffu, col0 = (c0[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                const uint32_t code1 = (k1[e >> 2] >> (8 * (e & 3))) & 0xffu, col1 = (c1[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                double t0 = tv[code0 & 0x7fu], t1 = tv[code1 & 0x7fu];
                t0 = (code0 & 0x80u) ? -t0 : t0; t1 = (code1 & 0x80u) ? -t1 : t1;
                a0 += t0 * lds[col0]; a1 += t1 * lds[col1];
            }
            dbl2v o;
            if (EPI) { o.x = s_cur * a0 - c_prev * cur.u.x; o.y = s_cur * a1 - c_prev * cur.u.y; }
            else { o.x = a0; o.y = a1; }
            *reinterpret_cast<dbl2v *>(A.out + row * ld + 2 * p) = o;
            cur = nxt;
        }
        __syncthreads();
        row = next;
    }
}

// Local pass of the FUSED Lanczos iteration (lz_iter_fused in lanczos.hpp).  The plain loop runs three passes over the
// vectors per iteration: local pass, dw pass, and the axpy+norm pass w -= alpha v, beta^2 = w.w (24 B per element), because
// alpha needs the whole product.  That last pass has the same operands as the NEXT iteration's local pass -- the residual
// r_{k-1} and u_{k-1} -- so it is folded into it: this kernel forms u_k = r_{k-1} - (alpha_{k-1} s_{k-1}) u_{k-1} row by
// row on its way into LDS (writing it back in place of r_{k-1} and summing |u_k|^2 = beta_{k-1}^2), then computes the
// UNSCALED local part T = d*u_k + up hops of u_k: the scale 1/beta_{k-1} is only known after this kernel's own
// reduction and is applied by the dw pass (k_dwpass2 with uprev2), which also subtracts beta_{k-1} v_{k-1} and takes the
// alpha dot.  72 B per element and iteration instead of 88, one launch fewer.  Same arithmetic as the three-pass loop up
// to the association (s*(T_loc + T_dw) - c*u_prev  vs  (s*T_loc - c*u_prev) + s*T_dw).
struct UpAxArgs {
    int32_t DimUp, DimPh; int64_t Qdw, dw0, nrows;
    const uint4 *packed; const double *tval; int32_t nterms;
    const double *diag, *Eup, *Edw, *cross; const int32_t *mapup, *mapdw; int32_t impmask, nI; double econst;
    double *r;               // in: r_{k-1}; out: u_k (in place)
    const double *vprev;     // u_{k-1}
    double *t_out;           // T = local part of H u_k, unscaled
    const double *scal; double *partial;
    double *acc; const double *zvec;   // optional eigenvector accumulation acc += zvec[K] * u_k (K = iteration counter in scal)
    int32_t ld;              // row pitch of r / vprev / t_out / acc in doubles (0: DimUp); the stored diagonal keeps pitch DimUp
    const uint32_t *pcols, *pcodes; int32_t upw;      // k_uppass_ax (one configuration per lane, up to 16 hops): packed ELL image as k_uppass
    const double *crowtab;   // DIRECT: [idw][nI] row constants (k_crow_table)
    int32_t dtrue;           // pair forms: true DimUp when DimUp is the odd row length rounded up to even (see UpArgs2); 0: same
    const uint4 *pelem16;    // k_uppass_ax: element-major image of 12 dwords per configuration; nullptr: the slot-major image (pcols / pcodes)
};

template <bool DIRECT, bool ACC, int NT, int NH = 8>      // NH: hop slots per up configuration, as k_uppass2
__global__ void __launch_bounds__(NT) k_uppass2_ax(UpAxArgs A)
{
    extern __shared__ double lds[];
    constexpr int MAXQ = (13312 / 2 + NT - 1) / NT;      // pairs per lane
    const int DimUp = A.DimUp, npair = DimUp >> 1, tid = threadIdx.x;
    const int64_t ld = A.ld ? A.ld : DimUp;
    const int dtrue = A.dtrue ? A.dtrue : DimUp;      // pitch of the stored diagonal, length of mapup
    double *tv = lds + DimUp;
    double *red = tv + 128;
    double *crow = red + 32;                             // DIRECT: Edw + const + cross term of this row, per up impurity word
    uint8_t *imp8 = reinterpret_cast<uint8_t *>(crow + 32);      // DIRECT: impurity word of every up configuration (the index into crow), staged once per launch
    for (int t = tid; t < A.nterms; t += NT) tv[t] = A.tval[t];
    if (DIRECT) for (int t = tid; t < DimUp; t += NT) imp8[t] = t < dtrue ? (uint8_t)((uint32_t)A.mapup[t] & A.impmask) : (uint8_t)0;
    const int nq = (npair + NT - 1) / NT;
    const double c_ax = A.scal[SC_ALPHA] * A.scal[SC_SCUR];          // alpha_{k-1} * s_{k-1}  (0 at k = 0)
    const double zk = ACC ? A.zvec[(int)A.scal[SC_K]] : 0.0;
    double bsum = 0.0;
    // u = r - c*vprev for one pair of row rw: written back over r, summed into the norm, kept by the caller for LDS
    auto combine = [&](int64_t rw, int p, const dbl2v &r, const dbl2v &v) -> dbl2v {
        const dbl2v u = r - c_ax * v;
        *reinterpret_cast<dbl2v *>(A.r + rw * ld + 2 * p) = u;
        bsum += u.x * u.x + u.y * u.y;
        if (ACC) { dbl2v *pa = reinterpret_cast<dbl2v *>(A.acc + rw * ld + 2 * p); *pa = *pa + zk * u; }
        return u;
    };
    dbl2v xu[MAXQ];                                      // u of the row that enters LDS next
    double crn = 0.0;
    if (DIRECT) crn = A.crowtab[(A.dw0 + (blockIdx.x < A.nrows ? blockIdx.x : 0) % A.Qdw) * A.nI + (tid < A.nI ? tid : 0)];
    const int pmax = npair > 0 ? npair - 1 : 0;
    int64_t row = blockIdx.x;
    if (row < A.nrows) {
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
            const int p = tid + q * NT;
            if (q < nq && p < npair)
                xu[q] = combine(row, p, *reinterpret_cast<const dbl2v *>(A.r + row * ld + 2 * p), *reinterpret_cast<const dbl2v *>(A.vprev + row * ld + 2 * p));
        }
    }
    // The operands of step q+1 of the hop loop are loaded during step q: the H_up image and diagonal of pair q+1 of this row
    // AND pair q+1 of the NEXT row's r and vprev, which step q+1 turns into u (registers, one pair per step) -- the next row
    // is complete when this one is, with 8 staging registers instead of a second row-sized register array.
    struct Ops { dbl2v d, nr, nv; uint4 w0, w1, w2, w3; };      // d: stored diagonal (KRON) or Eup (DIRECT); w3: NH = 10 only
    while (row < A.nrows) {
        const int64_t idwl = row % A.Qdw, idw = A.dw0 + idwl;
        const int64_t next = row + gridDim.x; const bool more = next < A.nrows;
        // No branch around an operand load (lanes beyond the row load a clamped pair, the last row loads itself again): a wait for
        // an older load can only carry the smallest count over all paths, and one path that skips the prefetch makes every such
        // wait vmcnt(0) (see k_uppass_halo).
        // (The on-the-fly form keeps the guarded loads: its extra operands push the straight-line form from 16 to 26 spilled
        // registers and the pass from 1.27 to 1.35-1.43 ms at Ns=16; the stored form gains nothing either way, 1.21-1.24 ms.)
        constexpr bool STRAIGHT = !DIRECT;
        const int64_t nrow = (STRAIGHT && !more) ? row : next;
        auto load_ops = [&](int p, Ops &o) {
            if (STRAIGHT) p = p < npair ? p : pmax;
            if (!DIRECT) { if (A.dtrue) { const double *dp = A.diag + idwl * dtrue + 2 * p; o.d.x = dp[0]; o.d.y = dp[1]; }      // odd rows: the dense diagonal is not 16-byte aligned row by row
                           else o.d = *reinterpret_cast<const dbl2v *>(A.diag + idwl * dtrue + 2 * p); }
            else o.d = *reinterpret_cast<const dbl2v *>(A.Eup + 2 * p);
            if (NH == 8) { o.w0 = A.packed[3 * p]; o.w1 = A.packed[3 * p + 1]; o.w2 = A.packed[3 * p + 2]; }
            else { o.w0 = A.packed[4 * p]; o.w1 = A.packed[4 * p + 1]; o.w2 = A.packed[4 * p + 2]; o.w3 = A.packed[4 * p + 3]; }
            if (STRAIGHT || more) { o.nr = *reinterpret_cast<const dbl2v *>(A.r + nrow * ld + 2 * p); o.nv = *reinterpret_cast<const dbl2v *>(A.vprev + nrow * ld + 2 * p); }
        };
        Ops cur, nxt;
        if (STRAIGHT || tid < npair) load_ops(tid, cur);
#pragma unroll
        for (int q = 0; q < MAXQ; q++) { const int p = tid + q * NT; if (q < nq && p < npair) *reinterpret_cast<dbl2v *>(lds + 2 * p) = xu[q]; }
        if (DIRECT && tid < A.nI) crow[tid] = crn;
        __syncthreads();
        if (DIRECT) crn = A.crowtab[(A.dw0 + (more ? next : row) % A.Qdw) * A.nI + (tid < A.nI ? tid : 0)];      // row constant of the next row (k_crow_table), a row ahead
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
            if (q >= nq) break;
            const int p = tid + q * NT;
            if (STRAIGHT) load_ops(p + NT, nxt);
            if (p < npair) {
                if (!STRAIGHT && p + NT < npair) load_ops(p + NT, nxt);
                if (more) xu[q] = combine(next, p, cur.nr, cur.nv);      // xu[q] of THIS row is in LDS already
                const dbl2v x = *reinterpret_cast<const dbl2v *>(lds + 2 * p);
                dbl2v dval;
                if (DIRECT) {
                    const uint32_t im = *reinterpret_cast<const uint16_t *>(imp8 + 2 * p);
                    dval.x = cur.d.x + crow[im & 0xffu];
                    dval.y = cur.d.y + crow[im >> 8];
                } else dval = cur.d;
                const uint32_t c0[5] = {cur.w0.x, cur.w0.y, cur.w0.z, cur.w0.w, NH == 8 ? 0u : cur.w1.x}, k0[3] = {NH == 8 ? cur.w1.x : cur.w1.y, NH == 8 ? cur.w1.y : cur.w1.z, NH == 8 ? 0u : cur.w1.w};
                const uint32_t c1[5] = {NH == 8 ? cur.w1.z : cur.w2.x, NH == 8 ? cur.w1.w : cur.w2.y, NH == 8 ? cur.w2.x : cur.w2.z, NH == 8 ? cur.w2.y : cur.w2.w, NH == 8 ? 0u : cur.w3.x},
                               k1[3] = {NH == 8 ? cur.w2.z : cur.w3.y, NH == 8 ? cur.w2.w : cur.w3.z, NH == 8 ? 0u : cur.w3.w};
                double a0 = dval.x * x.x, a1 = dval.y * x.y;
#pragma unroll
                for (int e = 0; e < NH; e++) {
                    const uint32_t code0 = (k0[e >> 2] >> (8 * (e & 3))) & 0xffu, col0 = (c0[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                    const uint32_t code1 = (k1[e >> 2] >> (8 * (e & 3))) & 0xffu, col1 = (c1[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                    double t0 = tv[code0 & 0x7fu], t1 = tv[code1 & 0x7fu];
                    t0 = (code0 & 0x80u) ? -t0 : t0; t1 = (code1 & 0x80u) ? -t1 : t1;
                    a0 += t0 * lds[col0]; a1 += t1 * lds[col1];
                }
                dbl2v o; o.x = a0; o.y = a1;
                *reinterpret_cast<dbl2v *>(A.t_out + row * ld + 2 * p) = o;
            }
            cur = nxt;
        }
        __syncthreads();
        row = next;
    }
    const double rr = block_sum(bsum, red);
    if (tid == 0) A.partial[blockIdx.x] = rr;
}

// Local pass for rows too long for LDS (Ns=18: DimUp=48620, 389 KB): the up-hop gathers go to the row in
// global memory.  `rsplit` workgroups with equal blockIdx % 8 (one XCD under round-robin dispatch) share a row
// so it is served from that XCD's L2; H_up comes from an element-major packed image of 8 dwords per up
// configuration (5 dwords of 16-bit columns + 3 dwords of 8-bit codes: up to 10 hops) -- 32 B instead of the
// 108 B of the wide image, which was the bulk of the L2 traffic of the unpacked kernel.
struct UpBigArgs {
    int32_t DimUp, DimPh, rsplit; int64_t Qdw, dw0, nrows;
    const uint4 *packed8;     // [iup][2] uint4
    const double *tval; int32_t nterms;
    const double *diag, *Eup, *Edw, *cross; const int32_t *mapup, *mapdw; int32_t impmask, nI; double econst;
    const double *v; double *out; const double *uprev; const double *scal;
    int32_t nodiag;           // 1: hops only (the transposed dw pass re-uses this kernel with H_dw's image)
};
template <bool DIRECT, bool EPI>
__global__ void __launch_bounds__(512) k_uppass_big(UpBigArgs A)
{
    __shared__ double tv[128];
    for (int t = threadIdx.x; t < A.nterms && t < 128; t += blockDim.x) tv[t] = A.tval[t];
    __syncthreads();
    const int64_t g = blockIdx.x / (8 * A.rsplit); const int w = (int)(blockIdx.x - g * 8 * A.rsplit);
    const int64_t row = g * 8 + (w & 7); const int part = w >> 3;
    if (row >= A.nrows) return;
    const int DimUp = A.DimUp;
    const int64_t idwl = row % A.Qdw, idw = A.dw0 + idwl, rbase = row * DimUp;
    const double *vrow = A.v + rbase;
    const int chunk = (DimUp + A.rsplit - 1) / A.rsplit, lo = part * chunk, hi = lo + chunk < DimUp ? lo + chunk : DimUp;
    double s_cur = 1.0, c_prev = 0.0;
    if (EPI) { s_cur = A.scal[SC_SCUR]; c_prev = A.scal[SC_BETA] * A.scal[SC_SPREV]; }
    uint32_t mdw = 0; double edw = 0.0;
    if (DIRECT) { mdw = (uint32_t)A.mapdw[idw] & A.impmask; edw = A.Edw[idw] + A.econst; }
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint4 p0 = A.packed8[2 * i], p1 = A.packed8[2 * i + 1];
        const double x = vrow[i];
        double dval;
        if (DIRECT) dval = edw + A.Eup[i] + A.cross[((uint32_t)A.mapup[i] & A.impmask) * A.nI + mdw];
        else dval = A.nodiag ? 0.0 : A.diag[idwl * DimUp + i];
        double up = 0.0;
        if (EPI) up = A.uprev[rbase + i];
        const uint32_t c[5] = {p0.x, p0.y, p0.z, p0.w, p1.x}, k[3] = {p1.y, p1.z, p1.w};
        double acc = dval * x;
#pragma unroll
        for (int e = 0; e < 10; e++) {
            const uint32_t code = (k[e >> 2] >> (8 * (e & 3))) & 0xffu, col = (c[e >> 1] >> (16 * (e & 1))) & 0xffffu;
            double t = tv[code & 0x7fu];
            t = (code & 0x80u) ? -t : t;
            acc += t * vrow[col];
        }
        A.out[rbase + i] = EPI ? s_cur * acc - c_prev * up : acc;
    }
}

// The same local half for up factors with 9-16 hops per configuration (multi-orbital replica baths, BASELINE config 5):
// one configuration per lane and step, packed ELL image of k_uppass.  Staging the next row pair by pair inside the hop loop
// (as the pair form does) would need the 13 steps of a row unrolled around a 16-hop body -- ~300 spilled registers -- so
// here only r of the next row is prefetched into registers during the hop loop; u_prev of the row is loaded at the row
// switch, where it is combined into u (one exposed memory round trip per row, ~10 % of a row's time).
template <bool DIRECT, bool ACC, int NT>
__global__ void __launch_bounds__(NT) k_uppass_ax(UpAxArgs A)
{
    extern __shared__ double lds[];
    constexpr int MAXQ = 13312 / NT, MAXC = 4;
    const int DimUp = A.DimUp, tid = threadIdx.x;
    const int64_t ld = A.ld ? A.ld : DimUp;
    double *tv = lds + ((DimUp + 1) & ~1);
    double *red = tv + 128;
    double *crow = red + 32;
    uint8_t *imp8 = reinterpret_cast<uint8_t *>(crow + 32);      // DIRECT: impurity word of every up configuration, staged once per launch
    for (int t = tid; t < A.nterms; t += NT) tv[t] = A.tval[t];
    if (DIRECT) for (int t = tid; t < DimUp; t += NT) imp8[t] = (uint8_t)((uint32_t)A.mapup[t] & A.impmask);
    const int nq = (DimUp + NT - 1) / NT;
    const int ncw = (A.upw + 1) >> 1, ndw = (A.upw + 3) >> 2;
    const double c_ax = A.scal[SC_ALPHA] * A.scal[SC_SCUR];
    const double zk = ACC ? A.zvec[(int)A.scal[SC_K]] : 0.0;
    double bsum = 0.0;
    double xr[MAXQ];
    int64_t row = blockIdx.x;
    double crn = 0.0;      // DIRECT: row constant of the row about to enter LDS (k_crow_table), requested a row ahead
    if (DIRECT) crn = A.crowtab[(A.dw0 + (row < A.nrows ? row : 0) % A.Qdw) * A.nI + (tid < A.nI ? tid : 0)];
    if (row < A.nrows) {
#pragma unroll
        for (int q = 0; q < MAXQ; q++) { const int i = tid + q * NT; xr[q] = (q < nq && i < DimUp) ? A.r[row * ld + i] : 0.0; }
    }
    struct Ops { double d; uint32_t codes[MAXC], cols[2 * MAXC]; };
    while (row < A.nrows) {
        const int64_t idwl = row % A.Qdw;
        auto load_ops = [&](int i, Ops &o) {
            o.d = DIRECT ? A.Eup[i] : A.diag[idwl * DimUp + i];
            if (A.pelem16) {      // three 16-byte loads: columns 0-7, 8-15, codes (padding slots: the row's own position, code 127)
                const uint4 *pe = A.pelem16 + 3 * (int64_t)i;
                const uint4 a = pe[0], b = pe[1], c4 = pe[2];
                o.cols[0] = a.x; o.cols[1] = a.y; o.cols[2] = a.z; o.cols[3] = a.w; o.cols[4] = b.x; o.cols[5] = b.y; o.cols[6] = b.z; o.cols[7] = b.w;
                o.codes[0] = c4.x; o.codes[1] = c4.y; o.codes[2] = c4.z; o.codes[3] = c4.w;
            } else {
#pragma unroll
            for (int c = 0; c < MAXC; c++) if (c < ndw) o.codes[c] = A.pcodes[(int64_t)c * DimUp + i];
#pragma unroll
            for (int c = 0; c < 2 * MAXC; c++) o.cols[c] = c < ncw ? A.pcols[(int64_t)c * DimUp + i] : ((uint32_t)i | ((uint32_t)i << 16));      // slots beyond the image (odd widths): the row's own position, like any padding slot -- never an uninitialised LDS index
            }
        };
        Ops cur, nxt;
        if (tid < DimUp) load_ops(tid, cur);
        // u = r - c * u_prev for this row: into LDS, back over r, into the norm (and the eigenvector accumulation)
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
            const int i = tid + q * NT;
            if (q < nq && i < DimUp) {
                const double u = xr[q] - c_ax * A.vprev[row * ld + i];
                lds[i] = u; A.r[row * ld + i] = u; bsum += u * u;
                if (ACC) A.acc[row * ld + i] += zk * u;
            }
        }
        if (DIRECT && tid < A.nI) crow[tid] = crn;
        __syncthreads();
        const int64_t next = row + gridDim.x;
        if (DIRECT) crn = A.crowtab[(A.dw0 + (next < A.nrows ? next : row) % A.Qdw) * A.nI + (tid < A.nI ? tid : 0)];
        if (next < A.nrows) {
#pragma unroll
            for (int q = 0; q < MAXQ; q++) { const int i = tid + q * NT; if (q < nq && i < DimUp) xr[q] = A.r[next * ld + i]; }
        }
        for (int i = tid; i < DimUp; i += NT) {
            if (i + NT < DimUp) load_ops(i + NT, nxt);
            const double x = lds[i];
            const double dval = DIRECT ? cur.d + crow[imp8[i]] : cur.d;
            double acc = dval * x;
#pragma unroll
            for (int c = 0; c < MAXC; c++)
                if (c < ndw && __any(cur.codes[c] != 0x7f7f7f7fu)) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const uint32_t code = (cur.codes[c] >> (8 * e)) & 0xffu;
                        const uint32_t col = (cur.cols[2 * c + (e >> 1)] >> (16 * (e & 1))) & 0xffffu;
                        double t = tv[code & 0x7fu];
                        t = (code & 0x80u) ? -t : t;
                        acc += t * lds[col];
                    }
                }
            A.t_out[row * ld + i] = acc;
            cur = nxt;
        }
        __syncthreads();
        row = next;
    }
    const double rr = block_sum(bsum, red);
    if (tid == 0) A.partial[blockIdx.x] = rr;
}

// DIRECT storage: per dw configuration the part of the diagonal that does not depend on the up bath levels, one value per up
// impurity word: Edw + econst + cross(up word, dw word).  n*nI doubles, rebuilt per product like the other on-the-fly tables.
__global__ void k_crow_table(const double *Edw, const int32_t *mapdw, const double *cross, double econst, int32_t impmask, int32_t nI, int64_t n, double *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nI) return;
    const int64_t idw = t / nI; const int i = (int)(t - idw * nI);
    out[t] = Edw[idw] + econst + cross[i * nI + ((uint32_t)mapdw[idw] & impmask)];
}

// Local pass for rows too long for LDS, segment + halo form (Ns=17/18).  The up configurations are sorted by value, so a
// contiguous 1/16 of a row is (nearly) "the configurations with given top bath bits": three quarters of the up hops of an
// element stay inside its own segment, only the hops on the top levels leave it.  A persistent 1024-thread workgroup owns
// ONE segment of the rows it walks.  That segment is fixed for the whole launch, so the columns outside it that its hops
// reach are a fixed ascending list (seg_ext, built once per sector): their values are fetched together with the segment
// -- one bulk prefetch per row, a full row ahead, with the list itself held in registers -- into a halo area behind the
// segment in LDS, and the hop loop gathers from LDS only (pseg8 holds LDS positions instead of columns).  Ns=18: 16
// segments of 3 040 elements + <= 7 962 halo values = 86 KB of LDS; 2.3 halo values per element, read from the XCD's L2
// where the sibling workgroups of the same row (same blockIdx % 8) are loading them as their own segments at the same time.
// Two earlier forms: k_uppass_big (all gathers through the texture-address path, 27 ms at Ns=18) and a segment form that
// sent only the out-of-segment gathers to global memory from inside the hop loop (25 ms: every such gather is a dependent
// load the whole workgroup waits for -- 66 % of the wave cycles waiting, HBM at 2 TB/s, VALU 39 %, LDS 21 % busy).
struct UpHaloArgs {
    int32_t DimUp, DimPh, nseg, seglen, halocap; int64_t Qdw, dw0, nrows;
    const uint4 *pseg8; const int32_t *ext, *ext_off;
    const double *tval; int32_t nterms;
    const double *diag, *Eup; int32_t nI;
    const double *v; double *out;
    int32_t ld;              // row pitch of v / out in doubles (0: DimUp)
    const double *crowtab;   // DIRECT: [idw][nI] Edw + econst + cross(up impurity word, dw impurity word), rebuilt with the other on-the-fly tables
};
constexpr int kHaloSegMax = 6144, kHaloMax = 12288;     // up to 3 pairs and 12 halo values per lane of a 1024-thread workgroup
// NQ: pairs per lane = steps per (row, segment) item; MAXH: halo values per lane; NH: hop slots walked per element (9 at half
// filling of Ns=18: the tenth slot of the image is padding there).
// Round 4 (profiles/r04/ns18_local_pass.txt):
//  * the hop words of a lane (pseg8: LDS positions + term codes of its NQ element pairs, and Eup for the on-the-fly diagonal)
//    do not depend on the row -- the segment of a workgroup is fixed for the launch -- so they are loaded ONCE and stay in
//    registers (the first form re-read 64 bytes per pair and row from L2);
//  * amplitudes come from a 256-entry SIGNED table (index = the code byte itself, sign bit included) at LDS address 0: a hop
//    is 2 address instructions per LDS read + 1 FMA = 5 vector instructions, where the mask / compare / xor / select of the
//    sign made it 10 -- the pass was bound by vector issue (4.15e9 wave instructions per launch, 6.8 ms of 17.4);
//  * every global load of the next row (segment pairs, halo values, stored diagonal, row constant) is issued right after the
//    first barrier, BEFORE this row's stores: on gfx9 loads and stores retire through one in-order counter (vmcnt), so a load
//    issued behind a store cannot be consumed before that store has been acknowledged; now nothing the next LDS fill waits
//    for sits behind a store of the same item.
template <bool DIRECT, int NQ, int MAXH, int NH>
__global__ void __launch_bounds__(1024, 4) k_uppass_halo(UpHaloArgs A)      // 4 waves per SIMD: one workgroup of 1024 per CU
{
    extern __shared__ double lds[];
    constexpr int NT = 1024;
    const int DimUp = A.DimUp, tid = threadIdx.x;
    const int64_t ld = A.ld ? A.ld : DimUp;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, G = (int)(gridDim.x >> 3) / A.nseg;
    const int g = j / A.nseg, sg = j - g * A.nseg;
    const int lo = sg * A.seglen, hi = lo + A.seglen < DimUp ? lo + A.seglen : DimUp;
    const int npair = (hi - lo) >> 1;
    const int eoff = A.ext_off[sg], nh = A.ext_off[sg + 1] - eoff;
    double *tv = lds;                                    // signed amplitude table: tv[code], code = sign << 7 | term (127: padding slot, 0)
    double *seg = lds + 256;                             // [segment | halo]: the positions stored in pseg8 index this
    double *halo = seg + A.seglen;
    double *crow = halo + A.halocap;
    for (int t = tid; t < 256; t += NT) { const int k = t & 127; const double a = k < A.nterms ? A.tval[k] : 0.0; tv[t] = (t & 128) ? -a : a; }
    if (g >= G) return;
    uint32_t hc[MAXH / 2];                               // this lane's halo columns, two 16-bit values per register: the same for every row
#pragma unroll
    for (int h = 0; h < MAXH; h += 2) {
        const int t0 = tid + h * NT, t1 = t0 + NT;
        hc[h >> 1] = (t0 < nh ? (uint32_t)A.ext[eoff + t0] : 0u) | ((t1 < nh ? (uint32_t)A.ext[eoff + t1] : 0u) << 16);
    }
    const int pmax = npair > 0 ? npair - 1 : 0;
    uint4 w[NQ][4]; dbl2v eup[NQ];                       // resident: hop words of the lane's pairs; Eup of their elements (DIRECT)
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        int p = tid + q * NT; p = p < npair ? p : pmax;
        const int i = lo + 2 * p;
        w[q][0] = A.pseg8[2 * i]; w[q][1] = A.pseg8[2 * i + 1]; w[q][2] = A.pseg8[2 * i + 2]; w[q][3] = A.pseg8[2 * i + 3];
        if (DIRECT) eup[q] = *reinterpret_cast<const dbl2v *>(A.Eup + i);
    }
    dbl2v xr[NQ], dg[NQ]; double xh[MAXH]; double crn = 0.0;
    const int64_t rstep = (int64_t)G * 8, lastrow = A.nrows - 1;
    // No branch around a prefetch load, neither per lane nor uniform: lanes and rows out of range load a clamped address and
    // discard the value (a wait for an older load is `s_waitcnt vmcnt(N)` with the SMALLEST N over all paths: one path that
    // skips the prefetch would turn every such wait into vmcnt(0), scripts/isa_waits.py).
    auto fetch_row = [&](int64_t rw) {                   // rows = dw columns of this rank (DimPh == 1 here: phonon sectors use k_kron2)
        rw = rw < lastrow ? rw : lastrow;
        const double *vr = A.v + rw * ld;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            int p = tid + q * NT; p = p < npair ? p : pmax;
            xr[q] = *reinterpret_cast<const dbl2v *>(vr + lo + 2 * p);
            if (!DIRECT) dg[q] = *reinterpret_cast<const dbl2v *>(A.diag + rw * DimUp + lo + 2 * p);
        }
#pragma unroll
        for (int h = 0; h < MAXH; h++) xh[h] = vr[(hc[h >> 1] >> (16 * (h & 1))) & 0xffffu];      // entries beyond the list hold column 0
        if (DIRECT) crn = A.crowtab[(A.dw0 + rw) * A.nI + (tid < A.nI ? tid : 0)];      // Edw + const + cross term of that row (k_crow_table)
    };
    int64_t row = (int64_t)g * 8 + x;
    fetch_row(row);
    while (row < A.nrows) {
#pragma unroll
        for (int q = 0; q < NQ; q++) { const int p = tid + q * NT; if (p < npair) *reinterpret_cast<dbl2v *>(seg + 2 * p) = xr[q]; }
#pragma unroll
        for (int h = 0; h < MAXH; h++) if (tid + h * NT < nh) halo[tid + h * NT] = xh[h];
        if (DIRECT && tid < A.nI) crow[tid] = crn;
        // the hop words are loop invariant, and so is every LDS address derived from them: left alone the compiler hoists all
        // 4 * NH * NQ address computations out of the row loop and spills what does not fit (64-100 bytes of scratch, reloaded
        // in every item through the vector-memory pipe).  Opaque per iteration: addresses are formed next to their reads.
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int c = 0; c < 4; c++) asm volatile("" : "+v"(w[q][c].x), "+v"(w[q][c].y), "+v"(w[q][c].z), "+v"(w[q][c].w));
        dbl2v dcur[NQ];                                  // stored diagonal of this row (the prefetch below overwrites dg)
#pragma unroll
        for (int q = 0; q < NQ; q++) dcur[q] = DIRECT ? eup[q] : dg[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int p = tid + q * NT;
            dbl2v r = {0.0, 0.0};
            if (p < npair) {
                const dbl2v xx = *reinterpret_cast<const dbl2v *>(seg + 2 * p);
                dbl2v dval = dcur[q];
                if (DIRECT) {           // byte 10 of the code dwords: the impurity word of the up configuration (k_seg_remap)
                    dval.x += crow[(w[q][1].w >> 16) & 0xffu];
                    dval.y += crow[(w[q][3].w >> 16) & 0xffu];
                }
                const uint32_t c0[5] = {w[q][0].x, w[q][0].y, w[q][0].z, w[q][0].w, w[q][1].x}, k0[3] = {w[q][1].y, w[q][1].z, w[q][1].w};
                const uint32_t c1[5] = {w[q][2].x, w[q][2].y, w[q][2].z, w[q][2].w, w[q][3].x}, k1[3] = {w[q][3].y, w[q][3].z, w[q][3].w};
                double a0 = dval.x * xx.x, a1 = dval.y * xx.y;
#pragma unroll
                for (int e = 0; e < NH; e++) {
                    const uint32_t code0 = (k0[e >> 2] >> (8 * (e & 3))) & 0xffu, col0 = (c0[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                    const uint32_t code1 = (k1[e >> 2] >> (8 * (e & 3))) & 0xffu, col1 = (c1[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                    a0 += tv[code0] * seg[col0]; a1 += tv[code1] * seg[col1];
                }
                r.x = a0; r.y = a1;
            }
            // the whole next row is requested after the first step's arithmetic (its staging registers are free until then) and
            // BEFORE the first store of this row
            if (q == 0) fetch_row(row + rstep);
            if (p < npair) *reinterpret_cast<dbl2v *>(A.out + row * ld + lo + 2 * p) = r;
        }
        __syncthreads();
        row += rstep;
    }
}

// pelem8 (columns) -> pseg8 (positions in the [segment | halo] LDS image of the segment that owns the element)
__global__ void k_seg_remap(const uint32_t *pelem8, uint32_t *pseg8, const int32_t *ext, const int32_t *ext_off, int32_t dim, int32_t seglen, int32_t halo_base, const int32_t *map, int32_t impmask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dim) return;
    const int sg = i / seglen, lo = sg * seglen, hi = lo + seglen < dim ? lo + seglen : dim;
    const int32_t *e = ext + ext_off[sg]; const int n = ext_off[sg + 1] - ext_off[sg];
    uint32_t w[8];
    for (int c = 0; c < 8; c++) w[c] = pelem8[(size_t)8 * i + c];
    for (int q = 0; q < 10; q++) {
        const uint32_t col = (w[q >> 1] >> (16 * (q & 1))) & 0xffffu;
        uint32_t pos;
        if ((int)col >= lo && (int)col < hi) pos = col - lo;
        else {
            int a = 0, b = n;                             // first entry >= col
            while (a < b) { const int m = (a + b) >> 1; if (e[m] < (int)col) a = m + 1; else b = m; }
            pos = (uint32_t)(halo_base + a);              // present by construction (the list is the set of such columns)
        }
        w[q >> 1] = (w[q >> 1] & ~(0xffffu << (16 * (q & 1)))) | (pos << (16 * (q & 1)));
    }
    w[7] = (w[7] & ~0x00ff0000u) | (((uint32_t)map[i] & (uint32_t)impmask & 0xffu) << 16);      // code byte 10: impurity word (on-the-fly diagonal)
    for (int c = 0; c < 8; c++) pseg8[(size_t)8 * i + c] = w[c];
}

// ---------------------------------------------------------------------------------------------
// dw-hop pass, column tiled: out(iup,idw) += sum_k Hdw[idw,k] * v(iup, jdw_k).
// Every neighbour row jdw is needed by ~Ns/2 different idw rows that are spread over the whole
// sector, so a row-major sweep re-fetches v about Ns/2 times from HBM.  Here the sweep is over
// narrow iup tiles instead: the slab v(tile, all idw) (DimDw*w*8 bytes, 3.3 MB for w=32 at
// Ns=16) stays in one XCD's 4 MB L2 while that XCD walks all idw rows of the tile, so the
// gathers are L2 hits and HBM sees v once.  Tiles are dealt to XCDs by blockIdx % 8 (the
// observed round-robin dispatch; speed only -- any placement gives the same result).
// ---------------------------------------------------------------------------------------------
struct DwArgs {
    int32_t DimUp, DimDw, DimPh; int64_t Qdw, dw0;
    const int32_t *dwc; const double *dwv; int32_t dww;
    const double *v, *vfull; double *out;
    const double *scal; double *partial;
    int32_t w, wshift, ntiles, R, chunks;
    int32_t ld;                // row pitch of v / vfull / out / uprev2 in doubles (0: DimUp)
    const double *uprev2;      // fused Lanczos iteration (EPI only): out holds the UNSCALED local part; out = s_cur*(out + dw part) - beta*s_prev*uprev2
    int32_t own_in_v;          // several ranks, needed-rows exchange: x-space rows < Qdw (the rank's own rows) are read from v itself, only received rows from vfull (no own-row copy per product)
#ifdef EDGPU_EXPERIMENTS
    int32_t abl;               // experiments build, timing only (WRONG results): 1 every gather reads the row's own segment, 2 no result store, 4 gathers from rows (j & 15), 8 no running-result read
#endif
};

// Column tiles dealt to the XCDs (workgroups go round-robin over the 8 XCDs): XCD x owns tiles x, x+8, ... and walks all row
// chunks of a tile, so the slab x(tile, all idw) the dw hops gather from is shared through that XCD's L2.  Only whole groups
// of 8 tiles are dealt that way; the ntiles % 8 left-over tiles are swept by all XCDs together (consecutive blocks =
// consecutive chunks), so that no XCD carries one tile more than the others (c3: 202 tiles = 25 per XCD + 2 shared instead
// of 26 / 25) and no block is idle.  Grid = ntiles * chunks.
__device__ __forceinline__ void xcd_tile_map(uint32_t b, uint32_t ntiles, uint32_t chunks, int &tile, uint32_t &chunk)
{
    const uint32_t t8 = ntiles & ~7u, nmain = t8 * chunks;
    if (b < nmain) { const uint32_t xcd = b & 7u, sidx = b >> 3, tl = sidx / chunks; chunk = sidx - tl * chunks; tile = (int)(tl * 8u + xcd); }
    else { const uint32_t r = b - nmain, tl = r / chunks; chunk = r - tl * chunks; tile = (int)(t8 + tl); }
}

template <bool EPI>
__global__ void __launch_bounds__(256) k_dwpass(DwArgs A)
{
    __shared__ double red[4];
    int tile; uint32_t chunk;
    xcd_tile_map(blockIdx.x, (uint32_t)A.ntiles, (uint32_t)A.chunks, tile, chunk);
    double asum = 0.0;
    if (tile < A.ntiles) {
        const int j = threadIdx.x & (A.w - 1), r = threadIdx.x >> A.wshift;
        const int64_t row = (int64_t)chunk * A.R + r;
        const int iup = tile * A.w + j;
        if (row < A.Qdw * A.DimPh && iup < A.DimUp) {
            const int iph = (int)(row / A.Qdw);
            const int64_t idw = A.dw0 + (row - (int64_t)iph * A.Qdw);
            const int64_t e = row * A.DimUp + iup, fbase = (int64_t)iph * A.DimDw * A.DimUp + iup;
            const int32_t *dwc = A.dwc + idw * A.dww; const double *dwv = A.dwv + idw * A.dww;
            double acc = 0.0;
            for (int k = 0; k < A.dww; k++) acc += dwv[k] * A.vfull[fbase + (int64_t)dwc[k] * A.DimUp];
            if (EPI) { const double s_cur = A.scal[SC_SCUR]; const double wv = A.out[e] + s_cur * acc; A.out[e] = wv; asum = (s_cur * A.v[e]) * wv; }
            else A.out[e] += acc;
        }
    }
    if (EPI) { double rr = block_sum(asum, red); if (threadIdx.x == 0) A.partial[blockIdx.x] = rr; }
}

// Same pass with two iup per lane (16-byte loads/stores; needs DimUp even) and the Hdw rows of
// the block staged in LDS once, which removes the per-lane table loads: ~5x fewer vector-memory
// instructions per element than k_dwpass.
template <bool EPI, bool NT, bool RAGGED>   // RAGGED: rows have different hop counts (multi-orbital models): skip the ELL padding
__global__ void __launch_bounds__(256) k_dwpass2(DwArgs A)
{
    // dynamic LDS sized by the launch to R*dww entries (was a fixed 64x64 image = 48 KB, i.e. 3 workgroups per CU whatever
    // the factor; 8 hops need 6 KB and leave room for the 8 workgroups a CU can hold)
    extern __shared__ double dyn_lds[];
    __shared__ double red[4];
    __shared__ int32_t tn[64];
    double *tv = dyn_lds;                                               // [row in block][k]
    int32_t *tc = reinterpret_cast<int32_t *>(dyn_lds + A.R * A.dww);
    int tile; uint32_t chunk;
    xcd_tile_map(blockIdx.x, (uint32_t)A.ntiles, (uint32_t)A.chunks, tile, chunk);
    double asum = 0.0;
    if (tile < A.ntiles) {
        const int64_t nrows = A.Qdw * A.DimPh, row0 = (int64_t)chunk * A.R;
        const int rb = (int)((nrows - row0) < A.R ? (nrows - row0) : A.R);   // rows in this block
        for (int t = threadIdx.x; t < rb * A.dww; t += 256) {
            const int r = t / A.dww, k = t - r * A.dww;
            const int64_t row = row0 + r; const int64_t idw = A.dw0 + (row % A.Qdw);
            tv[t] = A.dwv[idw * A.dww + k]; tc[t] = A.dwc[idw * A.dww + k];
        }
        __syncthreads();
        if (RAGGED) {
            if ((int)threadIdx.x < rb) { int n = A.dww; while (n > 0 && tv[threadIdx.x * A.dww + n - 1] == 0.0) n--; tn[threadIdx.x] = n; }
            __syncthreads();
        }
        const int lpr = A.w >> 1;                    // lanes per row
        const int j = threadIdx.x & (lpr - 1), r0 = threadIdx.x >> (A.wshift - 1), rstep = 256 >> (A.wshift - 1);
        const int iup = tile * A.w + 2 * j;
        const double s_cur = EPI ? A.scal[SC_SCUR] : 1.0;
        const double c_prev2 = (EPI && A.uprev2) ? A.scal[SC_BETA] * A.scal[SC_SPREV] : 0.0;
        const int64_t ld = A.ld ? A.ld : A.DimUp;
        if (iup < A.DimUp)
            for (int r = r0; r < rb; r += rstep) {
                const int64_t row = row0 + r;
                const int iph = (int)(row / A.Qdw);
                const int64_t e = row * ld + iup, fbase = (int64_t)iph * A.DimDw * ld + iup;
                dbl2v acc = {0.0, 0.0};
                const double *tvr = tv + r * A.dww; const int32_t *tcr = tc + r * A.dww;
                const int nk = RAGGED ? tn[r] : A.dww;   // hops of this row (ELL padding sits at the end: not gathered)
                dbl2v *po = reinterpret_cast<dbl2v *>(A.out + e);
                dbl2v o = {0.0, 0.0}, vl = {0.0, 0.0}, up = {0.0, 0.0};
                // rolled gather loop: 4 gathers in flight per lane; issuing all of a row step up front (103-120 VGPRs, half the
                // residency) measured slower at Ns=18 (profiles/r03/ns18_product_variants.txt)
                if (A.own_in_v) {
                    const int64_t obase = (int64_t)iph * A.Qdw * ld + iup;
                    for (int k = 0; k < nk; k++) {
                        const int64_t xr = tcr[k];
                        const dbl2v x = *reinterpret_cast<const dbl2v *>(xr < A.Qdw ? A.v + obase + xr * ld : A.vfull + fbase + xr * ld);
                        acc += tvr[k] * x;
                    }
                } else
                for (int k = 0; k < nk; k++) {
#ifdef EDGPU_EXPERIMENTS
                    const int64_t xr = (A.abl & 1) ? row % A.Qdw + A.dw0 : ((A.abl & 4) ? (int64_t)(tcr[k] & 15) : (int64_t)tcr[k]);
                    const dbl2v x = *reinterpret_cast<const dbl2v *>(A.vfull + fbase + xr * ld);
#else
                    const dbl2v x = *reinterpret_cast<const dbl2v *>(A.vfull + fbase + (int64_t)tcr[k] * ld);
#endif
                    acc += tvr[k] * x;
                }
#ifdef EDGPU_EXPERIMENTS
                if (!(A.abl & 8))
#endif
                o = NT ? __builtin_nontemporal_load(po) : *po;
                if (EPI) { vl = *reinterpret_cast<const dbl2v *>(A.v + e); if (A.uprev2) up = *reinterpret_cast<const dbl2v *>(A.uprev2 + e); }
                if (EPI) {
                    if (A.uprev2) o = s_cur * (o + acc) - c_prev2 * up;
                    else o += s_cur * acc;
                    asum += (s_cur * vl.x) * o.x + (s_cur * vl.y) * o.y;
                } else o += acc;
#ifdef EDGPU_EXPERIMENTS
                if ((A.abl & 2) && o.x != 1.2345e300) continue;
#endif
                if (NT) __builtin_nontemporal_store(o, po); else *po = o;
            }
    }
    if (EPI) { double rr = block_sum(asum, red); if (threadIdx.x == 0) A.partial[blockIdx.x] = rr; }
}

// ---------------------------------------------------------------------------------------------
// Non-local pass of the split product: out += [s_cur *] (spin-exchange Jx + pair-hopping Jp terms),
// stored/H_non_local.f90:21-83 / direct/HxV_non_local.f90 evaluated on the fly.  Both terms move one
// impurity electron in EACH spin word, so for a row (idw) the dw word fixes which (a,b) pairs can act and the
// partner row jdw; rows whose dw impurity bits allow nothing (half of them for two orbitals) are skipped
// without touching memory, and inside a row only the configurations with the matching up impurity bits
// gather (from the partner row, at an index that grows with iup) and are written back.
// Term order per element = that of k_kron2 (a outer, b inner, Jx then Jp).
// ---------------------------------------------------------------------------------------------
struct NdArgs {
    int32_t DimUp, DimDw, DimPh, Norb; int64_t Qdw, dw0;
    const int32_t *mapup, *mapdw, *rankup, *rankdw;
    const int32_t *xmap;     // needed-rows exchange: global dw row -> row of the compact x-space (else null)
    double Jx, Jp;
    const double *vfull; double *out; const double *scal;
    int32_t ld;              // row pitch of vfull / out in doubles (0: DimUp)
};
template <bool EPI>
__global__ void __launch_bounds__(256) k_ndpass(NdArgs A)
{
    __shared__ int32_t s_jdw[2 * EDGPU_MAX_ORB * EDGPU_MAX_ORB];      // partner row, -1: term cannot act on this row
    __shared__ double s_t[2 * EDGPU_MAX_ORB * EDGPU_MAX_ORB];         // +-J with the sign of the dw word
    __shared__ int s_any;
    const int64_t row = blockIdx.x;
    const int iph = (int)(row / A.Qdw);
    const int64_t idwl = row - (int64_t)iph * A.Qdw, idw = A.dw0 + idwl;
    const uint32_t mdw = (uint32_t)A.mapdw[idw];
    const int nslot = 2 * A.Norb * A.Norb;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    if ((int)threadIdx.x < nslot) {
        const int kind = threadIdx.x & 1, ab = threadIdx.x >> 1, a = ab / A.Norb, b = ab - a * A.Norb;
        const uint32_t ba = 1u << a, bb = 1u << b;
        int32_t jd = -1; double t = 0.0;
        if (a != b) {
            if (kind == 0 && A.Jx != 0.0 && (mdw & ba) && !(mdw & bb)) {
                const uint32_t k1 = mdw & ~ba, k2 = k1 | bb;
                const int sg = __popc(mdw & (ba - 1)) + __popc(k1 & (bb - 1));
                jd = A.rankdw[k2]; t = (sg & 1) ? -A.Jx : A.Jx;
            }
            if (kind == 1 && A.Jp != 0.0 && (mdw & bb) && !(mdw & ba)) {
                const uint32_t k1 = mdw & ~bb, k2 = k1 | ba;
                const int sg = __popc(mdw & (bb - 1)) + __popc(k1 & (ba - 1));
                jd = A.rankdw[k2]; t = (sg & 1) ? -A.Jp : A.Jp;
            }
        }
        if (jd >= 0 && A.xmap) jd = A.xmap[jd];
        s_jdw[threadIdx.x] = jd; s_t[threadIdx.x] = t;
        if (jd >= 0) s_any = 1;
    }
    __syncthreads();
    if (!s_any) return;
    const int DimUp = A.DimUp;
    const int64_t ld = A.ld ? A.ld : DimUp;
    const int64_t rbase = row * ld, fbase = (int64_t)iph * A.DimDw * ld;
    const double s_cur = EPI ? A.scal[SC_SCUR] : 1.0;
    for (int iup = threadIdx.x; iup < DimUp; iup += blockDim.x) {
        const uint32_t mup = (uint32_t)A.mapup[iup];
        double acc = 0.0; bool hit = false;
        for (int a = 0; a < A.Norb; a++)
            for (int b = 0; b < A.Norb; b++) {
                const uint32_t ba = 1u << a, bb = 1u << b;
                if (a == b || !(mup & bb) || (mup & ba)) continue;
                const uint32_t k3 = mup & ~bb, k4 = k3 | ba;
                const int sgu = __popc(mup & (bb - 1)) + __popc(k3 & (ba - 1));
                const int slot = 2 * (a * A.Norb + b);
#pragma unroll
                for (int kind = 0; kind < 2; kind++) {
                    const int32_t jd = s_jdw[slot + kind];
                    if (jd < 0) continue;
                    const double t = (sgu & 1) ? -s_t[slot + kind] : s_t[slot + kind];
                    acc += t * A.vfull[fbase + (int64_t)jd * ld + A.rankup[k4]];
                    hit = true;
                }
            }
        if (hit) A.out[rbase + iup] += s_cur * acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Stored form of the non-local block (the reference's spH0nd, ED_HAMILTONIAN_SPARSE_HxV.f90:473-483, filled by
// stored/H_non_local.f90:21-83) for storage=CSR.  Most rows of it are empty (a spin-exchange / pair-hopping term needs
// matching impurity bits in both spin words), so it is kept as a compressed-row matrix: the ascending list of non-empty
// rows, their entry offsets, and the entries (column in the x-space of the dw block, value, and the row's position in
// its group of 64 non-empty rows).  Built on the device in two sweeps of the k_ndpass term loop (count, fill) with a host
// scan over the per-dw-row totals in between; entries of a row in ascending column order.
// ---------------------------------------------------------------------------------------------
struct NdBuildArgs {
    int32_t DimUp, DimDw, Norb; int64_t Qdw, dw0;
    const int32_t *mapup, *mapdw, *rankup, *rankdw, *xmap; double Jx, Jp;
    int64_t *row_nr, *row_ne;                 // per dw row: non-empty elements / entries (count sweep: totals; fill sweep: exclusive offsets)
    int64_t *rows, *ptr; int32_t *cols; double *vals; uint8_t *slot;
};
constexpr int kNdMaxEnt = 2 * EDGPU_MAX_ORB * (EDGPU_MAX_ORB - 1);
__device__ __forceinline__ int nd_row_entries(const NdBuildArgs &A, const int32_t *s_jdw, const double *s_t, int iup, int32_t *cj, double *cv)
{
    const uint32_t mup = (uint32_t)A.mapup[iup];
    int n = 0;
    for (int a = 0; a < A.Norb; a++)
        for (int b = 0; b < A.Norb; b++) {
            const uint32_t ba = 1u << a, bb = 1u << b;
            if (a == b || !(mup & bb) || (mup & ba)) continue;
            const uint32_t k3 = mup & ~bb, k4 = k3 | ba;
            const int sgu = __popc(mup & (bb - 1)) + __popc(k3 & (ba - 1));
            const int sl = 2 * (a * A.Norb + b);
            for (int kind = 0; kind < 2; kind++) {
                const int32_t jd = s_jdw[sl + kind];
                if (jd < 0) continue;
                const int32_t col = (int32_t)((int64_t)jd * A.DimUp + A.rankup[k4]);
                const double t = (sgu & 1) ? -s_t[sl + kind] : s_t[sl + kind];
                int q = n++;                              // insertion by ascending column
                while (q > 0 && cj[q - 1] > col) { cj[q] = cj[q - 1]; cv[q] = cv[q - 1]; q--; }
                cj[q] = col; cv[q] = t;
            }
        }
    return n;
}
template <int FILL>
__global__ void __launch_bounds__(256) k_nd_build(NdBuildArgs A)
{
    __shared__ int32_t s_jdw[2 * EDGPU_MAX_ORB * EDGPU_MAX_ORB];
    __shared__ double s_t[2 * EDGPU_MAX_ORB * EDGPU_MAX_ORB];
    __shared__ int32_t sc_r[256], sc_e[256];
    __shared__ int64_t s_br, s_be;
    const int64_t idwl = blockIdx.x, idw = A.dw0 + idwl;
    const uint32_t mdw = (uint32_t)A.mapdw[idw];
    if ((int)threadIdx.x < 2 * A.Norb * A.Norb) {         // partner dw row and amplitude of every (a,b,kind): as k_ndpass
        const int kind = threadIdx.x & 1, ab = threadIdx.x >> 1, a = ab / A.Norb, b = ab - a * A.Norb;
        const uint32_t ba = 1u << a, bb = 1u << b;
        int32_t jd = -1; double t = 0.0;
        if (a != b) {
            if (kind == 0 && A.Jx != 0.0 && (mdw & ba) && !(mdw & bb)) {
                const uint32_t k1 = mdw & ~ba, k2 = k1 | bb;
                const int sg = __popc(mdw & (ba - 1)) + __popc(k1 & (bb - 1));
                jd = A.rankdw[k2]; t = (sg & 1) ? -A.Jx : A.Jx;
            }
            if (kind == 1 && A.Jp != 0.0 && (mdw & bb) && !(mdw & ba)) {
                const uint32_t k1 = mdw & ~bb, k2 = k1 | ba;
                const int sg = __popc(mdw & (bb - 1)) + __popc(k1 & (ba - 1));
                jd = A.rankdw[k2]; t = (sg & 1) ? -A.Jp : A.Jp;
            }
        }
        if (jd >= 0 && A.xmap) jd = A.xmap[jd];
        s_jdw[threadIdx.x] = jd; s_t[threadIdx.x] = t;
    }
    if (threadIdx.x == 0) { s_br = FILL ? A.row_nr[idwl] : 0; s_be = FILL ? A.row_ne[idwl] : 0; }
    __syncthreads();
    int32_t cj[kNdMaxEnt]; double cv[kNdMaxEnt];
    for (int base = 0; base < A.DimUp; base += 256) {
        const int iup = base + threadIdx.x;
        const int n = iup < A.DimUp ? nd_row_entries(A, s_jdw, s_t, iup, cj, cv) : 0;
        sc_r[threadIdx.x] = n > 0; sc_e[threadIdx.x] = n;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {               // inclusive scans of both counters over the chunk
            const int r = threadIdx.x >= d ? sc_r[threadIdx.x - d] : 0, e = threadIdx.x >= d ? sc_e[threadIdx.x - d] : 0;
            __syncthreads();
            sc_r[threadIdx.x] += r; sc_e[threadIdx.x] += e;
            __syncthreads();
        }
        if (FILL && n > 0) {
            const int64_t ir = s_br + sc_r[threadIdx.x] - 1, ie = s_be + sc_e[threadIdx.x] - n;
            A.rows[ir] = idwl * A.DimUp + iup; A.ptr[ir] = ie;
            for (int q = 0; q < n; q++) { A.cols[ie + q] = cj[q]; A.vals[ie + q] = cv[q]; A.slot[ie + q] = (uint8_t)(ir & 63); }
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_br += sc_r[255]; s_be += sc_e[255]; }
        __syncthreads();
    }
    if (!FILL && threadIdx.x == 0) { A.row_nr[idwl] = s_br; A.row_ne[idwl] = s_be; }
}

// Phonon part of the explicit storage (Nph > 0): the reference keeps H_eph = spH0e_eph (x) spH0ph_eph, an electronic
// diagonal g(iel) times the (b + b^+) matrix of the phonon index, and H_ph = w0 n_ph (stored/H_ph.f90:2-5,
// H_e_ph.f90:2-39).  Stored here the same way: g_el[iel] once per sector; this pass adds, for every phonon block,
//   y(iel,iph) += [s_cur *] ( w0*iph*x(iel,iph) + g(iel) * ( sqrt(iph+1) x(iel,iph+1) + sqrt(iph) x(iel,iph-1) ) )
// (phonon neighbours of a row are on the same rank).
__global__ void __launch_bounds__(256) k_build_gel(int64_t nel, int32_t DimUp, int64_t dw0, const int32_t *mapup, const int32_t *mapdw, int32_t impmask,
                                                   const double *gup, const double *gdw, double gsum, double *gel)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nel; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idwl = i / DimUp; const int iup = (int)(i - idwl * DimUp);
        gel[i] = gup[(uint32_t)mapup[iup] & impmask] + gdw[(uint32_t)mapdw[dw0 + idwl] & impmask] - gsum;
    }
}
template <bool EPI>
__global__ void __launch_bounds__(256) k_csr_ph(int64_t nel, int32_t DimPh, double w0, const double *__restrict__ gel, const double *__restrict__ x, double *__restrict__ y, const double *scal)
{
    const double sc = EPI ? scal[SC_SCUR] : 1.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nel * DimPh; i += (int64_t)gridDim.x * blockDim.x) {
        const int iph = (int)(i / nel); const int64_t iel = i - (int64_t)iph * nel;
        const double g = gel[iel];
        double acc = w0 * iph * x[i];
        if (iph + 1 < DimPh) acc += g * sqrt((double)(iph + 1)) * x[i + nel];
        if (iph > 0) acc += g * sqrt((double)iph) * x[i - nel];
        y[i] += sc * acc;
    }
}

// y += [s_cur *] ND x for the compressed-row block: a wave takes 64 consecutive non-empty rows, whose entries are one
// contiguous range; lanes sweep that range (coalesced), products are summed per row by a segmented scan across the wave
// (entries of a row sit in consecutive lanes) and the tail lane of each segment adds into the row's LDS slot.
struct NdCsrArgs { int64_t nr; const int64_t *rows, *ptr; const int32_t *cols; const double *vals; const uint8_t *slot; const double *x; double *y; const double *scal; };
template <bool EPI>
__global__ void __launch_bounds__(256) k_csr_nd(NdCsrArgs A)
{
    __shared__ double acc[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    const int n = i0 < A.nr ? (int)(A.nr - i0 < 64 ? A.nr - i0 : 64) : 0;
    acc[wave][lane] = 0.0;
    if (n > 0) {
        const int64_t e0 = A.ptr[i0], e1 = A.ptr[i0 + n];
        for (int64_t eb = e0; eb < e1; eb += 64) {
            const int64_t e = eb + lane;
            double p = 0.0; int sg = -1;
            if (e < e1) { p = A.vals[e] * A.x[A.cols[e]]; sg = A.slot[e]; }
            for (int d = 1; d < 64; d <<= 1) { const double q = __shfl_up(p, d); const int sq = __shfl_up(sg, d); if (lane >= d && sq == sg) p += q; }
            const int snext = __shfl_down(sg, 1);
            if (sg >= 0 && (lane == 63 || snext != sg)) acc[wave][sg] += p;      // a row that continues in the next 64 entries adds again there
        }
        if (lane < n) { const int64_t r = A.rows[i0 + lane]; const double sc = EPI ? A.scal[SC_SCUR] : 1.0; A.y[r] += sc * acc[wave][lane]; }
    }
}

// Expectation values of the spin-exchange and pair-hopping operators in a state (lanc_local_energy, ED_OBSERVABLES.f90:
// 470-527: ed_Dse, ed_Dph): sum_i sgn * v_i * v_j over the same (a,b) terms as k_ndpass, with unit amplitude and the two
// kinds kept apart.  out[row*2 + kind]: partial sums per (idw,iph) row in a fixed order.
struct NdExpArgs {
    int32_t DimUp, DimDw, Norb; int64_t Qdw, dw0;
    const int32_t *mapup, *mapdw, *rankup, *rankdw;
    int32_t do_se, do_ph;
    const double *v; const double *vfull; int64_t full_rows;   // vfull: gathered vector (global rows) with several ranks, else the shard itself
    double *out;
};
__global__ void __launch_bounds__(256) k_nd_expect(NdExpArgs A)
{
    __shared__ int32_t s_jdw[2 * EDGPU_MAX_ORB * EDGPU_MAX_ORB];
    __shared__ double s_sg[2 * EDGPU_MAX_ORB * EDGPU_MAX_ORB];
    __shared__ double red[4];
    const int64_t row = blockIdx.x;
    const int iph = (int)(row / A.Qdw);
    const int64_t idwl = row - (int64_t)iph * A.Qdw, idw = A.dw0 + idwl;
    const uint32_t mdw = (uint32_t)A.mapdw[idw];
    const int nslot = 2 * A.Norb * A.Norb;
    if ((int)threadIdx.x < nslot) {
        const int kind = threadIdx.x & 1, ab = threadIdx.x >> 1, a = ab / A.Norb, b = ab - a * A.Norb;
        const uint32_t ba = 1u << a, bb = 1u << b;
        int32_t jd = -1; double sg = 0.0;
        if (a != b) {
            if (kind == 0 && A.do_se && (mdw & ba) && !(mdw & bb)) {
                const uint32_t k1 = mdw & ~ba, k2 = k1 | bb;
                jd = A.rankdw[k2]; sg = ((__popc(mdw & (ba - 1)) + __popc(k1 & (bb - 1))) & 1) ? -1.0 : 1.0;
            }
            if (kind == 1 && A.do_ph && (mdw & bb) && !(mdw & ba)) {
                const uint32_t k1 = mdw & ~bb, k2 = k1 | ba;
                jd = A.rankdw[k2]; sg = ((__popc(mdw & (bb - 1)) + __popc(k1 & (ba - 1))) & 1) ? -1.0 : 1.0;
            }
        }
        s_jdw[threadIdx.x] = jd; s_sg[threadIdx.x] = sg;
    }
    __syncthreads();
    const int DimUp = A.DimUp;
    const double *vrow = A.v + row * DimUp;
    double acc[2] = {0.0, 0.0};
    for (int iup = threadIdx.x; iup < DimUp; iup += blockDim.x) {
        const uint32_t mup = (uint32_t)A.mapup[iup];
        const double vi = vrow[iup];
        for (int a = 0; a < A.Norb; a++)
            for (int b = 0; b < A.Norb; b++) {
                const uint32_t ba = 1u << a, bb = 1u << b;
                if (a == b || !(mup & bb) || (mup & ba)) continue;
                const uint32_t k3 = mup & ~bb, k4 = k3 | ba;
                const double sgu = ((__popc(mup & (bb - 1)) + __popc(k3 & (ba - 1))) & 1) ? -1.0 : 1.0;
                const int slot = 2 * (a * A.Norb + b);
#pragma unroll
                for (int kind = 0; kind < 2; kind++) {
                    const int32_t jd = s_jdw[slot + kind];
                    if (jd < 0) continue;
                    const int64_t jrow = A.vfull == A.v ? (int64_t)iph * A.Qdw + (jd - A.dw0) : (int64_t)iph * A.full_rows + jd;
                    acc[kind] += sgu * s_sg[slot + kind] * vi * A.vfull[jrow * DimUp + A.rankup[k4]];
                }
            }
    }
#pragma unroll
    for (int kind = 0; kind < 2; kind++) { const double t = block_sum(acc[kind], red); if (threadIdx.x == 0) A.out[row * 2 + kind] = t; __syncthreads(); }
}

// ---------------------------------------------------------------------------------------------
// Distributed-transpose form of the dw hops (the reference's own MPI scheme: vector_transpose_MPI,
// ED_HAMILTONIAN_COMMON.f90:53-125, used by spMatVec_mpi_main 621-692): every rank turns its (Qdw x DimUp)
// shard into (DimUp x Qdw), an all-to-all hands rank r the rows iup in its up-block from everybody, so that
// it holds vt(iup_local, all idw); there the dw hops are a LOCAL pass along contiguous rows; the result goes
// back the same way.  Volume per rank and product: 2 x (P-1)/P shards, spread evenly over all peers -- against
// ~(log2 P)/2 shards concentrated on log2 P peers for the needed-rows exchange (better from 8 ranks on).
// ---------------------------------------------------------------------------------------------
// out[c*R + r] = in[r*C + c]   (R x C row-major -> C x R), 32x32 tiles through LDS
__global__ void __launch_bounds__(256) k_transpose2d(const double *__restrict__ in, double *__restrict__ out, int64_t R, int64_t C)
{
    __shared__ double tile[32][33];
    const int64_t tc = (C + 31) / 32, t = blockIdx.x, tr0 = (t / tc) * 32, tc0 = (t % tc) * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int y = ly; y < 32; y += 8) { const int64_t r = tr0 + y, c = tc0 + lx; if (r < R && c < C) tile[y][lx] = in[r * C + c]; }
    __syncthreads();
    for (int y = ly; y < 32; y += 8) { const int64_t c = tc0 + y, r = tr0 + lx; if (r < R && c < C) out[c * R + r] = tile[lx][y]; }
}
// received blocks (sender s: [Qup][Qdw_s] at offset Qup*dw0_s) <-> vt[Qup][DimDw]; dir 0: blocks -> vt, 1: vt -> blocks
struct TBlkArgs { int32_t P; int64_t Qup, DimDw; int64_t dw0[65]; double *blocks; double *vt; int32_t dir; };
__global__ void __launch_bounds__(256) k_tblocks(TBlkArgs A)
{
    const int64_t n = A.Qup * A.DimDw;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t il = e / A.DimDw, idw = e - il * A.DimDw;
        int sl = 0; while (sl + 1 < A.P && A.dw0[sl + 1] <= idw) sl++;            // owner of this dw column
        const int64_t q = A.dw0[sl + 1] - A.dw0[sl], b = A.Qup * A.dw0[sl] + il * q + (idw - A.dw0[sl]);
        if (A.dir == 0) A.vt[e] = A.blocks[b]; else A.blocks[b] = A.vt[e];
    }
}
// dw hops in the transposed space: hvt(row, i) = sum_k Hdw[i,k] vt(row, j_k); ELL of H_dw in [k][i] layout
struct TdwArgs { int32_t DimDw, dww, use_lds; const int32_t *ellc; const double *ellv; const double *vt; double *hvt; };
__global__ void __launch_bounds__(1024) k_tdw(TdwArgs A)
{
    extern __shared__ double lds[];
    const double *r = A.vt + (int64_t)blockIdx.x * A.DimDw;
    if (A.use_lds) { for (int i = threadIdx.x; i < A.DimDw; i += blockDim.x) lds[i] = r[i]; __syncthreads(); }
    for (int i = threadIdx.x; i < A.DimDw; i += blockDim.x) {
        double acc = 0.0;
        for (int k = 0; k < A.dww; k++) { const int32_t j = A.ellc[(int64_t)k * A.DimDw + i]; acc += A.ellv[(int64_t)k * A.DimDw + i] * (A.use_lds ? lds[j] : r[j]); }
        A.hvt[(int64_t)blockIdx.x * A.DimDw + i] = acc;
    }
}
// the same pass reading and writing the all-to-all block layout directly (rows that fit LDS): row il of vt is the
// concatenation of one contiguous segment per sender, staged into LDS in place of the k_tblocks pass; the result is
// written straight into the block layout of the way back -- two passes over the shard less
struct TdwFusedArgs { int32_t DimDw, dww, P; int64_t Qup; int64_t dw0[65]; const int32_t *ellc; const double *ellv; const double *blk_in; double *blk_out; int64_t il0; };   // il0: first row of this launch (chunked pipeline)
__global__ void __launch_bounds__(1024) k_tdw_blocks(TdwFusedArgs A)
{
    extern __shared__ double lds[];
    const int64_t il = A.il0 + blockIdx.x;
    for (int i = threadIdx.x; i < A.DimDw; i += blockDim.x) {
        int sl = 0; while (sl + 1 < A.P && A.dw0[sl + 1] <= i) sl++;
        lds[i] = A.blk_in[A.Qup * A.dw0[sl] + il * (A.dw0[sl + 1] - A.dw0[sl]) + (i - A.dw0[sl])];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < A.DimDw; i += blockDim.x) {
        double acc = 0.0;
        for (int k = 0; k < A.dww; k++) acc += A.ellv[(int64_t)k * A.DimDw + i] * lds[A.ellc[(int64_t)k * A.DimDw + i]];
        int sl = 0; while (sl + 1 < A.P && A.dw0[sl + 1] <= i) sl++;
        A.blk_out[A.Qup * A.dw0[sl] + il * (A.dw0[sl + 1] - A.dw0[sl]) + (i - A.dw0[sl])] = acc;
    }
}
// out(idw_l, iup) += [s_cur *] back(iup, idw_l)  (back: DimUp x Qdw), with the Lanczos alpha partials
template <bool EPI>
__global__ void __launch_bounds__(256) k_tadd(const double *__restrict__ back, double *__restrict__ out, const double *__restrict__ v,
                                              int64_t Qdw, int64_t DimUp, const double *__restrict__ scal, double *__restrict__ partial)
{
    __shared__ double tile[32][33];
    __shared__ double red[4];
    const int64_t tc = (Qdw + 31) / 32, t = blockIdx.x, tr0 = (t / tc) * 32, tc0 = (t % tc) * 32;     // back is DimUp x Qdw
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int y = ly; y < 32; y += 8) { const int64_t iu = tr0 + y, id = tc0 + lx; if (iu < DimUp && id < Qdw) tile[y][lx] = back[iu * Qdw + id]; }
    __syncthreads();
    const double s_cur = EPI ? scal[SC_SCUR] : 1.0;
    double asum = 0.0;
    for (int y = ly; y < 32; y += 8) {
        const int64_t id = tc0 + y, iu = tr0 + lx;
        if (iu < DimUp && id < Qdw) {
            const int64_t e = id * DimUp + iu;
            const double w = out[e] + s_cur * tile[lx][y];
            out[e] = w;
            if (EPI) asum += (s_cur * v[e]) * w;
        }
    }
    if (EPI) { double r = block_sum(asum, red); if (threadIdx.x == 0) partial[blockIdx.x] = r; }
}

// ---------------------------------------------------------------------------------------------
// Generic multi-factor operator (ed_total_ud=F: spMatVec_orbs / directMatVec_orbs,
// ED_HAMILTONIAN_SPARSE_HxV.f90:487-564, ED_HAMILTONIAN_DIRECT_HxV.f90:100-171).
// State index is the mixed-radix number over [DimUps..., DimDws...]; factor k acts on digit k.
// ---------------------------------------------------------------------------------------------
struct GenArgs {
    int32_t nfac, DimPh, Norb, direct;
    int32_t dims[2 * EDGPU_MAX_ORB];
    int64_t strides[2 * EDGPU_MAX_ORB];
    const int32_t *rowptr[2 * EDGPU_MAX_ORB];
    const int32_t *cols[2 * EDGPU_MAX_ORB];
    const double *vals[2 * EDGPU_MAX_ORB];
    const double *E[2 * EDGPU_MAX_ORB];
    const int32_t *maps[2 * EDGPU_MAX_ORB];
    const double *diag, *cross, *gup, *gdw;
    int32_t nI, total_ud, impmask;
    double econst, w0, gsum;
    int64_t Nel;               // local electronic rows (DimUp * Qdw)
    int64_t NelG, row0;        // global electronic dimension; global index of the first local row (dw split)
    const double *v; double *out;
    const double *x;           // input vector in global layout (the gathered vector with more than one rank, else v)
    const double *uprev; const double *scal; double *partial; int32_t epi;
};

__global__ void __launch_bounds__(256) k_generic(GenArgs A)
{
    __shared__ double red[8];
    const int64_t N = A.Nel * A.DimPh;
    double s_cur = 1.0, c_prev = 0.0, asum = 0.0;
    if (A.epi) { s_cur = A.scal[SC_SCUR]; c_prev = A.scal[SC_BETA] * A.scal[SC_SPREV]; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int iph = (int)(i / A.Nel);
        const int64_t iel = i - (int64_t)iph * A.Nel;
        int32_t dig[2 * EDGPU_MAX_ORB];
        { int64_t c = iel + A.row0; for (int k = 0; k < A.nfac; k++) { dig[k] = (int32_t)(c % A.dims[k]); c /= A.dims[k]; } }
        const int64_t gi = (int64_t)iph * A.NelG + iel + A.row0;   // position of this row in the global layout of x
        const double x = A.v[i];
        double dval;
        uint32_t mu = 0, md = 0;
        if (A.direct || A.DimPh > 1) {  // impurity-bit words of both spins
            const int nud = A.nfac / 2;
            if (A.total_ud) { mu = (uint32_t)A.maps[0][dig[0]] & A.impmask; md = (uint32_t)A.maps[1][dig[1]] & A.impmask; }
            else for (int a = 0; a < nud; a++) { mu |= ((uint32_t)A.maps[a][dig[a]] & 1u) << a; md |= ((uint32_t)A.maps[a + nud][dig[a + nud]] & 1u) << a; }
        }
        if (A.direct) {
            dval = A.econst + A.cross[mu * A.nI + md];
            for (int k = 0; k < A.nfac; k++) dval += A.E[k][dig[k]];
        } else dval = A.diag[iel];
        if (A.DimPh > 1) dval += A.w0 * iph;
        double acc = dval * x;
        for (int k = 0; k < A.nfac; k++) {
            const int32_t r0 = A.rowptr[k][dig[k]], r1 = A.rowptr[k][dig[k] + 1];
            for (int32_t p = r0; p < r1; p++) acc += A.vals[k][p] * A.x[gi + (int64_t)(A.cols[k][p] - dig[k]) * A.strides[k]];
        }
        if (A.DimPh > 1) {
            double g = A.gup[mu] + A.gdw[md] - A.gsum;
            if (iph + 1 < A.DimPh) acc += g * sqrt((double)(iph + 1)) * A.x[gi + A.NelG];
            if (iph > 0) acc += g * sqrt((double)iph) * A.x[gi - A.NelG];
        }
        if (A.epi) { double w = s_cur * acc - c_prev * A.uprev[i]; A.out[i] = w; asum += (s_cur * x) * w; }
        else A.out[i] = acc;
    }
    if (A.epi) { double r = block_sum(asum, red); if (threadIdx.x == 0) A.partial[blockIdx.x] = r; }
}

// ---------------------------------------------------------------------------------------------
// On-the-fly rebuild of the per-spin tables from the bit patterns (DIRECT storage): every
// matrix element of H_sigma is recomputed at each product, as directMatVec does -- but once per
// spin configuration (O(Dim_sigma) work) instead of once per state (O(N)).
// ---------------------------------------------------------------------------------------------
struct TabArgs {
    int32_t dim, nterms, width, transposed;  // transposed: ELL [width][dim] (up) else [dim][width] (dw)
    const int32_t *map, *rank;
    const int32_t *talpha, *tbeta; const double *tval;
    int32_t *ellc; double *ellv;
    uint32_t *pcols, *pcodes;   // optional packed image (k_uppass), zeroed by the caller
    uint32_t *pelem;            // optional element-major packed image (k_uppass2): 6 dwords per row, width <= 8
    uint32_t *pelem8;           // optional element-major image of 8 dwords (k_uppass_big): width <= 10
    uint32_t *pelem16;          // optional element-major image of 12 dwords (one-configuration-per-lane local passes): width <= 16
    const double *lin; int32_t nlev; int32_t impmask; double pair;  // E[i] = sum lin[l] n_l + pair * C(#imp,2)
    double *E;
};

// emit one finished row (entries sorted by column in row buffers cj/cv/ct, n valid) into every image
#define TAB_EMIT_ROW(CJ, CV, CT)                                                                                         \
    do {                                                                                                                 \
        if (A.pcols) {                                                                                                   \
            const int ncw = (A.width + 1) >> 1, ndw = (A.width + 3) >> 2;                                                \
            for (int c = 0; c < ncw; c++) {                                                                              \
                uint32_t wd_ = 0;                                                                                         \
                for (int e = 0; e < 2; e++) { const int q = 2 * c + e; wd_ |= (uint32_t)(q < n ? CJ(q) : i) << (16 * e); } \
                A.pcols[(int64_t)c * A.dim + i] = wd_;                                                                   \
            }                                                                                                            \
            for (int c = 0; c < ndw; c++) {                                                                              \
                uint32_t wd_ = 0;                                                                                         \
                for (int e = 0; e < 4; e++) { const int q = 4 * c + e; wd_ |= (uint32_t)(q < n ? CT(q) : 127u) << (8 * e); } \
                A.pcodes[(int64_t)c * A.dim + i] = wd_;                                                                  \
            }                                                                                                            \
        }                                                                                                                \
        if (A.pelem8) {                                                                                                  \
            for (int c = 0; c < 5; c++) {                                                                                \
                uint32_t wd_ = 0;                                                                                         \
                for (int e = 0; e < 2; e++) { const int q = 2 * c + e; wd_ |= (uint32_t)(q < n ? CJ(q) : i) << (16 * e); } \
                A.pelem8[(int64_t)8 * i + c] = wd_;                                                                      \
            }                                                                                                            \
            for (int c = 0; c < 3; c++) {                                                                                \
                uint32_t wd_ = 0;                                                                                         \
                for (int e = 0; e < 4; e++) { const int q = 4 * c + e; wd_ |= (uint32_t)(q < n ? CT(q) : 127u) << (8 * e); } \
                A.pelem8[(int64_t)8 * i + 5 + c] = wd_;                                                                  \
            }                                                                                                            \
        }                                                                                                                \
        if (A.pelem16) {                                                                                                 \
            for (int c = 0; c < 8; c++) {                                                                                \
                uint32_t wd_ = 0;                                                                                         \
                for (int e = 0; e < 2; e++) { const int q = 2 * c + e; wd_ |= (uint32_t)(q < n ? CJ(q) : i) << (16 * e); } \
                A.pelem16[(int64_t)12 * i + c] = wd_;                                                                    \
            }                                                                                                            \
            for (int c = 0; c < 4; c++) {                                                                                \
                uint32_t wd_ = 0;                                                                                         \
                for (int e = 0; e < 4; e++) { const int q = 4 * c + e; wd_ |= (uint32_t)(q < n ? CT(q) : 127u) << (8 * e); } \
                A.pelem16[(int64_t)12 * i + 8 + c] = wd_;                                                                \
            }                                                                                                            \
        }                                                                                                                \
        if (A.pelem) {                                                                                                   \
            for (int c = 0; c < 4; c++) {                                                                                \
                uint32_t wd_ = 0;                                                                                         \
                for (int e = 0; e < 2; e++) { const int q = 2 * c + e; wd_ |= (uint32_t)(q < n ? CJ(q) : i) << (16 * e); } \
                A.pelem[(int64_t)6 * i + c] = wd_;                                                                       \
            }                                                                                                            \
            for (int c = 0; c < 2; c++) {                                                                                \
                uint32_t wd_ = 0;                                                                                         \
                for (int e = 0; e < 4; e++) { const int q = 4 * c + e; wd_ |= (uint32_t)(q < n ? CT(q) : 127u) << (8 * e); } \
                A.pelem[(int64_t)6 * i + 4 + c] = wd_;                                                                   \
            }                                                                                                            \
        }                                                                                                                \
        for (int k = 0; k < A.width; k++) {                                                                              \
            int32_t j = k < n ? CJ(k) : i; double val = k < n ? CV(k) : 0.0;                                             \
            int64_t o = A.transposed ? (int64_t)k * A.dim + i : (int64_t)i * A.width + k;                                \
            A.ellc[o] = j; A.ellv[o] = val;                                                                              \
        }                                                                                                                \
        double e = 0.0;                                                                                                  \
        for (int l = 0; l < A.nlev; l++) if (mi & (1u << l)) e += A.lin[l];                                              \
        int ni = __popc(mi & A.impmask);                                                                                 \
        e += A.pair * (double)(ni * (ni - 1) / 2);                                                                       \
        A.E[i] = e;                                                                                                      \
    } while (0)

// General form: one lane per row, terms walked serially (any number of terms, duplicate (alpha,beta) pairs summed in
// term order).  The row buffers live in LDS, [entry][lane]: as private arrays they are dynamically indexed, i.e.
// scratch memory, and every dispatch of a scratch-using kernel pays a set-up of tens of microseconds.
template <int MAXW>
__global__ void __launch_bounds__(64) k_direct_tables(TabArgs A)
{
    __shared__ int32_t cj_s[MAXW][64]; __shared__ double cv_s[MAXW][64]; __shared__ uint8_t ct_s[MAXW][64];
    const int lane = threadIdx.x;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.dim) return;
    const uint32_t mi = (uint32_t)A.map[i];
    int n = 0;
#define cj(q) cj_s[q][lane]
#define cv(q) cv_s[q][lane]
#define ct(q) ct_s[q][lane]
    for (int t = 0; t < A.nterms; t++) {
        const uint32_t ba = 1u << (A.talpha[t] - 1), bb = 1u << (A.tbeta[t] - 1);
        if (!(mi & ba) || (mi & bb)) continue;
        const uint32_t k1 = mi & ~ba, mj = k1 | bb;
        int sg = __popc(mj & (bb - 1)) + __popc(k1 & (ba - 1));   // c_beta on source, c^+_alpha on k1
        double val = (sg & 1) ? -A.tval[t] : A.tval[t];
        int32_t j = A.rank[mj];
        int p = n;
        bool dup = false;
        for (int q = 0; q < n; q++) if (cj(q) == j) { cv(q) += val; dup = true; break; }
        if (dup || n >= MAXW) continue;
        while (p > 0 && cj(p - 1) > j) { cj(p) = cj(p - 1); cv(p) = cv(p - 1); ct(p) = ct(p - 1); p--; }   // keep cols ascending
        cj(p) = j; cv(p) = val; ct(p) = (uint8_t)(t | ((sg & 1) ^ (A.tval[t] < 0.0 ? 1 : 0) ? 0x80 : 0)); n++;
    }
    TAB_EMIT_ROW(cj, cv, ct);
#undef cj
#undef cv
#undef ct
}

// Fast form (<= 64 terms with distinct (alpha,beta) pairs -- every 1-orbital model and most others): one WAVE per
// row, one lane per term.  The serial form is a chain of dependent look-ups (term -> bit test -> rank table ->
// sorted insert) and takes 25-35 us whatever the factor size; here each lane tests its term, the slot of a
// surviving hop is the number of surviving hops with a smaller target pattern (columns ascend with the pattern),
// found with 'nterms' readlane+compare steps, and lane 0 emits the row.
__global__ void __launch_bounds__(256) k_direct_tables_wave(TabArgs A)
{
    __shared__ int32_t cj_s[4][64]; __shared__ double cv_s[4][64]; __shared__ uint8_t ct_s[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + w;
    const bool row = i < A.dim;
    const uint32_t mi = row ? (uint32_t)A.map[i] : 0u;
    const bool in = row && lane < A.nterms;
    const int ta = in ? A.talpha[lane] : 1, tb = in ? A.tbeta[lane] : 1;
    const double tv = in ? A.tval[lane] : 0.0;
    const uint32_t ba = 1u << (ta - 1), bb = 1u << (tb - 1);
    const bool ok = in && (mi & ba) && !(mi & bb);
    const uint32_t k1 = mi & ~ba, mj = k1 | bb;
    const int sg = __popc(mj & (bb - 1)) + __popc(k1 & (ba - 1));   // c_beta on source, c^+_alpha on k1
    const int32_t j = row ? A.rank[ok ? mj : mi] : 0;
    const uint32_t key = ok ? mj : 0xffffffffu;
    int p = 0;
    for (int u = 0; u < A.nterms; u++) p += (__shfl(key, u) < key) ? 1 : 0;
    const int n = __popcll(__ballot(ok));
    if (ok) {
        cj_s[w][p] = j; cv_s[w][p] = (sg & 1) ? -tv : tv;
        ct_s[w][p] = (uint8_t)(lane | ((sg & 1) ^ (tv < 0.0 ? 1 : 0) ? 0x80 : 0));
    }
    __syncthreads();
    if (!row || lane != 0) return;
#define cj(q) cj_s[w][q]
#define cv(q) cv_s[w][q]
#define ct(q) ct_s[w][q]
    TAB_EMIT_ROW(cj, cv, ct);
#undef cj
#undef cv
#undef ct
}
#undef TAB_EMIT_ROW

// ---------------------------------------------------------------------------------------------
// Explicit CSR (storage=CSR): expansion of the Kronecker sum, then a row-block streaming SpMV.
// ---------------------------------------------------------------------------------------------
// rows a peer needs, copied out of the shard into one contiguous message per peer
struct PackArgs { int32_t DimUp, DimPh; int64_t Qdw, nrows; const int32_t *list; const double *v; double *out; int64_t ld; };      // ld: row pitch of v and of the packed rows (DimUp, or the padded pitch of the explicit CSR)
// One workgroup per packed row (grid-stride over rows): 16-byte accesses where the pitch keeps rows 16-byte aligned, no index
// arithmetic per element.  (The first form ran one 64-bit division per element over a flat index: 2.9 ms for the 3.9 GB a rank of
// eight packs at Ns=18, 0.99 ms for 266 MB at Ns=16 -- on the critical path in front of every exchange; profiles/r04/rank_of_8_*.txt.)
__global__ void __launch_bounds__(256) k_pack_rows(PackArgs A)
{
    const int64_t nr = A.nrows * A.DimPh;
    const bool vec = (A.ld & 1) == 0 && (A.DimUp & 1) == 0 && ((reinterpret_cast<uintptr_t>(A.v) | reinterpret_cast<uintptr_t>(A.out)) & 15) == 0;      // every row start is 16-byte aligned on both sides
    for (int64_t r = blockIdx.x; r < nr; r += gridDim.x) {
        const int64_t iph = r / A.nrows, q = r - iph * A.nrows;
        const double *src = A.v + (iph * A.Qdw + A.list[q]) * A.ld; double *dst = A.out + r * A.ld;      // padding columns of the message rows stay zero (buffer cleared once)
        if (vec) { for (int i = threadIdx.x; i < (A.DimUp >> 1); i += 256) reinterpret_cast<dbl2v *>(dst)[i] = reinterpret_cast<const dbl2v *>(src)[i]; }
        else for (int i = threadIdx.x; i < A.DimUp; i += 256) dst[i] = src[i];
    }
}

struct CsrBuildArgs {
    int32_t DimUp, DimDw; int64_t Qdw, dw0, rowshift;
    const int32_t *uprp, *upc; const double *upv;   // CSR of H_up (rowptr32)
    const int32_t *dwrp, *dwc; const double *dwv;   // CSR of H_dw
    const int64_t *uppre;                           // prefix over iup of (1 + nnz_up(iup)), [DimUp+1]
    const int64_t *dwpre;                           // prefix over local idw of row-block nnz, [Qdw+1]
    const double *diag;
    int64_t *rowptr; int32_t *cols; double *vals;
    int32_t sel; const int64_t *selpre;   // part 2 only: 0 all dw hops; 1 those into the rank's own rows; 2 those into received (halo)
                                          // rows.  selpre[idwl] = entries of the selected kind in the rows before idwl (times DimUp)
    int64_t ld;                           // row pitch of the vectors the columns index (DimUp, or DimUp rounded up to the 128-B line)
    int32_t tm; const int64_t *cpre;      // tm (dw-type blocks, with il): blocks stored tile-major, block (jb, idw) at jb*rpb*cpre[Qdw] + rows(jb)*cpre[idw]
    int32_t il, rpb;                      // il: entries of each block of `rpb` consecutive rows of one idw row are stored lane-major
                                          // (entry e of the block's row lr at block_start + e*rows_in_block + lr when the rows of the
                                          // block have equal lengths; level by level otherwise, see `jag` below).  rowptr keeps the
                                          // row-major numbers (block start, row lengths).
};

// row (iup, idw) holds, in ascending column order: dw hops to jdw<idw, up hops jup<iup, the
// diagonal, up hops jup>iup, dw hops jdw>idw.  Columns index the gathered vector (global idw).
// part 0: the whole row; part 1: only diagonal + up hops ("local" block); part 2: only dw hops.
__global__ void __launch_bounds__(256) k_csr_build(CsrBuildArgs A, int part)
{
    const int64_t il = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nloc = A.Qdw * A.DimUp;
    if (il > nloc) return;
    const int64_t upall = A.uppre[A.DimUp];
    if (il == nloc) {
        A.rowptr[nloc] = part == 0 ? A.dwpre[A.Qdw] : (part == 1 ? A.Qdw * upall : (A.sel ? A.selpre[A.Qdw] : A.dwpre[A.Qdw] - A.Qdw * upall));
        return;
    }
    const int64_t idwl = il / A.DimUp; const int iup = (int)(il - idwl * A.DimUp); const int64_t idw = A.dw0 + idwl;
    const int32_t d0 = A.dwrp[idw], d1 = A.dwrp[idw + 1], u0 = A.uprp[iup], u1 = A.uprp[iup + 1];
    const int ndw = d1 - d0;
    auto keep = [&](int32_t xr) { return A.sel == 0 || (A.sel == 1) == (xr < A.Qdw); };   // x-space rows [0,Qdw) are the rank's own
    int nsel = 0;
    if (part == 2 && A.sel) for (int32_t q = d0; q < d1; q++) nsel += keep(A.dwc[q]) ? 1 : 0;
    auto rowstart = [&](int ix) -> int64_t {
        if (part == 0) return A.dwpre[idwl] + (int64_t)ix * ndw + A.uppre[ix];
        if (part == 1) return idwl * upall + A.uppre[ix];
        if (A.sel == 0) return A.dwpre[idwl] - idwl * upall + (int64_t)ix * ndw;
        return A.selpre[idwl] + (int64_t)ix * nsel;
    };
    const int64_t p = rowstart(iup);
    A.rowptr[il] = p;
    // position of the row's e-th entry: row-major p + e, or lane-major inside the row block
    int64_t q0 = p, qs = 1;
    if (A.il) {
        const int iup0 = (iup / A.rpb) * A.rpb; int nr = A.DimUp - iup0; if (nr > A.rpb) nr = A.rpb;
        q0 = (A.tm && part == 2 ? (int64_t)iup0 * A.cpre[A.Qdw] + (int64_t)nr * A.cpre[idwl] : rowstart(iup0)) + (iup - iup0); qs = nr;
    }
    int e = 0;
    // local block (part 1) in lane-major order with rows of DIFFERENT lengths inside a 64-row block: the rows that own an
    // e-th entry are packed in row order, level after level: position = block_start + sum_i min(len_i, e) + #{i < r : len_i > e}
    const bool jag = A.il && part == 1;
    int jag0 = 0, jagn = 0; int64_t jlev = 0;
    if (jag) { jag0 = (iup / A.rpb) * A.rpb; jagn = A.DimUp - jag0; if (jagn > A.rpb) jagn = A.rpb; jlev = rowstart(jag0); }
    auto emit = [&](int32_t c, double v) {
        int64_t q;
        if (jag) {
            int cnt = 0, before = 0;
            for (int i = 0; i < jagn; i++) { const bool has = (int)(A.uppre[jag0 + i + 1] - A.uppre[jag0 + i]) > e; cnt += has; before += has && (jag0 + i < iup); }
            q = jlev + before; jlev += cnt;
        } else q = q0 + (int64_t)e * qs;
        A.cols[q] = c; A.vals[q] = v; e++;
    };
    const int64_t xrow = idw - A.rowshift;     // index of this dw row in the column (x) space
    if (part != 1) for (int32_t q = d0; q < d1; q++) if (A.dwc[q] < xrow && keep(A.dwc[q])) emit((int32_t)((int64_t)A.dwc[q] * A.ld + iup), A.dwv[q]);
    if (part != 2) {
        int32_t r = u0;
        for (; r < u1 && A.upc[r] < iup; r++) emit((int32_t)(xrow * A.ld + A.upc[r]), A.upv[r]);
        emit((int32_t)(xrow * A.ld + iup), A.diag[il]);
        for (; r < u1; r++) emit((int32_t)(xrow * A.ld + A.upc[r]), A.upv[r]);
    }
    if (part != 1) for (int32_t q = d0; q < d1; q++) if (A.dwc[q] >= xrow && keep(A.dwc[q])) emit((int32_t)((int64_t)A.dwc[q] * A.ld + iup), A.dwv[q]);
}

struct CsrArgs {
    int64_t nrows; int32_t rows_per_block; int32_t DimUp; int32_t blocks_per_row; int64_t Qdw; int32_t tile;
    const int64_t *rowptr; const int32_t *cols; const double *vals;
    const double *x; double *y;
    const double *v_local;  // x restricted to the local rows (== x on one GPU), for the alpha dot
    const double *uprev; const double *scal; double *partial; int32_t epi; int32_t xcd_rows;
    int32_t pass;      // 0: whole product; 1: first block of a split CSR (no alpha); 2: last block (y +=, alpha);
                       // 3: a block in between (y +=, no alpha)
    int32_t coltiles;  // 1: blocks sweep XCD-affine column tiles (idw fastest) -- the dw block of a split CSR
    // dw-type blocks: every row of one idw row has the same length, rowptr(iup, idw) = rowbase[idw] + iup * rowcnt[idw];
    // when set the row pointer array is not read at all
    const int64_t *rowbase; const int32_t *rowcnt;
    int64_t ld;             // row pitch of x / y / uprev / v_local in doubles (the stored columns index that space); rowptr stays dense
    const int64_t *cpre;    // lane-major dw-type blocks stored tile-major (k_csr_il only): block (jb, idw) starts at jb*64*cpre[Qdw] + rows(jb)*cpre[idw]
};

// Row-block CSR product.  A workgroup owns `rows_per_block` consecutive rows of one idw row.  Their
// contiguous (value, column) range is copied to LDS with wide coalesced loads (16 B per lane, all
// issued before the first use), then lane r walks row r out of LDS.  For a fixed entry slot k the
// lanes of a wave gather x at consecutive addresses for the dw hops and the diagonal (9 of 17 entries
// at half filling), and no product array / segmented sum is needed.
typedef double dbl2_t __attribute__((ext_vector_type(2)));
typedef int int4_t __attribute__((ext_vector_type(4)));

// blockIdx -> (local dw row, row block inside it) for the two sweep orders; false: idle block
__device__ __forceinline__ bool csr_block_map(const CsrArgs &A, int64_t &idwl, int &jb)
{
    if (A.coltiles) {
        // column tile jb is owned by XCD (jb % 8); that XCD walks all idw rows of the tile, so the
        // x slab x(tile, all idw) that the dw hops gather from stays in its L2
        // (32-bit arithmetic: blockIdx and the number of local dw rows both fit, and a 64-bit division costs hundreds of instructions per block)
        uint32_t ch; xcd_tile_map(blockIdx.x, (uint32_t)A.blocks_per_row, (uint32_t)A.Qdw, jb, ch); idwl = ch;
        return true;
    }
    const int64_t per_tile = (int64_t)A.tile * A.Qdw;
    const int nt_full = A.blocks_per_row / A.tile;
    if (!A.xcd_rows && per_tile < ((int64_t)1 << 31)) {      // the common case in 32-bit arithmetic
        const uint32_t b = blockIdx.x, pt = (uint32_t)per_tile;
        uint32_t t, r, tw;
        if (b < (uint32_t)nt_full * pt) { t = b / pt; r = b - t * pt; tw = (uint32_t)A.tile; }
        else { t = (uint32_t)nt_full; r = b - (uint32_t)nt_full * pt; tw = (uint32_t)(A.blocks_per_row - nt_full * A.tile); }
        const uint32_t q = r / tw; idwl = q; jb = (int)(t * (uint32_t)A.tile + (r - q * tw));
        return true;
    }
    const int64_t b = blockIdx.x;
    int64_t t, r; int tw;
    if (b < (int64_t)nt_full * per_tile) { t = b / per_tile; r = b - t * per_tile; tw = A.tile; }
    else { t = nt_full; r = b - (int64_t)nt_full * per_tile; tw = A.blocks_per_row - nt_full * A.tile; }
    if (A.xcd_rows) {
        // workgroups are dealt round-robin over the 8 XCDs: keep all row blocks of one idw row on
        // one XCD (its L2 then serves that row's up-hop gathers).  Speed only, never correctness.
        const int64_t q8 = A.Qdw & ~(int64_t)7, body = q8 * tw;
        if (r < body) { const int64_t g = r / (8 * (int64_t)tw), w = r - g * 8 * tw; idwl = g * 8 + (w & 7); jb = (int)(t * A.tile + (w >> 3)); }
        else { const int64_t w = r - body; idwl = q8 + w / tw; jb = (int)(t * A.tile + (w - (w / tw) * tw)); }
    } else { idwl = r / tw; jb = (int)(t * A.tile + (r - idwl * tw)); }
    return true;
}

// Row-block product on the LANE-MAJOR image of a block (csr_il): every row of the 64-row block has the same length, entry e
// of the block's row lr sits at block_start + e*rows + lr, so lane lr reads its own row with coalesced loads (512 B of
// values, 256 B of columns per instruction and wave) -- no LDS staging, no transposition, and every load of the block is in
// flight before the first gather.  Occupancy is bounded by registers only, which is what keeps enough of the stream in
// flight per CU.  The local block takes block start and row length from the streamed row pointer; dw-type blocks from the
// factored form.
template <int C, bool NT>
__device__ __forceinline__ double csr_il_row(const double *__restrict__ pv, const int32_t *__restrict__ pc, int nr, const double *__restrict__ x)
{
    double v[C]; int32_t c[C];
#pragma unroll
    for (int e = 0; e < C; e++) {
        if (NT) { v[e] = __builtin_nontemporal_load(pv + (int64_t)e * nr); c[e] = __builtin_nontemporal_load(pc + (int64_t)e * nr); }
        else { v[e] = pv[(int64_t)e * nr]; c[e] = pc[(int64_t)e * nr]; }
    }
    double xv[C];
#pragma unroll
    for (int e = 0; e < C; e++) xv[e] = x[c[e]];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int e = 0; e < C; e++) { if (e & 1) a1 += v[e] * xv[e]; else a0 += v[e] * xv[e]; }
    return a0 + a1;
}

template <bool NT, int BLK /* 0: local block, 1: a dw-type block -- keeps the two apart in kernel traces */>
__global__ void __launch_bounds__(64) k_csr_il(CsrArgs A)
{
    __shared__ double red[4];
    int64_t idwl; int jb;
    if (!csr_block_map(A, idwl, jb)) { if (A.epi && (A.pass == 0 || A.pass == 2) && threadIdx.x == 0) A.partial[blockIdx.x] = 0.0; return; }
    const int tid = threadIdx.x;
    const int iup0 = jb * A.rows_per_block;
    int nr = A.DimUp - iup0; if (nr > A.rows_per_block) nr = A.rows_per_block;
    const int64_t row0 = idwl * A.DimUp + iup0, vrow0 = idwl * A.ld + iup0;      // row pointer index / vector index
    const bool act = tid < nr;
    int64_t p0; int cnt, len = 0;
    if (A.cpre) { cnt = A.rowcnt[idwl]; p0 = (int64_t)iup0 * A.cpre[A.Qdw] + (int64_t)nr * A.cpre[idwl]; }
    else if (A.rowbase) { cnt = A.rowcnt[idwl]; p0 = A.rowbase[idwl] + (int64_t)iup0 * cnt; }
    else {
        // streamed row pointer: lane r holds rowptr[row0 + r] and the next one; block start = first lane's, length = difference
        const int64_t ra = A.rowptr[row0 + (act ? tid : 0)], re = A.rowptr[row0 + (act ? tid : 0) + 1];
        p0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(ra >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(ra & 0xffffffffll));
        cnt = __builtin_amdgcn_readfirstlane((int)(re - ra)); len = act ? (int)(re - ra) : 0;
    }
    double up_i = 0.0, xl_i = 0.0;
    if (act) {
        if (A.epi && A.pass < 2) up_i = A.uprev[vrow0 + tid];
        if (A.epi && (A.pass == 0 || A.pass == 2)) xl_i = A.v_local[vrow0 + tid];
        if (A.pass >= 2) up_i = A.y[vrow0 + tid];        // running value of the earlier blocks
    }
    double s_cur = 1.0, c_prev = 0.0, asum = 0.0;
    if (A.epi) { s_cur = A.scal[SC_SCUR]; c_prev = A.scal[SC_BETA] * A.scal[SC_SPREV]; }
    double accj = 0.0; bool jag = false;
    if (BLK == 0 && !A.rowbase) {
        // rows of different lengths inside the block (up configurations with different numbers of hops): level e holds the
        // rows with more than e entries, packed in row order -- positions from a ballot, two levels in flight per step
        jag = __ballot(act && len != cnt) != 0ull;
        if (jag) {
            const uint64_t lt = (tid == 0) ? 0ull : (~0ull >> (64 - tid));
            int64_t base = p0;
            for (int e = 0;; e += 2) {
                const bool h0 = e < len, h1 = e + 1 < len;
                const uint64_t m0 = __ballot(h0), m1 = __ballot(h1);
                if (!m0) break;
                const int n0 = __popcll(m0);
                double v0 = 0.0, v1 = 0.0, x0 = 0.0, x1 = 0.0; int32_t c0 = 0, c1 = 0;
                if (h0) { const int64_t q = base + __popcll(m0 & lt); v0 = A.vals[q]; c0 = A.cols[q]; }
                if (h1) { const int64_t q = base + n0 + __popcll(m1 & lt); v1 = A.vals[q]; c1 = A.cols[q]; }
                if (h0) x0 = A.x[c0];
                if (h1) x1 = A.x[c1];
                accj += v0 * x0; accj += v1 * x1;
                base += n0 + __popcll(m1);
            }
        }
    }
    if (act) {
        const double *pv = A.vals + p0 + tid; const int32_t *pc = A.cols + p0 + tid;
        double acc;
        if (jag) acc = accj;
        else switch (cnt) {
        case 9: acc = csr_il_row<9, NT>(pv, pc, nr, A.x); break;
        case 8: acc = csr_il_row<8, NT>(pv, pc, nr, A.x); break;
        case 7: acc = csr_il_row<7, NT>(pv, pc, nr, A.x); break;
        case 6: acc = csr_il_row<6, NT>(pv, pc, nr, A.x); break;
        case 5: acc = csr_il_row<5, NT>(pv, pc, nr, A.x); break;
        default: {
            acc = 0.0; int e = 0;
            for (; e + 4 <= cnt; e += 4) acc += csr_il_row<4, NT>(pv + (int64_t)e * nr, pc + (int64_t)e * nr, nr, A.x);
            for (; e < cnt; e++) acc += csr_il_row<1, NT>(pv + (int64_t)e * nr, pc + (int64_t)e * nr, nr, A.x);
        }
        }
        const int64_t row = vrow0 + tid;
        if (A.epi) {
            const double w = A.pass >= 2 ? up_i + s_cur * acc : s_cur * acc - c_prev * up_i;
            A.y[row] = w; asum = (s_cur * xl_i) * w;
        } else A.y[row] = A.pass >= 2 ? up_i + acc : acc;
    }
    if (A.epi && (A.pass == 0 || A.pass == 2)) { double r = block_sum(asum, red); if (tid == 0) A.partial[blockIdx.x] = r; }
}

template <int ROWS, int CAPN, bool NT>
__global__ void __launch_bounds__(ROWS) k_csr_spmv2(CsrArgs A)
{
    // LDS image of the row block's (value, column) range, shifted by (p0 & 3) so that 4-entry
    // groups are 32-byte aligned both in HBM and in LDS (wide loads, wide conflict-free stores)
    __shared__ __attribute__((aligned(32))) double sval[CAPN + 4];
    __shared__ __attribute__((aligned(16))) int32_t scol[CAPN + 4];
    __shared__ double red[4];
    int64_t idwl; int jb;
    if (!csr_block_map(A, idwl, jb)) { if (A.epi && (A.pass == 0 || A.pass == 2) && threadIdx.x == 0) A.partial[blockIdx.x] = 0.0; return; }
    const int tid = threadIdx.x;
    const int iup0 = jb * A.rows_per_block;
    int nr = A.DimUp - iup0; if (nr > A.rows_per_block) nr = A.rows_per_block;
    const int64_t row0 = idwl * A.DimUp + iup0, vrow0 = idwl * A.ld + iup0;      // row pointer index / vector index
    int64_t p0; int n, ucnt = 0;
    if (A.rowbase) { ucnt = A.rowcnt[idwl]; p0 = A.rowbase[idwl] + (int64_t)iup0 * ucnt; n = nr * ucnt; }
    else { p0 = A.rowptr[row0]; n = (int)(A.rowptr[row0 + nr] - p0); }
    const int sh = (int)(p0 & 3);
    int ra = 0, re = 0;
    double up_i = 0.0, xl_i = 0.0;   // epilogue operands, requested before the stream so their latency hides under it
    if (tid < nr) {
        if (A.rowbase) { ra = tid * ucnt + sh; re = ra + ucnt; }
        else { ra = (int)(A.rowptr[row0 + tid] - p0) + sh; re = (int)(A.rowptr[row0 + tid + 1] - p0) + sh; }
        if (A.epi && A.pass < 2) up_i = A.uprev[vrow0 + tid];
        if (A.epi && (A.pass == 0 || A.pass == 2)) xl_i = A.v_local[vrow0 + tid];
        if (A.pass >= 2) up_i = A.y[vrow0 + tid];        // running value of the earlier blocks
    }
    {
        const double *gv = A.vals + (p0 - sh); const int32_t *gc = A.cols + (p0 - sh);   // 4-aligned base
        const int end = sh + n;                       // entries [sh, end) are ours
        const int g0 = sh ? 1 : 0, g1 = end >> 2;     // full 4-groups [g0, g1)
        if (sh && tid >= sh && tid < 4 && tid < end) { sval[tid] = gv[tid]; scol[tid] = gc[tid]; }
        const dbl2_t *pv = reinterpret_cast<const dbl2_t *>(gv);
        const int4_t *pc = reinterpret_cast<const int4_t *>(gc);
        dbl2_t *sv2 = reinterpret_cast<dbl2_t *>(sval);
        int4_t *sc4 = reinterpret_cast<int4_t *>(scol);
        for (int g = g0 + tid; g < g1; g += ROWS) {
            dbl2_t a, b; int4_t c;
            if (NT) { a = __builtin_nontemporal_load(pv + 2 * g); b = __builtin_nontemporal_load(pv + 2 * g + 1); c = __builtin_nontemporal_load(pc + g); }
            else { a = pv[2 * g]; b = pv[2 * g + 1]; c = pc[g]; }
            sv2[2 * g] = a; sv2[2 * g + 1] = b; sc4[g] = c;
        }
        int ts = g1 << 2; if (sh && ts < 4) ts = 4;        // tail: entries [ts, end) past the last full group
        const int tt = ts + tid;
        if (tt < end) { sval[tt] = gv[tt]; scol[tt] = gc[tt]; }
    }
    __syncthreads();
    double s_cur = 1.0, c_prev = 0.0, asum = 0.0;
    if (A.epi) { s_cur = A.scal[SC_SCUR]; c_prev = A.scal[SC_BETA] * A.scal[SC_SPREV]; }
    if (tid < nr) {
        const int64_t row = vrow0 + tid;
        double acc0 = 0.0, acc1 = 0.0;
        int k = ra;
        for (; k + 4 <= re; k += 4) {
            const double x0 = A.x[scol[k]], x1 = A.x[scol[k + 1]], x2 = A.x[scol[k + 2]], x3 = A.x[scol[k + 3]];
            acc0 += sval[k] * x0; acc1 += sval[k + 1] * x1; acc0 += sval[k + 2] * x2; acc1 += sval[k + 3] * x3;
        }
        for (; k < re; k++) acc0 += sval[k] * A.x[scol[k]];
        const double acc = acc0 + acc1;
        if (A.epi) {
            const double w = A.pass >= 2 ? up_i + s_cur * acc : s_cur * acc - c_prev * up_i;
            A.y[row] = w; asum = (s_cur * xl_i) * w;
        } else A.y[row] = A.pass >= 2 ? up_i + acc : acc;
    }
    if (A.epi && (A.pass == 0 || A.pass == 2)) { double r = block_sum(asum, red); if (tid == 0) A.partial[blockIdx.x] = r; }
}

// ---------------------------------------------------------------------------------------------
// |out> = c^+_{orb,spin} |in>  or  c_{orb,spin} |in>  between two sectors, gather form over the
// target index (ED_GF_NORMAL.f90:190-213 and 264-283 loop over the source and scatter).
// The sign involves only the spin string that changes (cdg/c of ED_SETUP.f90:805-831).
// ---------------------------------------------------------------------------------------------
struct OpArgs {
    int32_t nfac, fb, bitpos, dagger, DimPh;        // fb: factor (digit) that changes; bitpos: 0-based level
    int32_t dims_to[2 * EDGPU_MAX_ORB], dims_from[2 * EDGPU_MAX_ORB];
    const int32_t *map_to_fb, *rank_from_fb;
    int64_t nel_to, nel_from;                        // rows of the local target shard; phonon-block stride of vin
    int64_t row0_to, src_off;                        // global index of the first target row; global -> vin index shift
    const double *vin; double *vout;
};

__global__ void __launch_bounds__(256) k_apply_op(OpArgs A)
{
    const int64_t N = A.nel_to * A.DimPh;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t iph = j / A.nel_to; int64_t c = j - iph * A.nel_to + A.row0_to;
        int64_t isrc = -A.src_off, stride = 1; double val = 0.0; bool ok = true; double sgn = 1.0;
        for (int k = 0; k < A.nfac; k++) {
            int32_t dg = (int32_t)(c % A.dims_to[k]); c /= A.dims_to[k];
            if (k == A.fb) {
                const uint32_t r = (uint32_t)A.map_to_fb[dg], bit = 1u << A.bitpos;
                if (A.dagger ? !(r & bit) : (r & bit)) ok = false;
                const uint32_t m = A.dagger ? (r & ~bit) : (r | bit);
                sgn = (__popc(m & (bit - 1u)) & 1) ? -1.0 : 1.0;
                dg = ok ? A.rank_from_fb[m] : 0;
            }
            isrc += (int64_t)dg * stride; stride *= A.dims_from[k];
        }
        if (ok) val = sgn * A.vin[isrc + iph * A.nel_from];
        A.vout[j] = val;
    }
}

// ---------------------------------------------------------------------------------------------
// Generic device CSR product for the sparse-matrix container (edgpu_sp_*): y (+)= A x.
// Short rows: one lane per row; long rows (mean nnz > 32): one wave per row with a shuffle reduction.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_spgen_lane(int64_t nrow, const int64_t *__restrict__ rp, const int32_t *__restrict__ cols,
                                                    const double *__restrict__ vals, const double *__restrict__ x, double *__restrict__ y, int accumulate)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (int64_t)gridDim.x * blockDim.x) {
        double acc = accumulate ? y[r] : 0.0;
        for (int64_t p = rp[r]; p < rp[r + 1]; p++) acc += vals[p] * x[cols[p]];
        y[r] = acc;
    }
}
__global__ void __launch_bounds__(256) k_spgen_wave(int64_t nrow, const int64_t *__restrict__ rp, const int32_t *__restrict__ cols,
                                                    const double *__restrict__ vals, const double *__restrict__ x, double *__restrict__ y, int accumulate)
{
    const int lane = threadIdx.x & 63;
    for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < nrow; r += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        double acc = 0.0;
        for (int64_t p = rp[r] + lane; p < rp[r + 1]; p += 64) acc += vals[p] * x[cols[p]];
        acc = wave_sum(acc);
        if (lane == 0) y[r] = accumulate ? y[r] + acc : acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Lanczos vector kernels (SciFortran lanczos_iteration semantics, driver in edgpu.hip)
// ---------------------------------------------------------------------------------------------
// out[0] = sum(in[0..n)) in a fixed order; single workgroup
__global__ void __launch_bounds__(1024) k_reduce_final(const double *__restrict__ in, int64_t n, double *__restrict__ out)
{
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    double r = block_sum(s, red);
    if (threadIdx.x == 0) *out = r;
}

// stage 1 of a large reduction: block b sums the contiguous chunk [b*chunk, (b+1)*chunk) in a
// fixed order (deterministic for a given n), so the single-workgroup final pass stays short
__global__ void __launch_bounds__(256) k_reduce_chunks(const double *__restrict__ in, int64_t n, int64_t chunk, double *__restrict__ out)
{
    __shared__ double red[4];
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    double s = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) s += in[i];
    double r = block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// partial[b] = sum over the block's grid-stride elements of a[i]*b[i]
__global__ void __launch_bounds__(256) k_dot_partial(const double *__restrict__ a, const double *__restrict__ b, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += a[i] * b[i];
    double r = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// w -= alpha*(s_cur*u) ; partial beta^2 ; optionally acc += zk*s_cur*u (eigenvector pass)
// two elements per lane (16-byte accesses); n2 = n/2 pairs, the odd tail element handled by thread 0 of block 0
__global__ void __launch_bounds__(256) k_axpy_norm(double *__restrict__ w, const double *__restrict__ u, double *__restrict__ acc,
                                                   const double *__restrict__ scal, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[4];
    const double s_cur = scal[SC_SCUR], alpha = scal[SC_ALPHA], zk = scal[SC_ZK];
    double s = 0.0;
    const int64_t n2 = n >> 1;
    dbl2v *w2 = reinterpret_cast<dbl2v *>(w); const dbl2v *u2 = reinterpret_cast<const dbl2v *>(u); dbl2v *a2 = reinterpret_cast<dbl2v *>(acc);
    // Every workgroup owns ONE contiguous piece of the vectors (a grid-stride walk puts consecutive accesses of a lane
    // gridDim*4 KB apart: 2 MB at 512 workgroups, a new page each time on the 19 GB vectors of Ns=18, where the pass ran at
    // 4.0 TB/s against 5.7 at Ns=16); four pairs per trip, all eight loads issued before the first use.
    const int64_t per = (n2 + gridDim.x - 1) / gridDim.x, stride = blockDim.x;
    const int64_t n2b = (int64_t)(blockIdx.x + 1) * per < n2 ? (int64_t)(blockIdx.x + 1) * per : n2;
    int64_t i = (int64_t)blockIdx.x * per + threadIdx.x;
    for (; i + 3 * stride < n2b; i += 4 * stride) {
        dbl2v uu[4], ww[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { uu[q] = u2[i + q * stride]; ww[q] = w2[i + q * stride]; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const dbl2v vi = s_cur * uu[q];
            const dbl2v wi = ww[q] - alpha * vi;
            w2[i + q * stride] = wi; s += wi.x * wi.x + wi.y * wi.y;
            if (acc) a2[i + q * stride] += zk * vi;
        }
    }
    for (; i < n2b; i += stride) {
        const dbl2v vi = s_cur * u2[i];
        const dbl2v wi = w2[i] - alpha * vi;
        w2[i] = wi; s += wi.x * wi.x + wi.y * wi.y;
        if (acc) a2[i] += zk * vi;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const double vi = s_cur * u[n - 1], wi = w[n - 1] - alpha * vi;
        w[n - 1] = wi; s += wi * wi;
        if (acc) acc[n - 1] += zk * vi;
    }
    double r = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// Deferred-norm form (several ranks, one all-reduce per iteration): w holds H u_k of the UNNORMALISED u_k, the scales are known
// only now:  r_k = s_cur*w - alpha*s_cur*u - beta*s_prev*u_prev  (in place over w), partial |r_k|^2, optionally acc += zk*s_cur*u.
// the axpy/norm pass of the self-reducing iteration: alpha = sum of the product kernel's partial sums, taken by every workgroup;
// reads the in-flight scales (SC_N_*), commits them, writes alpha_k
__global__ void __launch_bounds__(256) k_axpy_norm_sr(double *__restrict__ w, const double *__restrict__ u, const double *__restrict__ scal, double *__restrict__ scal_w,
                                                      const double *__restrict__ part_a, int n_a, double *__restrict__ alphas, int64_t n, double *__restrict__ partial,
                                                      double *__restrict__ partial_host)      // pinned host copy of the beta^2 partial sums (null: none): edgpu_eigsh sums them itself
{
    __shared__ double red[17];
    const double s_cur = scal[SC_N_SCUR], sprev = scal[SC_N_SPREV], beta = scal[SC_N_BETA], K = scal[SC_N_K];
    const double alpha = sr_sum(part_a, n_a, red);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        alphas[(int)K] = alpha; scal_w[SC_ALPHA] = alpha;
        scal_w[SC_SCUR] = s_cur; scal_w[SC_SPREV] = sprev; scal_w[SC_BETA] = beta; scal_w[SC_K] = K;
    }
    double s = 0.0;
    const int64_t n2 = n >> 1;
    dbl2v *w2 = reinterpret_cast<dbl2v *>(w); const dbl2v *u2 = reinterpret_cast<const dbl2v *>(u);
    const int64_t per = (n2 + gridDim.x - 1) / gridDim.x, stride = blockDim.x;
    const int64_t n2b = (int64_t)(blockIdx.x + 1) * per < n2 ? (int64_t)(blockIdx.x + 1) * per : n2;
    for (int64_t i = (int64_t)blockIdx.x * per + threadIdx.x; i < n2b; i += stride) {
        const dbl2v vi = s_cur * u2[i];
        const dbl2v wi = w2[i] - alpha * vi;
        w2[i] = wi; s += wi.x * wi.x + wi.y * wi.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { const double vi = s_cur * u[n - 1], wi = w[n - 1] - alpha * vi; w[n - 1] = wi; s += wi * wi; }
    __syncthreads();
    const double r = block_sum(s, red);
    if (threadIdx.x == 0) { partial[blockIdx.x] = r; if (partial_host) partial_host[blockIdx.x] = r; }
}

__global__ void __launch_bounds__(256) k_axpy3_norm(double *__restrict__ w, const double *__restrict__ u, const double *__restrict__ up, double *__restrict__ acc,
                                                    const double *__restrict__ scal, int64_t n, double *__restrict__ partial)
{
    __shared__ double red[4];
    const double s_cur = scal[SC_SCUR], as = scal[SC_ALPHA] * scal[SC_SCUR], cp = scal[SC_BETA] * scal[SC_SPREV], zk = scal[SC_ZK];
    double s = 0.0;
    const int64_t n2 = n >> 1;
    dbl2v *w2 = reinterpret_cast<dbl2v *>(w); const dbl2v *u2 = reinterpret_cast<const dbl2v *>(u), *p2 = reinterpret_cast<const dbl2v *>(up); dbl2v *a2 = reinterpret_cast<dbl2v *>(acc);
    const int64_t per = (n2 + gridDim.x - 1) / gridDim.x, stride = blockDim.x;      // one contiguous piece per workgroup, as k_axpy_norm
    const int64_t n2b = (int64_t)(blockIdx.x + 1) * per < n2 ? (int64_t)(blockIdx.x + 1) * per : n2;
    int64_t i = (int64_t)blockIdx.x * per + threadIdx.x;
    for (; i + stride < n2b; i += 2 * stride) {
        dbl2v uu[2], ww[2], pp[2];
#pragma unroll
        for (int q = 0; q < 2; q++) { uu[q] = u2[i + q * stride]; ww[q] = w2[i + q * stride]; pp[q] = p2[i + q * stride]; }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const dbl2v wi = s_cur * ww[q] - as * uu[q] - cp * pp[q];
            w2[i + q * stride] = wi; s += wi.x * wi.x + wi.y * wi.y;
            if (acc) a2[i + q * stride] += (zk * s_cur) * uu[q];
        }
    }
    for (; i < n2b; i += stride) {
        const dbl2v uu = u2[i];
        const dbl2v wi = s_cur * w2[i] - as * uu - cp * p2[i];
        w2[i] = wi; s += wi.x * wi.x + wi.y * wi.y;
        if (acc) a2[i] += (zk * s_cur) * uu;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const double uu = u[n - 1], wi = s_cur * w[n - 1] - as * uu - cp * up[n - 1];
        w[n - 1] = wi; s += wi * wi;
        if (acc) acc[n - 1] += zk * s_cur * uu;
    }
    double r = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// scalar bookkeeping, one thread.  mode 0: alpha = TMP ; store alphas[k].
// mode 1: beta = sqrt(TMP) ; store betas[k+1] ; rotate scales (s_prev=s_cur, s_cur=1/beta).
// mode 2: start: s_cur = 1/sqrt(TMP), s_prev = 0, beta = 0.   mode 3: ZK = zvec[k].
__device__ __forceinline__ void scalar_step(double *scal, double *alphas, double *betas, const double *zvec, int k, int mode)
{
    if (k < 0) k = (int)scal[SC_K];      // device-side iteration counter
    if (mode == 0) { scal[SC_ALPHA] = scal[SC_TMP]; alphas[k] = scal[SC_TMP]; if (zvec) scal[SC_ZK] = zvec[k]; }
    else if (mode == 1) {
        double b = sqrt(scal[SC_TMP]);
        betas[k + 1] = b; scal[SC_BETA] = b; scal[SC_SPREV] = scal[SC_SCUR]; scal[SC_SCUR] = b > 0.0 ? 1.0 / b : 0.0; scal[SC_K] = (double)(k + 1);
    } else if (mode == 2) { scal[SC_SCUR] = 1.0 / sqrt(scal[SC_TMP]); scal[SC_SPREV] = 0.0; scal[SC_BETA] = 0.0; scal[SC_ALPHA] = 0.0; scal[SC_ZK] = 0.0; scal[SC_K] = 0.0; }
    else if (mode == 3) { scal[SC_ZK] = zvec ? zvec[k] : 0.0; }
    // fused iteration (k_uppass2_ax + k_dwpass2 with uprev2): the norm comes first, the alpha dot second
    else if (mode == 4) {      // after the local pass of iteration k: TMP = |u_k|^2
        const double b = sqrt(scal[SC_TMP]);
        if (k == 0) { scal[SC_SCUR] = b > 0.0 ? 1.0 / b : 0.0; scal[SC_SPREV] = 0.0; scal[SC_BETA] = 0.0; }
        else { betas[k] = b; scal[SC_BETA] = b; scal[SC_SPREV] = scal[SC_SCUR]; scal[SC_SCUR] = b > 0.0 ? 1.0 / b : 0.0; }
    } else if (mode == 5) { scal[SC_ALPHA] = scal[SC_TMP]; alphas[k] = scal[SC_TMP]; scal[SC_K] = (double)(k + 1); }   // after the dw pass: alpha_k
    // multi-eigenpair solver (eigsh): the newest Lanczos vector was re-orthogonalised in place, TMP = its new |.|^2:
    // beta of the iteration just completed and the scale of that vector follow it; nothing rotates, the counter stays
    else if (mode == 6) { const double b = sqrt(scal[SC_TMP]); const int kk = k > 0 ? k - 1 : 0; betas[kk + 1] = b; scal[SC_BETA] = b; scal[SC_SCUR] = b > 0.0 ? 1.0 / b : 0.0; }
    // thick restart: the current vector keeps its scale, the recurrence starts afresh (no previous vector), counter 0
    else if (mode == 7) { scal[SC_SPREV] = 0.0; scal[SC_BETA] = 0.0; scal[SC_ALPHA] = 0.0; scal[SC_K] = 0.0; }
    // fused iteration, local half REDONE on a vector that was modified after it was formed (re-orthogonalised in place):
    // TMP = its new |.|^2; beta and the scale follow, the previous scale stays (mode 4 would rotate it a second time)
    else if (mode == 8) { const double b = sqrt(scal[SC_TMP]); if (k > 0) betas[k] = b; scal[SC_BETA] = k > 0 ? b : 0.0; scal[SC_SCUR] = b > 0.0 ? 1.0 / b : 0.0; }
    else if (mode == 9) { scal[SC_ALPHA] = 0.0; }      // before such a redo: u = r - alpha*s*u_prev must leave r as it is
    // Several ranks, ONE all-reduce per iteration (lz_iter_deferred): the norm of the newest vector u_k = r_{k-1} stays a per-rank
    // partial sum (NPEND) and travels together with the next product's u_k.H u_k.
    // 10: before that product -- it runs on the unnormalised u_k with neutral scales (w = H u_k, TMP = u_k.H u_k)
    else if (mode == 10) { scal[SC_SV_SCUR] = scal[SC_SCUR]; scal[SC_SV_BETA] = scal[SC_BETA]; scal[SC_SCUR] = 1.0; scal[SC_BETA] = 0.0; }
    // 13: the two partial sums side by side for the 16-byte all-reduce
    else if (mode == 13) { scal[SC_PAIR] = scal[SC_TMP]; scal[SC_PAIR + 1] = scal[SC_NPEND]; }
    // 11: after it -- beta_k = |u_k|, the scales rotate, alpha_k = (u_k.H u_k) / |u_k|^2
    else if (mode == 11) {
        const double b = sqrt(scal[SC_PAIR + 1]), sn = b > 0.0 ? 1.0 / b : 0.0;
        betas[k] = b; scal[SC_SPREV] = scal[SC_SV_SCUR]; scal[SC_SCUR] = sn; scal[SC_BETA] = b;
        scal[SC_ALPHA] = scal[SC_PAIR] * sn * sn; alphas[k] = scal[SC_ALPHA];
        scal[SC_ZK] = zvec ? zvec[k] : 0.0;
    }
    // 12: flush -- NPEND (summed over ranks) is the norm of the vector iteration k left behind: what mode 1 does with TMP
    else if (mode == 12) { const double b = sqrt(scal[SC_NPEND]); betas[k + 1] = b; scal[SC_BETA] = b; scal[SC_SPREV] = scal[SC_SCUR]; scal[SC_SCUR] = b > 0.0 ? 1.0 / b : 0.0; scal[SC_K] = (double)(k + 1); }
}
__global__ void k_scalar_step(double *scal, double *alphas, double *betas, const double *zvec, int k, int mode)
{
    if (threadIdx.x | blockIdx.x) return;
    scalar_step(scal, alphas, betas, mode == 0 ? nullptr : zvec, k, mode);
}

// final reduction fused with the scalar step that consumes it (single rank: no all-reduce in between):
// scal[TMP] = sum(in[0..n)) in the same fixed order as k_reduce_final, then step `mode` with the device counter
__global__ void __launch_bounds__(1024) k_reduce_final_step(const double *__restrict__ in, int64_t n, double *scal, double *alphas, double *betas,
                                                            const double *zvec, int mode)
{
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    double r = block_sum(s, red);
    if (threadIdx.x == 0) { scal[SC_TMP] = r; scalar_step(scal, alphas, betas, zvec, -1, mode); }
}

// ---------------------------------------------------------------------------------------------
// Whole Lanczos loop of a SMALL sector in ONE workgroup (round 4).  A sector of N <= ~8 000 states (8 elements per lane) is launch bound: an
// iteration of the device loop above is four kernels of a few microseconds of work each, 24-30 us in all (profiles/r03/
// small_sectors.txt), and the reference's everyday sectors (Ns <= 12, O(Ns^2) of them per DMFT iteration, 200-step legs per
// state: ED_DIAG.f90:83-276, ED_GF_NORMAL.f90:224-239) are of that size.  Here 1024 threads keep v_k in LDS (the gathers of
// both hop families read it), their share of w and v_{k-1} in registers, and run `nsteps` iterations of SciFortran's
// lanczos_iteration (w = H v_k - beta_k v_{k-1}; alpha_k = v_k.w; w -= alpha_k v_k; beta_{k+1} = |w|) between two barriers each;
// alpha/beta go to the coefficient arrays, v_k optionally to a Krylov slab (the eigenvector is a weighted sum over it).
// Layout: the two-factor index i = iup + idw*DimUp is split into a LANE dimension A (the longer of the two, <= 1024) and a
// wave-uniform dimension B.  Wave w owns the 64-wide chunk c = w % nC of A and the B rows (w / nC) + (16/nC)*j: the hop entries
// of its A configuration stay in registers for the whole launch, those of B are wave-uniform scalar loads -- per element the
// loop touches LDS only (plus one L1-resident table read for the on-site coupling).
// ---------------------------------------------------------------------------------------------
struct LzWgArgs {
    int32_t DimA, DimB, sA, sB;            // flat index = a*sA + b*sB
    int32_t nC, E;                         // chunks of A (power of two, <= 16); B rows per wave
    const int32_t *cA; const double *vA; int32_t wA, ksA, asA;      // hop entry k of configuration a: cA[k*ksA + a*asA]
    const int32_t *cB; const double *vB; int32_t wB, ksB, bsB;
    const uint32_t *brec;                  // per row b of the uniform dimension ONE record (k_wg_pack_b): WB pre-scaled columns | WB amplitudes | E_b | impurity word * multB
    const double *EA, *EB; const int32_t *mapA, *mapB; const double *cross; int32_t multA, multB, impmask; double econst;
    double *v, *vprev;                     // v_k (any norm on entry when k0 == 0), v_{k-1}; both updated for a following launch
    double *alphas, *betas; int32_t k0, nsteps; double threshold;
    double *krylov; int64_t kstride;       // optional: v_k stored at krylov + k*kstride
    int32_t *done;                         // out: iterations completed (k0 + nsteps, or fewer when beta fell below the threshold)
};
// record of row b: dwords [0, WB): column * sB (slots beyond the row's width: 0 with amplitude 0.0) | [WB, 3*WB): amplitudes | E_b | imp * multB | pad
template <int WB> struct WgRec { static constexpr int kDwords = (3 * WB + 3 + 3) & ~3; };
template <int WB>
__global__ void __launch_bounds__(64) k_wg_pack_b(LzWgArgs A, uint32_t *rec)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= A.DimB) return;
    uint32_t *r = rec + (int64_t)b * WgRec<WB>::kDwords;
    for (int q = 0; q < WB; q++) {
        const bool h = q < A.wB;
        const int32_t c = h ? A.cB[(int64_t)q * A.ksB + (int64_t)b * A.bsB] * A.sB : 0;
        const double v = h ? A.vB[(int64_t)q * A.ksB + (int64_t)b * A.bsB] : 0.0;
        r[q] = (uint32_t)c; r[WB + 2 * q] = (uint32_t)__double2loint(v); r[WB + 2 * q + 1] = (uint32_t)__double2hiint(v);
    }
    const double e = A.EB[b];
    r[3 * WB] = (uint32_t)__double2loint(e); r[3 * WB + 1] = (uint32_t)__double2hiint(e);
    r[3 * WB + 2] = (uint32_t)((A.mapB[b] & A.impmask) * A.multB);
}
// VPREG: v_{k-1} stays in registers (EMAX <= 8); otherwise it lives in A.vprev (L2-resident, 8 bytes read and written per element
// and iteration) -- with 16-20 elements per lane two register arrays do not fit 128 registers
template <int EMAX, int WA, int WB, bool VPREG>
__global__ void __launch_bounds__(1024) k_lanczos_wg(LzWgArgs A)
{
    extern __shared__ double lds[];
    __shared__ double red[16]; __shared__ double bc[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = A.DimA * A.DimB;
    const int c = wv % A.nC, b0 = wv / A.nC, TB = 16 / A.nC;
    const int a = 64 * c + lane; const bool aok = a < A.DimA;
    const int ac = aok ? a : 0;
    // register-resident tables of this lane's A configuration
    int32_t colA[WA]; double valA[WA];
#pragma unroll
    for (int k = 0; k < WA; k++) { const bool h = k < A.wA; colA[k] = h ? A.cA[(int64_t)k * A.ksA + (int64_t)ac * A.asA] * A.sA : 0; valA[k] = (h && aok) ? A.vA[(int64_t)k * A.ksA + (int64_t)ac * A.asA] : 0.0; }
    const double eA = A.EA[ac] + A.econst; const int impA = (A.mapA[ac] & A.impmask) * A.multA;
    auto bsum = [&](double x) -> double {      // sum over the workgroup, same value on every thread
        x = wave_sum_dpp(x);
        __syncthreads();
        if (lane == 0) red[wv] = x;
        __syncthreads();
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) r += red[i];
        return r;
    };
    double w[EMAX], vp[VPREG ? EMAX : 1];
    // start: v_k into LDS; k0 == 0: normalise the input and clear v_{-1}
    for (int i = tid; i < N; i += 1024) lds[i] = A.v[i];
#pragma unroll
    for (int j = 0; j < EMAX; j++) {
        const int b = b0 + TB * j; const bool in = j < A.E && aok && b < A.DimB;
        if (VPREG) vp[j] = (A.k0 > 0 && in) ? A.vprev[a * A.sA + b * A.sB] : 0.0;
        else if (A.k0 == 0 && in) A.vprev[a * A.sA + b * A.sB] = 0.0;
    }
    __syncthreads();
    double beta = A.k0 > 0 ? A.betas[A.k0] : 0.0;
    if (A.k0 == 0) {
        double n2 = 0.0;
#pragma unroll
        for (int j = 0; j < EMAX; j++) { const int b = b0 + TB * j; if (j < A.E && aok && b < A.DimB) { const double x = lds[a * A.sA + b * A.sB]; n2 += x * x; } }
        n2 = bsum(n2);
        const double sc = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EMAX; j++) { const int b = b0 + TB * j; if (j < A.E && aok && b < A.DimB) lds[a * A.sA + b * A.sB] *= sc; }
        __syncthreads();
    }
    int k = A.k0; const int kend = A.k0 + A.nsteps;
    for (; k < kend; k++) {
        // everything derived from the row numbers is loop invariant; hoisted out of the k loop it would need ~6 registers per
        // element (addresses, row energies, coupling entries): opaque per iteration, so that it is recomputed next to its use
        int b0v = b0; asm volatile("" : "+s"(b0v));
        if (A.krylov) { double *kv = A.krylov + (int64_t)k * A.kstride; for (int i = tid; i < N; i += 1024) kv[i] = lds[i]; }
        double asum = 0.0;
#pragma unroll
        for (int j = 0; j < EMAX; j++) {
            const int b = __builtin_amdgcn_readfirstlane(b0v + TB * j);
            if (j < A.E && b < A.DimB) {
                const int ib = b * A.sB, ia = a * A.sA;
                // the hop list of row b is wave-uniform: ONE record per row (k_wg_pack_b), read with a few wide scalar loads through
                // the constant address space (the tables are not written while the kernel runs; only a load the compiler knows to be
                // invariant becomes a scalar load).  Entry by entry out of the ELL images it was 2*WB loads with 64-bit index
                // arithmetic and a wait each: 178 instructions and ~14 serial round trips per element.
                typedef const __attribute__((address_space(4))) uint32_t *cu32;
                const cu32 rb = (cu32)(uintptr_t)A.brec + (int64_t)b * WgRec<WB>::kDwords;
                int cb[WB]; double vb[WB];
#pragma unroll
                for (int q = 0; q < WB; q++) { cb[q] = (int)rb[q]; vb[q] = __hiloint2double((int)rb[WB + 2 * q + 1], (int)rb[WB + 2 * q]); }
                const double eB = __hiloint2double((int)rb[3 * WB + 1], (int)rb[3 * WB]);
                const int impB = (int)rb[3 * WB + 2];
                double acc = 0.0;
                if (aok) {
                    const double x = lds[ia + ib];
                    acc = (eA + eB + A.cross[impA + impB]) * x;
#pragma unroll
                    for (int q = 0; q < WA; q++) acc += valA[q] * lds[colA[q] + ib];
#pragma unroll
                    for (int q = 0; q < WB; q++) acc += vb[q] * lds[ia + cb[q]];
                    if (VPREG) { acc -= beta * vp[j]; asum += x * acc; }
                }
                w[j] = acc;
            }
            __builtin_amdgcn_sched_barrier(0);      // one element's scalar loads at a time: hoisted across elements they spill the scalar registers into lanes
        }
        if (!VPREG) {      // v_{k-1} from L2: every element's value requested before the first is used (one exposed round trip per iteration, not E)
            double t[EMAX];
#pragma unroll
            for (int j = 0; j < EMAX; j++) { const int b = b0v + TB * j; t[j] = (j < A.E && aok && b < A.DimB) ? A.vprev[a * A.sA + b * A.sB] : 0.0; }
#pragma unroll
            for (int j = 0; j < EMAX; j++) { const int b = b0v + TB * j; if (j < A.E && aok && b < A.DimB) { w[j] -= beta * t[j]; asum += lds[a * A.sA + b * A.sB] * w[j]; } }
        }
        const double alpha = bsum(asum);
        double nsum = 0.0;
#pragma unroll
        for (int j = 0; j < EMAX; j++) {
            const int b = b0v + TB * j;
            if (j < A.E && aok && b < A.DimB) { const double x = lds[a * A.sA + b * A.sB]; w[j] -= alpha * x; if (VPREG) vp[j] = x; else A.vprev[a * A.sA + b * A.sB] = x; nsum += w[j] * w[j]; }
        }
        const double bnew = sqrt(bsum(nsum));            // the second barrier inside bsum also ends every read of v_k in LDS
        const double sc = bnew > 0.0 ? 1.0 / bnew : 0.0;
#pragma unroll
        for (int j = 0; j < EMAX; j++) { const int b = b0v + TB * j; if (j < A.E && aok && b < A.DimB) lds[a * A.sA + b * A.sB] = w[j] * sc; }
        if (tid == 0) { A.alphas[k] = alpha; A.betas[k + 1] = bnew; }
        beta = bnew;
        __syncthreads();
        if (bnew < A.threshold) { k++; break; }          // invariant subspace: same decision on every thread
    }
    // state for a following launch: v_k, v_{k-1}
    for (int i = tid; i < N; i += 1024) A.v[i] = lds[i];
#pragma unroll
    for (int j = 0; j < EMAX; j++) { const int b = b0 + TB * j; if (VPREG && j < A.E && aok && b < A.DimB) A.vprev[a * A.sA + b * A.sB] = vp[j]; }
    if (tid == 0) *A.done = k;
}

// x[i] *= scal[idx]
// zero the padding columns [DimUp, ld) of every row of a pitched vector (elementwise passes then run over the padded length)
__global__ void __launch_bounds__(256) k_zero_pads(double *__restrict__ x, int64_t nrows, int32_t DimUp, int32_t ld)
{
    const int np = ld - DimUp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrows * np; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / np; x[r * ld + DimUp + (i - r * np)] = 0.0;
    }
}
__global__ void __launch_bounds__(256) k_scale(double *__restrict__ x, const double *__restrict__ scal, int idx, int64_t n)
{
    const double s = scal[idx];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= s;
}

// ---------------------------------------------------------------------------------------------
// Diagonal impurity operator applied to a state: out_i = (sum_a cu[a] n_up,a(i) + cd[a] n_dw,a(i)) * in_i.
// Covers the start vectors of the susceptibility builders: Sz_a (cu=+1/2, cd=-1/2), n_a (1,1), their
// two-orbital and total combinations (ED_GF_CHISPIN.f90:169-171, 282-284, 393-394; ED_GF_CHIDENS.f90:166-168,
// 266-278, 389-390).
// ---------------------------------------------------------------------------------------------
struct DiagOpArgs {
    int32_t nfac, Norb, total_ud; int32_t dims[2 * EDGPU_MAX_ORB]; const int32_t *maps[2 * EDGPU_MAX_ORB];
    double cu[EDGPU_MAX_ORB], cd[EDGPU_MAX_ORB]; int64_t Nel, N, row0; const double *vin; double *vout;
};
__global__ void __launch_bounds__(256) k_apply_diagop(DiagOpArgs A)
{
    const int nud = A.nfac / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.N; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = i % A.Nel + A.row0;
        uint32_t mu = 0, md = 0;
        for (int k = 0; k < A.nfac; k++) {
            const int dg = (int)(c % A.dims[k]); c /= A.dims[k];
            const uint32_t m = (uint32_t)A.maps[k][dg];
            if (A.total_ud) { if (k == 0) mu = m; else md = m; }
            else { if (k < nud) mu |= (m & 1u) << k; else md |= (m & 1u) << (k - nud); }
        }
        double f = 0.0;
        for (int a = 0; a < A.Norb; a++) f += A.cu[a] * (double)((mu >> a) & 1u) + A.cd[a] * (double)((md >> a) & 1u);
        A.vout[i] = f * A.vin[i];
    }
}

// State of the twin sector B(ndw,nup) from the state of A(nup,ndw): vector_B(i) = cvec_A(Order(i)) with
// Order = twin_sector_order (ED_SETUP.f90:852-880, ED_EIGENSPACE.f90:486-497).  Sorting the flipped global
// state numbers is the index permutation that swaps the up and dw halves of the digit tuple, i.e. a
// generalised transpose per phonon block -- computed directly, no sort.
struct TwinArgs { int32_t nud, DimPh; int32_t dimsA[2 * EDGPU_MAX_ORB]; int64_t nel; int64_t nel_out, row0_out; const double *vin; double *vout; };   // nel: global electronic dim (vin is global); output window [row0_out, row0_out+nel_out) per phonon block
__global__ void __launch_bounds__(256) k_twin_reorder(TwinArgs A)
{
    const int64_t n = A.nel_out * A.DimPh;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t iph = j / A.nel_out; int64_t c = j - iph * A.nel_out + A.row0_out;
        // digits of the B index: [ups_B (= dws_A) ..., dws_B (= ups_A) ...]
        int32_t dg[2 * EDGPU_MAX_ORB];
        for (int k = 0; k < A.nud; k++) { const int d = A.dimsA[k + A.nud]; dg[k + A.nud] = (int32_t)(c % d); c /= d; }   // dws_A
        for (int k = 0; k < A.nud; k++) { const int d = A.dimsA[k]; dg[k] = (int32_t)(c % d); c /= d; }                  // ups_A
        int64_t i = 0, stride = 1;
        for (int k = 0; k < 2 * A.nud; k++) { i += dg[k] * stride; stride *= A.dimsA[k]; }
        A.vout[j] = A.vin[i + iph * A.nel];
    }
}

// Off-diagonal impurity density matrix <c^+_a c_b> per spin (ED_OBSERVABLES.f90:283-313, ed_total_ud=T):
// dm[s][a][b] = sum_i sgn1*sgn2 * v_i * v_j, |j> = c^+_a c_b |i> acting on the spin-s word (a != b).
// One workgroup per (idw,iph) row; out[row*50 + s*25 + a*5 + b] partial sums in a fixed order.
struct DmArgs { int32_t DimUp, DimDw, Norb; int64_t Qdw, dw0, nrows; const int32_t *mapup, *mapdw, *rankup, *rankdw; const double *v; const double *vg; double *out; };   // vg: gathered vector (global layout) with more than one rank, else null
__global__ void __launch_bounds__(256) k_imp_dm_rows(DmArgs A)
{
    __shared__ double red[4];
    const int64_t row = blockIdx.x;
    const int64_t iph = row / A.Qdw, idwl = row - iph * A.Qdw, idw = A.dw0 + idwl;
    const uint32_t mdw = (uint32_t)A.mapdw[idw];
    const double *vrow = A.v + row * A.DimUp;
    for (int s = 0; s < 2; s++)
        for (int a = 0; a < A.Norb; a++)
            for (int b = 0; b < A.Norb; b++) {
                if (a == b) continue;
                double acc = 0.0;
                const uint32_t ba = 1u << a, bb = 1u << b;
                if (s == 0) {          // up word changes: partner in the same row
                    for (int i = threadIdx.x; i < A.DimUp; i += blockDim.x) {
                        const uint32_t m = (uint32_t)A.mapup[i];
                        if ((m & bb) && !(m & ba)) {
                            const uint32_t r = m & ~bb, k = r | ba;
                            const int sg = __popc(m & (bb - 1u)) + __popc(r & (ba - 1u));
                            acc += ((sg & 1) ? -1.0 : 1.0) * vrow[i] * vrow[A.rankup[k]];
                        }
                    }
                } else if ((mdw & bb) && !(mdw & ba)) {   // dw word changes: partner row (of another rank: read from the gathered vector)
                    const uint32_t r = mdw & ~bb, k = r | ba;
                    const int sg = __popc(mdw & (bb - 1u)) + __popc(r & (ba - 1u));
                    const double sgn = (sg & 1) ? -1.0 : 1.0;
                    const double *prow = A.vg ? A.vg + (iph * (int64_t)A.DimDw + A.rankdw[k]) * A.DimUp : A.v + (iph * A.Qdw + (A.rankdw[k] - A.dw0)) * A.DimUp;
                    for (int i = threadIdx.x; i < A.DimUp; i += blockDim.x) acc += sgn * vrow[i] * prow[i];
                }
                double t = block_sum(acc, red);
                if (threadIdx.x == 0) A.out[row * 50 + s * 25 + a * 5 + b] = t;
            }
}

// (b + b^+)|in> on the phonon index (ED_GF_PHONON.f90:76-90 scatters; gather form here):
// out(e,p) = sqrt(p+1) in(e,p+1) + sqrt(p) in(e,p-1)
__global__ void __launch_bounds__(256) k_apply_phonon_x(const double *__restrict__ vin, double *__restrict__ vout, int64_t nel, int dimph)
{
    const int64_t n = nel * dimph;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / nel);
        double x = 0.0;
        if (p + 1 < dimph) x += sqrt((double)(p + 1)) * vin[i + nel];
        if (p > 0) x += sqrt((double)p) * vin[i - nel];
        vout[i] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// Impurity-configuration weights of a state: hist[(mu_imp*nI + md_imp) + nI*nI*iph] = sum |v_i|^2.
// Every local observable of lanc_observables (ED_OBSERVABLES.f90:150-215: dens, dens_up/dw, docc,
// magz, sz2, n2, s2tot, Prob, prob_ph, dens_ph) is a function of this (2^Norb)^2 x DimPh table, so
// the N-vector never leaves HBM.  Two-factor layout: one workgroup per (idw,iph) row produces the
// nI partial sums over iup classes in a fixed order (deterministic); out[row*nI + mu].
// ---------------------------------------------------------------------------------------------
struct HistArgs { int32_t DimUp, nI, impmask; int64_t nrows; const int32_t *mapup; const double *v; double *out; };
__global__ void __launch_bounds__(256) k_imp_hist_rows(HistArgs A)
{
    __shared__ double red[4];
    const int64_t row = blockIdx.x;
    double acc[32];
#pragma unroll
    for (int c = 0; c < 32; c++) acc[c] = 0.0;
    for (int i = threadIdx.x; i < A.DimUp; i += blockDim.x) {
        const double x = A.v[row * A.DimUp + i];
        const int c = A.mapup[i] & A.impmask;
#pragma unroll
        for (int q = 0; q < 32; q++) if (q == c) acc[q] += x * x;
    }
#pragma unroll
    for (int c = 0; c < 32; c++) if (c < A.nI) { double r = block_sum(acc[c], red); if (threadIdx.x == 0) A.out[row * A.nI + c] = r; }   // uniform guard, registers only
}
// generic digit layout (ed_total_ud=F): LDS histogram with fp64 atomics (summation order not fixed)
struct HistGenArgs { int32_t nfac, nI, DimPh; int32_t dims[2 * EDGPU_MAX_ORB]; const int32_t *maps[2 * EDGPU_MAX_ORB]; int64_t Nel, row0; const double *v; double *out; };
__global__ void __launch_bounds__(256) k_imp_hist_generic(HistGenArgs A)
{
    extern __shared__ double hs[];   // nI*nI*DimPh
    const int nb = A.nI * A.nI * A.DimPh;
    for (int t = threadIdx.x; t < nb; t += blockDim.x) hs[t] = 0.0;
    __syncthreads();
    const int64_t N = A.Nel * A.DimPh; const int nud = A.nfac / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t iph = i / A.Nel; int64_t c = i - iph * A.Nel + A.row0;
        uint32_t mu = 0, md = 0;
        for (int k = 0; k < A.nfac; k++) {
            const int dg = (int)(c % A.dims[k]); c /= A.dims[k];
            const uint32_t bit = (uint32_t)A.maps[k][dg] & 1u;
            if (k < nud) mu |= bit << k; else md |= bit << (k - nud);
        }
        const double x = A.v[i];
        atomicAdd(&hs[(mu * A.nI + md) + (int64_t)A.nI * A.nI * iph], x * x);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nb; t += blockDim.x) A.out[(int64_t)blockIdx.x * nb + t] = hs[t];
}

}  // namespace edgpu
