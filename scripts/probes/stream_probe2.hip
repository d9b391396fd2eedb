// Development probe 2: the local-block stream (values+columns+row pointers+u read, y write) under the launch shapes of the
// product kernels: one-wave blocks vs persistent 512-thread workgroups (one per CU, 160 KB of LDS), with and without the
// per-dw-row window load + barriers, with the entry range taken from a row-pointer load (dependent) or computed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Args { const dbl2 *a, *b; const int64_t *rp; const double *u; double *y; const double *x; int64_t nchunk; int cpr; int64_t nrow; int dep, win, d2; };
__device__ __forceinline__ double chunk(const Args &A, int64_t c, int lane, int64_t p0)
{
    dbl2 r[7];
    const dbl2 *pv = A.a + (p0 >> 1), *pc = A.b + (p0 >> 2);
#pragma unroll
    for (int j = 0; j < 5; j++) { int g = lane + 64 * j; g = g < 287 ? g : 287; r[j] = __builtin_nontemporal_load(pv + g); }
#pragma unroll
    for (int j = 0; j < 2; j++) { int g = lane + 64 * j; g = g < 143 ? g : 143; r[5 + j] = __builtin_nontemporal_load(pc + g); }
    double acc = (double)A.rp[c * 64 + lane] + A.u[c * 64 + lane];
#pragma unroll
    for (int j = 0; j < 7; j++) acc += r[j].x + r[j].y;
    return acc;
}
template <int NT> __global__ void __launch_bounds__(NT) k(Args A)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = NT / 64;
    double acc = 0.0;
    if (NT == 64) {
        for (int64_t c = blockIdx.x; c < A.nchunk; c += gridDim.x) {
            const int64_t p0 = A.dep ? A.rp[c * 64] : c * 576;
            const double s = chunk(A, c, lane, p0);
            A.y[c * 64 + lane] = s; acc += s;
        }
    } else {
        // persistent: workgroup b owns dw rows b, b+grid, ...; wave w the chunks w, w+nw, ... of the row (cpr chunks per row)
        for (int64_t row = blockIdx.x; row < A.nrow; row += gridDim.x) {
            if (A.win) {
                __syncthreads();
                for (int i = threadIdx.x; i < 6435; i += NT) reinterpret_cast<dbl2 *>(lds)[i] = reinterpret_cast<const dbl2 *>(A.x + row * 12870)[i];
                __syncthreads();
            }
            for (int q = wv; q < A.cpr; q += nw) {
                const int64_t c = row * A.cpr + q;
                if (c < A.nchunk) {
                    const int64_t p0 = A.dep ? A.rp[c * 64] : c * 576;
                    const double s = chunk(A, c, lane, p0);
                    A.y[c * 64 + lane] = s; acc += s + (A.win ? lds[(lane * 97 + q) % 12870] : 0.0);
                }
            }
        }
    }
    if (acc == 1.2345) A.y[0] = acc;
}
int main()
{
    const int64_t nrow = 12870; const int cpr = 201; const int64_t nchunk = nrow * cpr;
    Args A{}; dbl2 *a, *b; int64_t *rp; double *u, *y, *x;
    CK(hipMalloc(&a, (size_t)nchunk * 4608 + 4096)); CK(hipMalloc(&b, (size_t)nchunk * 2304 + 4096));
    CK(hipMalloc(&rp, (size_t)nchunk * 512 + 64)); CK(hipMalloc(&u, (size_t)nchunk * 512)); CK(hipMalloc(&y, (size_t)nchunk * 512)); CK(hipMalloc(&x, (size_t)nrow * 12870 * 8));
    CK(hipMemset(a, 0, (size_t)nchunk * 4608)); CK(hipMemset(b, 0, (size_t)nchunk * 2304)); CK(hipMemset(u, 0, (size_t)nchunk * 512)); CK(hipMemset(x, 0, (size_t)nrow * 12870 * 8));
    { int64_t *h = (int64_t *)malloc((size_t)nchunk * 512 + 64); for (int64_t i = 0; i <= nchunk * 64; i++) h[i] = i * 9; CK(hipMemcpy(rp, h, (size_t)nchunk * 512 + 8, hipMemcpyHostToDevice)); free(h); }
    A.a = a; A.b = b; A.rp = rp; A.u = u; A.y = y; A.x = x; A.nchunk = nchunk; A.cpr = cpr; A.nrow = nrow;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const double bytes = (double)nchunk * (6912 + 512 + 512 + 512);
    struct Cfg { int nt, grid, lds, dep, win; } cfgs[] = {
        {64, 5120, 0, 0, 0}, {64, 5120, 0, 1, 0}, {64, 2048, 0, 1, 0},
        {512, 256, 0, 0, 0}, {512, 256, 159 * 1024, 0, 0}, {512, 256, 159 * 1024, 1, 0}, {512, 256, 159 * 1024, 0, 1}, {512, 256, 159 * 1024, 1, 1},
        {1024, 256, 159 * 1024, 0, 0}, {1024, 256, 159 * 1024, 1, 1}, {512, 512, 70 * 1024, 0, 0}};
    for (auto &c : cfgs) {
        A.dep = c.dep; A.win = c.win;
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            if (c.nt == 64) hipLaunchKernelGGL(k<64>, dim3(c.grid), dim3(64), c.lds, 0, A);
            else if (c.nt == 512) hipLaunchKernelGGL(k<512>, dim3(c.grid), dim3(512), c.lds, 0, A);
            else hipLaunchKernelGGL(k<1024>, dim3(c.grid), dim3(1024), c.lds, 0, A);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("nt %4d grid %5d lds %6d dep %d win %d: %.3f ms  %.0f GB/s\n", c.nt, c.grid, c.lds, c.dep, c.win, best, bytes / best / 1e6);
    }
    return 0;
}
