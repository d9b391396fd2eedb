// Development probe: how much HBM read bandwidth does a CSR-like access pattern get from ONE interleaved stream versus
// separate value / column / row-pointer / vector streams?  hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// one wave per block; chunk = 64 rows x 9 entries: 4608 B of values + 2304 B of columns (+512 B row pointers, +512 B vector read, +512 B write)
// mode 0: one stream (6912 B contiguous per chunk)   1: values and columns from two arrays   2: + row pointers   3: + vector read + write
template <int MODE> __global__ void __launch_bounds__(64) k(const dbl2 *__restrict__ a, const dbl2 *__restrict__ b, const double *__restrict__ rp,
                                                          const double *__restrict__ u, double *__restrict__ y, int64_t nchunk)
{
    const int lane = threadIdx.x;
    double acc = 0.0;
    for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
        dbl2 r[7];
        if (MODE == 0) {
            const dbl2 *p = a + c * 432;              // 6912 B = 432 x 16 B
#pragma unroll
            for (int j = 0; j < 7; j++) { int g = lane + 64 * j; g = g < 431 ? g : 431; r[j] = __builtin_nontemporal_load(p + g); }
        } else {
            const dbl2 *pv = a + c * 288, *pc = b + c * 144;   // 4608 B + 2304 B
#pragma unroll
            for (int j = 0; j < 5; j++) { int g = lane + 64 * j; g = g < 287 ? g : 287; r[j] = __builtin_nontemporal_load(pv + g); }
#pragma unroll
            for (int j = 0; j < 2; j++) { int g = lane + 64 * j; g = g < 143 ? g : 143; r[5 + j] = __builtin_nontemporal_load(pc + g); }
        }
#pragma unroll
        for (int j = 0; j < 7; j++) acc += r[j].x + r[j].y;
        if (MODE >= 2) acc += rp[c * 64 + lane];
        if (MODE >= 3) { acc += u[c * 64 + lane]; y[c * 64 + lane] = acc; }
    }
    if (acc == 1.2345) y[0] = acc;
}
int main(int argc, char **argv)
{
    const int64_t nchunk = 165636900LL / 64;           // local block of c3: 2.59 M chunks, 17.9 GB
    dbl2 *a, *b; double *rp, *u, *y;
    CK(hipMalloc(&a, (size_t)nchunk * 6912)); CK(hipMalloc(&b, (size_t)nchunk * 2304));
    CK(hipMalloc(&rp, (size_t)nchunk * 512)); CK(hipMalloc(&u, (size_t)nchunk * 512)); CK(hipMalloc(&y, (size_t)nchunk * 512));
    CK(hipMemset(a, 0, (size_t)nchunk * 6912)); CK(hipMemset(b, 0, (size_t)nchunk * 2304)); CK(hipMemset(rp, 0, (size_t)nchunk * 512)); CK(hipMemset(u, 0, (size_t)nchunk * 512));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {256 * 8, 256 * 16, 256 * 20, 256 * 32}) {
        for (int mode = 0; mode < 4; mode++) {
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, a, b, rp, u, y, nchunk);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, a, b, rp, u, y, nchunk);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, a, b, rp, u, y, nchunk);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, a, b, rp, u, y, nchunk);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double bytes = (double)nchunk * (6912 + (mode >= 2 ? 512 : 0) + (mode >= 3 ? 1024 : 0));
            printf("grid %5d mode %d: %.3f ms  %.0f GB/s\n", grid, mode, best, bytes / best / 1e6);
        }
    }
    return 0;
}
