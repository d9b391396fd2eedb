// Development probe 5: the result write inside a CSR-shaped read stream -- does its cost depend on how the writes are
// batched?  A wave handles G consecutive 64-row chunks per step (G x 6912 B read) and then writes the G x 512 B of results
// in one go (16-B stores when G >= 2); variants: plain / non-temporal stores; 256-thread workgroups (4 waves, results
// exchanged through LDS and written as one 2 KB x G block after a barrier).
// hipcc --offload-arch=gfx950 -O3 -o stream_probe5 stream_probe5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int G, bool NT, bool WR> __global__ void __launch_bounds__(64) k(const dbl2 *__restrict__ a, double *__restrict__ y, int64_t ngroup)
{
    const int lane = threadIdx.x;
    double acc = 0.0;
    for (int64_t g = blockIdx.x; g < ngroup; g += gridDim.x) {
        double s[G];
#pragma unroll
        for (int q = 0; q < G; q++) {
            dbl2 r[7];
            const dbl2 *p = a + (g * G + q) * 432;
#pragma unroll
            for (int j = 0; j < 7; j++) { int i = lane + 64 * j; i = i < 431 ? i : 431; r[j] = __builtin_nontemporal_load(p + i); }
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 7; j++) t += r[j].x + r[j].y;
            s[q] = t; acc += t;
        }
        if (WR) {
            double *yo = y + g * G * 64;
            if (G == 1) { if (NT) __builtin_nontemporal_store(s[0], yo + lane); else yo[lane] = s[0]; }
            else {
#pragma unroll
                for (int q = 0; q < G; q += 2) {
                    dbl2 v; v.x = s[q]; v.y = s[q + 1];
                    dbl2 *d = reinterpret_cast<dbl2 *>(yo + q * 64) + lane;
                    if (NT) __builtin_nontemporal_store(v, d); else *d = v;
                }
            }
        }
    }
    if (acc == 1.2345) y[0] = acc;
}
// 256-thread workgroup: 4 waves x G chunks, results staged in LDS, written after a barrier as one contiguous block
template <int G, bool NT> __global__ void __launch_bounds__(256) kw(const dbl2 *__restrict__ a, double *__restrict__ y, int64_t ngroup)
{
    __shared__ double st[4 * G * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc = 0.0;
    for (int64_t g = blockIdx.x; g < ngroup; g += gridDim.x) {
#pragma unroll
        for (int q = 0; q < G; q++) {
            dbl2 r[7];
            const dbl2 *p = a + ((g * 4 + wv) * G + q) * 432;
#pragma unroll
            for (int j = 0; j < 7; j++) { int i = lane + 64 * j; i = i < 431 ? i : 431; r[j] = __builtin_nontemporal_load(p + i); }
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 7; j++) t += r[j].x + r[j].y;
            st[(wv * G + q) * 64 + lane] = t; acc += t;
        }
        __syncthreads();
        dbl2 *d = reinterpret_cast<dbl2 *>(y + g * 4 * G * 64);
        for (int i = threadIdx.x; i < 4 * G * 32; i += 256) { dbl2 v = reinterpret_cast<dbl2 *>(st)[i]; if (NT) __builtin_nontemporal_store(v, d + i); else d[i] = v; }
        __syncthreads();
    }
    if (acc == 1.2345) y[0] = acc;
}
int main()
{
    const int64_t nchunk = (165636900LL / 64) / 32 * 32;
    dbl2 *a; double *y;
    CK(hipMalloc(&a, (size_t)nchunk * 6912)); CK(hipMalloc(&y, (size_t)nchunk * 512));
    CK(hipMemset(a, 0, (size_t)nchunk * 6912)); CK(hipMemset(y, 0, (size_t)nchunk * 512));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch, bool wr) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double bytes = (double)nchunk * (6912 + (wr ? 512 : 0));
        printf("%-52s %.3f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
    };
#define R1(G, NT, WR, GRID, NAME) run(NAME, [&] { hipLaunchKernelGGL((k<G, NT, WR>), dim3(GRID), dim3(64), 0, 0, a, y, nchunk / G); }, WR)
#define RW(G, NT, GRID, NAME) run(NAME, [&] { hipLaunchKernelGGL((kw<G, NT>), dim3(GRID), dim3(256), 0, 0, a, y, nchunk / (4 * G)); }, true)
    R1(1, false, false, 4096, "wave, G=1, no write");
    R1(1, false, true, 4096, "wave, G=1, plain 8-B stores");
    R1(2, false, false, 4096, "wave, G=2, no write");
    R1(2, false, true, 4096, "wave, G=2, plain 16-B stores");
    R1(2, true, true, 4096, "wave, G=2, nt 16-B stores");
    R1(4, false, true, 4096, "wave, G=4, plain 16-B stores");
    R1(4, true, true, 4096, "wave, G=4, nt 16-B stores");
    R1(8, false, false, 2048, "wave, G=8, no write");
    R1(8, false, true, 2048, "wave, G=8, plain 16-B stores");
    R1(8, true, true, 2048, "wave, G=8, nt 16-B stores");
    RW(1, false, 2048, "256-thread wg, 4 chunks, 2 KB block write, plain");
    RW(1, true, 2048, "256-thread wg, 4 chunks, 2 KB block write, nt");
    RW(4, false, 1024, "256-thread wg, 16 chunks, 8 KB block write, plain");
    RW(4, true, 1024, "256-thread wg, 16 chunks, 8 KB block write, nt");
    RW(8, false, 1024, "256-thread wg, 32 chunks, 16 KB block write, plain");
    return 0;
}
