// Development probe 4: what does the small result write (512 B per 64-row chunk) cost inside a CSR-shaped read stream?
// One-wave blocks, persistent grid; per chunk 6912 B of entries are read; variants add the operand read and/or the result
// write in several forms.  hipcc --offload-arch=gfx950 -O3 -o stream_probe4 stream_probe4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// MODE 0: entries only   1: + u read   2: + y write (plain)   3: + y write (non-temporal)   4: + y read-modify-write (plain)
//      5: + y read-modify-write (nt store)   6: plain write, each wave owns a CONTIGUOUS run of chunks   7: as 6, nt store
//      8: u read + y write nt (the local block's shape)   9: plain y write issued BEFORE the chunk's loads are waited for (value of previous chunk)
template <int MODE> __global__ void __launch_bounds__(64) k(const dbl2 *__restrict__ a, const double *__restrict__ u, double *__restrict__ y, int64_t nchunk)
{
    const int lane = threadIdx.x;
    double acc = 0.0, prev = 0.0;
    const bool contig = MODE == 6 || MODE == 7;
    const int64_t per = (nchunk + gridDim.x - 1) / gridDim.x;
    const int64_t c0 = contig ? blockIdx.x * per : blockIdx.x, c1 = contig ? (c0 + per < nchunk ? c0 + per : nchunk) : nchunk, cs = contig ? 1 : gridDim.x;
    for (int64_t c = c0; c < c1; c += cs) {
        dbl2 r[7];
        const dbl2 *p = a + c * 432;
#pragma unroll
        for (int j = 0; j < 7; j++) { int g = lane + 64 * j; g = g < 431 ? g : 431; r[j] = (MODE >= 13) ? p[g] : __builtin_nontemporal_load(p + g); }
        double s = 0.0;
        if (MODE == 1 || MODE == 8) s = u[c * 64 + lane];
        if (MODE == 4 || MODE == 5) s = y[c * 64 + lane];
        if (MODE == 9) { y[c * 64 + lane] = prev; }
#pragma unroll
        for (int j = 0; j < 7; j++) s += r[j].x + r[j].y;
        if (MODE == 2 || MODE == 4 || MODE == 6 || MODE == 14) y[c * 64 + lane] = s;
        if (MODE == 3 || MODE == 5 || MODE == 7 || MODE == 8) __builtin_nontemporal_store(s, y + c * 64 + lane);
        if (MODE == 10) y[(c * 64 + lane) & ((1 << 17) - 1)] = s;          // ring of 1 MiB: stays in the L2s
        if (MODE == 11) y[(c * 64 + lane) & ((1 << 23) - 1)] = s;          // ring of 64 MiB: beyond the L2s, inside the Infinity Cache
        if (MODE == 12) y[(c * 64 + lane) & ((1 << 20) - 1)] = s;          // ring of 8 MiB
        acc += s; prev = s;
    }
    if (acc == 1.2345) y[0] = acc;
}
int main()
{
    const int64_t nchunk = 165636900LL / 64;
    dbl2 *a; double *u, *y;
    CK(hipMalloc(&a, (size_t)nchunk * 6912)); CK(hipMalloc(&u, (size_t)nchunk * 512)); CK(hipMalloc(&y, (size_t)nchunk * 512));
    CK(hipMemset(a, 0, (size_t)nchunk * 6912)); CK(hipMemset(u, 0, (size_t)nchunk * 512)); CK(hipMemset(y, 0, (size_t)nchunk * 512));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *nm[15] = {"entries only", "+u read", "+y write", "+y write nt", "+y rmw", "+y rmw nt-store", "+y write, contiguous chunks per wave", "+y write nt, contiguous chunks", "+u read +y write nt", "+y write (previous chunk's value, before the wait)", "+y write into a 1 MiB ring", "+y write into a 64 MiB ring", "+y write into an 8 MiB ring", "entries only, PLAIN loads", "PLAIN loads +y write"};
    for (int grid : {256 * 8, (int)nchunk}) {
        for (int mode = 0; mode < 15; mode++) {
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
#define L(M) if (mode == M) hipLaunchKernelGGL(k<M>, dim3(grid), dim3(64), 0, 0, a, u, y, nchunk);
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14)
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const int extra = (mode == 0 ? 0 : (mode == 4 || mode == 5 || mode == 8) ? 1024 : 512);
            const double bytes = (double)nchunk * (6912 + extra);
            printf("grid %7d mode %d: %.3f ms  %.0f GB/s   %s\n", grid, mode, best, bytes / best / 1e6, nm[mode]);
        }
    }
    return 0;
}
