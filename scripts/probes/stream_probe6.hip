// Development probe 6: read/write PHASE SEPARATION across the whole device.  Persistent one-wave blocks read G chunks
// (results kept in registers), then write the G x 512 B of results; with SYNC the whole grid meets at a counter barrier
// before and after the write step, so that the memory system sees (almost) pure read phases and pure write phases.
// hipcc --offload-arch=gfx950 -O3 -o stream_probe6 stream_probe6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ bool grid_meet(unsigned *ctr, unsigned target, int lane)
{
    // every block adds one; leaves when the count reaches target (all blocks are resident: grid <= 8 per CU).  Bounded spin.
    bool ok = true;
    if (lane == 0) {
        __threadfence();
        atomicAdd(ctr, 1u);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(4); if (++spins > 2000000) { ok = false; break; } }
    }
    return __shfl((int)ok, 0) != 0;
}
template <int G, bool SYNC, bool WR> __global__ void __launch_bounds__(64) k(const dbl2 *__restrict__ a, double *__restrict__ y, int64_t nstep, unsigned *ctr, int *err)
{
    const int lane = threadIdx.x;
    double acc = 0.0; unsigned phase = 0;
    for (int64_t st = 0; st < nstep; st++) {
        const int64_t g = st * gridDim.x + blockIdx.x;     // group of G chunks
        double s[G];
#pragma unroll 4
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
        if (SYNC) { if (!grid_meet(ctr, ++phase * gridDim.x, lane)) { if (lane == 0) *err = 1; return; } }
        if (WR) {
            double *yo = y + g * G * 64;
#pragma unroll
            for (int q = 0; q < G; q += 2) { dbl2 v; v.x = s[q]; v.y = s[q + 1]; reinterpret_cast<dbl2 *>(yo + q * 64)[lane] = v; }
        }
        if (SYNC) { if (!grid_meet(ctr, ++phase * gridDim.x, lane)) { if (lane == 0) *err = 1; return; } }
    }
    if (acc == 1.2345) y[0] = acc;
}
int main()
{
    const int grid = 2048;
    dbl2 *a; double *y; unsigned *ctr; int *err;
    const int64_t nchunk_max = 165636900LL / 64;
    CK(hipMalloc(&a, (size_t)nchunk_max * 6912)); CK(hipMalloc(&y, (size_t)nchunk_max * 512)); CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4));
    CK(hipMemset(a, 0, (size_t)nchunk_max * 6912)); CK(hipMemset(y, 0, (size_t)nchunk_max * 512)); CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, int G, auto launch, bool wr) {
        const int64_t nstep = nchunk_max / ((int64_t)grid * G); const int64_t nchunk = nstep * grid * G;
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipMemset(ctr, 0, 4));
            CK(hipEventRecord(e0)); launch(nstep); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        const double bytes = (double)nchunk * (6912 + (wr ? 512 : 0));
        printf("%-44s steps %3lld  %.3f ms  %.0f GB/s%s\n", name, (long long)nstep, best, bytes / best / 1e6, herr ? "  BARRIER TIMEOUT" : "");
    };
#define R(G, SYNC, WR, NAME) run(NAME, G, [&](int64_t ns) { hipLaunchKernelGGL((k<G, SYNC, WR>), dim3(grid), dim3(64), 0, 0, a, y, ns, ctr, err); }, WR)
    R(32, false, false, "G=32 no write");
    R(32, false, true, "G=32 write, unsynchronised");
    R(32, true, false, "G=32 no write, grid barriers");
    R(32, true, true, "G=32 write between grid barriers");
    R(64, false, true, "G=64 write, unsynchronised");
    R(64, true, false, "G=64 no write, grid barriers");
    R(64, true, true, "G=64 write between grid barriers");
    R(16, true, true, "G=16 write between grid barriers");
    return 0;
}
