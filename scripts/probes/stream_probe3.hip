// Development probe: does the sustained HBM read rate depend on the DATA being read?  One 16 GiB array, streamed by a
// persistent grid with 16-B non-temporal loads, filled with (0) zeros, (1) one repeated 64-bit constant, (2) random bits,
// (3) a CSR-like image: doubles from a set of 16 values + ascending int32 columns.  Each pattern runs for ~2 s so that a
// background `rocm-smi --showpower --showclocks` sampler (run_power_probe.sh) sees steady state.
// hipcc --offload-arch=gfx950 -O3 -o stream_probe3 stream_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <chrono>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void __launch_bounds__(256) k_fill(uint64_t *a, int64_t n, int mode)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x1234567ull; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        uint64_t v = 0;
        if (mode == 1) v = 0x3FD5555555555555ull;
        else if (mode == 2) v = z;
        else if (mode == 3) {
            // two thirds of the words: a double out of 16 hopping amplitudes / diagonal values; one third: two ascending int32 columns
            if (i % 3 != 2) { const double t = 0.1 * (double)((z >> 8) & 15) - 0.8; v = (uint64_t)__double_as_longlong(t); }
            else { const uint32_t c = (uint32_t)(i & 0x7ffffff); v = ((uint64_t)(c + 12870) << 32) | c; }
        }
        a[i] = v;
    }
}
__global__ void __launch_bounds__(256) k_read(const dbl2 *__restrict__ a, int64_t n2, double *out)
{
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const dbl2 r = __builtin_nontemporal_load(a + i); acc += r.x + r.y;
    }
    if (acc == 1.2345e-300) out[0] = acc;
}
int main()
{
    const int64_t bytes = (int64_t)16 << 30, n = bytes / 8, n2 = bytes / 16;
    uint64_t *a; double *out; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[4] = {"zeros", "one constant", "random bits", "csr-like image"};
    for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 4; mode++) {
        hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, a, n, mode); CK(hipDeviceSynchronize());
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_read, dim3(256 * 8), dim3(256), 0, 0, (const dbl2 *)a, n2, out);
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now(); double first = 0, last = 0; int it = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_read, dim3(256 * 8), dim3(256), 0, 0, (const dbl2 *)a, n2, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it == 0) first = ms; last = ms; it++;
        }
        const double ts = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        printf("t=%.1f pattern %-14s: first %.3f ms %.0f GB/s | after %d launches %.3f ms %.0f GB/s\n", ts, names[mode], first, bytes / first / 1e6, it, last, bytes / last / 1e6);
        fflush(stdout);
    }
    return 0;
}
