// What does the gather of x cost an irregular SpMV, apart from everything else?  E entries of (column, value) are streamed
// with 16-byte loads (4 entries per lane per load, the shape of k_spmv_jds), x[column] is gathered with 4-byte loads from an
// n-vector, value * x is summed per lane.  Three column distributions over the same E and n (fp32, n = 10^6 = 4 MB of x,
// E = 34 M -- the configs[4] stand-in of fixtures.irregular_matrix):
//   stream  : column = the entry's own index mod n rounded to the lane -> perfectly coalesced gathers (the streaming floor)
//   banded  : column uniform in [i*n/E - 2000, i*n/E + 2000]                 (an RCM-ordered FE matrix)
//   random  : column uniform in [0, n)                                       (no locality: every lane its own cache line)
// Build: hipcc --offload-arch=gfx950 -O3 gather_random.hip -o gather_random
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool GATHER> __global__ __launch_bounds__(256) void k(const i4 *__restrict__ col, const f4 *__restrict__ val, const float *__restrict__ x,
                                                                        float *__restrict__ y, long groups)
{
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long g0 = wave * 64 * U + lane;                   // a wave owns U consecutive 1-KiB pieces of both streams
    if (g0 >= groups) return;
    i4 c[U]; f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long g = g0 + 64 * u < groups ? g0 + 64 * u : g0;
        c[u] = __builtin_nontemporal_load(col + g);
        v[u] = __builtin_nontemporal_load(val + g);
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += v[u][e] * (GATHER ? x[c[u][e]] : (float)c[u][e]);
    y[wave * 64 + lane] = acc;
}

static inline uint32_t h32(uint64_t x) { x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return (uint32_t)x; }

int main()
{
    const long n = 1000000, E = 34 * 1000 * 1000, groups = E / 4;
    std::vector<int> hc(E);
    std::vector<float> hv(E, 1.0f), hx(n, 1.0f);
    int *col; float *val, *x, *y;
    CK(hipMalloc(&col, E * 4)); CK(hipMalloc(&val, E * 4)); CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, (groups + 4096) * 4));
    CK(hipMemcpy(val, hv.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int r = 0; r < 3; ++r) launch();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 20; ++r) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms / 20 * 1e3;
        printf("%-44s %7.1f us  = %5.2f TB/s on the %ld MB of the (column, value) streams\n", name, us, E * 8.0 / us / 1e6, E * 8 / 1000000);
    };
    constexpr int U = 4;
    const int blocks = (int)((groups + 256 * U - 1) / (256 * U));
    for (int kind = 0; kind < 3; ++kind) {
        for (long i = 0; i < E; ++i) {
            const long centre = i / 34;
            if (kind == 0) hc[i] = (int)((i / 4 % 64 + (i / 1024) * 64 + (i % 4) * 64 * 0) % n);   // lanes of a wave read 64 consecutive x
            else if (kind == 1) { long c = centre + (long)(h32(i) % 4001) - 2000; hc[i] = (int)((c % n + n) % n); }
            else hc[i] = (int)(h32(i) % n);
        }
        CK(hipMemcpy(col, hc.data(), E * 4, hipMemcpyHostToDevice));
        const char *nm[3] = {"coalesced gather", "banded gather (+-2000)", "random gather (4 MB of x)"};
        if (kind == 0) run("streams only (no gather)", [&] { hipLaunchKernelGGL((k<U, false>), dim3(blocks), dim3(256), 0, 0, (const i4 *)col, (const f4 *)val, x, y, groups); });
        run(nm[kind], [&] { hipLaunchKernelGGL((k<U, true>), dim3(blocks), dim3(256), 0, 0, (const i4 *)col, (const f4 *)val, x, y, groups); });
    }
    return 0;
}
