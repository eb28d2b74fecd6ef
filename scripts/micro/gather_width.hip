// Is a 64-lane gather priced per instruction or per byte?  y[i] = x[i-N*N] + x[i-N] + x[i] + x[i+N] + x[i+N*N] over 256^3
// doubles (interior rows only; the others copy x), one row per lane with 8-byte loads against two consecutive rows per lane
// with 16-byte loads.  Build: hipcc --offload-arch=gfx950 -O3 gather_width.hip -o gather_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// workgroup b -> block of rows: XCD b & 7 takes strips of S consecutive blocks (1/8 of a plane), as the library's SpMV does
__device__ __forceinline__ long strip(long b, long S) { const long xcd = b & 7, q = b >> 3; return ((q / S) * 8 + xcd) * S + q % S; }

template <int SLOTS> __global__ __launch_bounds__(256) void k_one(const double *__restrict__ x, double *__restrict__ y, long n, long N)
{
    const long i = strip(blockIdx.x, N / 8) * 256 + threadIdx.x;
    if (i >= n) return;
    const long NN = N * N;
    double a = x[i];
    if (i >= NN && i + NN < n) {
        if (SLOTS >= 3) { a += x[i - N]; a += x[i + N]; }
        if (SLOTS >= 5) { a += x[i - NN]; a += x[i + NN]; }
    }
    __builtin_nontemporal_store(a, y + i);
}

template <int SLOTS> __global__ __launch_bounds__(256) void k_two(const double *__restrict__ x, double *__restrict__ y, long n, long N)
{
    const long i = (strip(blockIdx.x, N / 16) * 256 + threadIdx.x) * 2;
    if (i >= n) return;
    const long NN = N * N;
    d2 a = *(const d2 *)(x + i);
    if (i >= NN && i + 1 + NN < n) {
        if (SLOTS >= 3) { a += *(const d2 *)(x + i - N); a += *(const d2 *)(x + i + N); }
        if (SLOTS >= 5) { a += *(const d2 *)(x + i - NN); a += *(const d2 *)(x + i + NN); }
    }
    __builtin_nontemporal_store(a, (d2 *)(y + i));
}

int main()
{
    const long N = 256, n = N * N * N;
    double *x, *y;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8));
    CK(hipMemset(x, 0, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int r = 0; r < 3; ++r) launch();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 20; ++r) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f us\n", name, ms / 20 * 1e3);
    };
    const int b1 = (int)((n + 255) / 256), b2 = (int)((n / 2 + 255) / 256);
    run("1 row/lane, 1 slot ", [&] { hipLaunchKernelGGL(k_one<1>, dim3(b1), dim3(256), 0, 0, x, y, n, N); });
    run("1 row/lane, 3 slots", [&] { hipLaunchKernelGGL(k_one<3>, dim3(b1), dim3(256), 0, 0, x, y, n, N); });
    run("1 row/lane, 5 slots", [&] { hipLaunchKernelGGL(k_one<5>, dim3(b1), dim3(256), 0, 0, x, y, n, N); });
    run("2 rows/lane, 1 slot ", [&] { hipLaunchKernelGGL(k_two<1>, dim3(b2), dim3(256), 0, 0, x, y, n, N); });
    run("2 rows/lane, 3 slots", [&] { hipLaunchKernelGGL(k_two<3>, dim3(b2), dim3(256), 0, 0, x, y, n, N); });
    run("2 rows/lane, 5 slots", [&] { hipLaunchKernelGGL(k_two<5>, dim3(b2), dim3(256), 0, 0, x, y, n, N); });
    return 0;
}
