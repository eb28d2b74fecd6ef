// Micro-benchmark (development): what does a workgroup barrier cost a short 256-thread workgroup on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/barrier_cost scripts/micro/barrier_cost.hip && /tmp/barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(int mode, const int *__restrict__ ids, float *out)
{
    extern __shared__ float lds[];
    const int t = threadIdx.x;
    int v = 0;
    if (mode & 1) v = ids[blockIdx.x];                 // one scalar load per workgroup
    if (mode & 2) __syncthreads();
    if (mode & 4) { lds[t] = (float)v; __syncthreads(); v += (int)lds[(t + 64) & 255]; }
    if (mode & 8) { asm volatile("s_sleep 20"); }       // ~1280 cycles of "work" per wave
    if (v == 0x7fffffff) out[t] = 1.0f;
}

int main()
{
    const int wgs = 16384;
    int *ids; float *out;
    hipMalloc(&ids, wgs * 4); hipMalloc(&out, 1024);
    hipMemset(ids, 0, wgs * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ldss[] = {0, 4096, 16384, 28688, 65536};
    const int modes[] = {0, 1, 2, 3, 7, 8, 10, 11};
    for (int lds : ldss)
        for (int mode : modes) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds, 0, mode, ids, out);
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds, 0, mode, ids, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("lds %6d mode %2d : %7.2f us per launch (%d workgroups x 256)\n", lds, mode, ms / 20 * 1e3, wgs);
        }
    return 0;
}
