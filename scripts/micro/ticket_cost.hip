// Micro-benchmark (development): what does one device-scope ticket per workgroup cost a kernel of 32,768 short workgroups?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ticket_cost scripts/micro/ticket_cost.hip && /tmp/ticket_cost
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(int mode, unsigned *tickets, float *out, const float *in)
{
    const int t = threadIdx.x;
    float v = in[(blockIdx.x * 256 + t) & 0xffff];            // a little real work
    if (mode) {
        __syncthreads();
        if (t == 0) {
            unsigned *c = tickets + (mode == 2 ? (blockIdx.x & 7) * 64 : mode == 3 ? (blockIdx.x & 63) * 64 : 0);   // 1 / 8 / 64 counters
            const unsigned tk = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk == 0xffffffffu) out[0] = 1.0f;
        }
    }
    if (v == 12345.0f) out[t] = v;
}

int main()
{
    const int wgs = 32768;
    unsigned *tickets; float *out, *in;
    (void)hipMalloc(&tickets, 64 * 64 * 4); (void)hipMalloc(&out, 4096); (void)hipMalloc(&in, 65536 * 4);
    (void)hipMemset(tickets, 0, 64 * 64 * 4); (void)hipMemset(in, 0, 65536 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, mode, tickets, out, in);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, mode, tickets, out, in);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %7.2f us per launch of %d workgroups\n", mode, mode == 0 ? "no ticket" : mode == 1 ? "one counter" : mode == 2 ? "8 counters" : "64 counters",
               ms / 20 * 1e3, wgs);
    }
    return 0;
}
