// Which XCD does workgroup b of a launch run on?  (HW_REG_XCC_ID, gfx942/gfx950.)  The single-launch Gram-Schmidt's XCD-local form
// (k_mgs_fused XL, csrc/mik_kernels.h) needs "b % 8 == 0 -> one XCD".
// Build: hipcc --offload-arch=gfx950 -O3 xcc_map.hip -o xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out, int spin)
{
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
    if (spin && (blockIdx.x & 7) == 0) for (int i = 0; i < 20000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main()
{
    unsigned *d, h[1024];
    hipMalloc(&d, sizeof(h));
    for (int spin = 0; spin < 2; ++spin)
        for (int nb : {16, 64, 1024}) {
            hipMemset(d, 0xFF, sizeof(h));
            hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, spin);
            hipMemcpy(h, d, sizeof(unsigned) * nb, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int b = 0; b < nb; ++b) if (h[b] != h[b & 7]) ++bad;
            printf("spin %d, %4d workgroups: xcc of blocks 0..15:", spin, nb);
            for (int b = 0; b < 16; ++b) printf(" %u", h[b]);
            printf("   blocks whose xcc differs from that of block b %% 8: %d\n", bad);
        }
    return 0;
}
