// What does one grid-wide hand-off of the single-launch Gram-Schmidt (k_mgs_fused, csrc/mik_kernels.h) cost, and what would it cost if
// all participants sat on ONE XCD and the slots only had to be coherent in that XCD's L2?  The skeleton of the kernel without its
// arithmetic: NW workgroups, H dependent hops; in a hop every workgroup publishes one slot, then reads all NW slots of the hop
// (polling while a slot is still "empty") and sums them.  Optionally every workgroup also loads its share of a `colbytes` column
// per hop, issued before the poll (the v_{i+1} prefetch).
//   mode 0: agent scope -- write-through store, L2-bypassing load (what k_mgs_fused does); workgroups over all XCDs
//   mode 1: one XCD (launch 8 x NW workgroups, those with blockIdx % 8 == 0 take part) -- plain store + "buffer_inv sc0" + plain load
//   mode 2: one XCD, agent-scope accesses (what the placement alone changes)
//   mode 3: one XCD, store sc0 + "buffer_inv sc0" + plain load;  mode 4: store sc0 + load sc0;  mode 5: store sc1 + "buffer_inv sc0" + plain load
//   mode 6: one XCD, store sc0 + load nt;  mode 7: store sc1 + load nt
//   mode 10: as mode 0 with four polls in flight per lane;  mode 11 / 12: as mode 0 with s_sleep 2 / 8 after a poll that found the slot empty
//   mode 8: all XCDs, slots in FINE-GRAINED device memory, agent-scope accesses;  mode 9: the same slots, plain store + load nt
// Build: hipcc --offload-arch=gfx950 -O3 xl_hop.hip -o xl_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long U;
static constexpr U EMPTY = ~0ull;

template <int MODE> __device__ __forceinline__ void slot_store(U *p, U v)
{
    if (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");      // no scope bits
    else if (MODE == 9) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 3 || MODE == 4 || MODE == 6) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");   // workgroup scope
    else if (MODE == 5 || MODE == 7) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");   // agent scope (write-through)
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE> __device__ __forceinline__ U slot_load(const U *p)
{
    if (MODE == 6 || MODE == 7 || MODE == 9) {                // a non-temporal load: is it served past the L1 every time?
        U v;
        asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    if (MODE == 4) {
        U v;
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    if (MODE == 1 || MODE == 3 || MODE == 5) {
        U v;                                                  // drop this CU's L1: the load is then served by the XCD's L2
        asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_hops(U *slots /* [H][NW] */, int NW, int H, const double *col, size_t colwords, int ncols, double *sink, U *result,
                                                int *err)
{
    constexpr bool ALL = MODE == 0 || MODE >= 8;          // workgroups over all XCDs
    if (!ALL && (blockIdx.x & 7u)) return;
    const int s = !ALL ? blockIdx.x >> 3 : blockIdx.x, t = threadIdx.x;
    __shared__ U sh;
    U carry = 0;
    double keep = 0.0;
    const size_t share = colwords / NW;                                  // doubles of a column per workgroup
    for (int h = 0; h < H; ++h) {
        double v[16];
        const double *c = col + (size_t)(h % ncols) * colwords + (size_t)s * share;
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (size_t)(t + 256 * q) < share ? c[t + 256 * q] : 0.0;     // the prefetch of the next column
        if (t == 0) slot_store<MODE>(slots + (size_t)h * NW + s, (U)(s + 1) * (U)(h + 1) + carry);
        U mine = 0;
        if (t < NW) {
            U b = EMPTY;
            if (MODE == 10) {             // four polls in flight, issued a fraction of a round trip apart: a slot that fills is seen by the next ISSUED load
                const U *q = slots + (size_t)h * NW + t;
                U p0 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_sleep(3);
                U p1 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_sleep(3);
                U p2 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_sleep(3);
                U p3 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; spin < (1 << 11); ++spin) {
                    if (p0 != EMPTY) { b = p0; break; }
                    p0 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (p1 != EMPTY) { b = p1; break; }
                    p1 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (p2 != EMPTY) { b = p2; break; }
                    p2 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (p3 != EMPTY) { b = p3; break; }
                    p3 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else
            for (int spin = 0; spin < (1 << 13); ++spin) {
                b = slot_load<MODE == 11 || MODE == 12 ? 0 : MODE>(slots + (size_t)h * NW + t);
                if (b != EMPTY) break;
                if (MODE == 11) __builtin_amdgcn_s_sleep(2);
                if (MODE == 12) __builtin_amdgcn_s_sleep(8);
            }
            if (b == EMPTY) { *err = 1; b = 0; }
            mine = b;
        }
        // sum over the workgroup (t < NW <= 256): wave shuffles, then LDS
        for (int off = 32; off; off >>= 1) mine += __shfl_down(mine, off);
        if (t == 0) sh = 0;
        __syncthreads();
        if ((t & 63) == 0) atomicAdd(&sh, mine);
        __syncthreads();
        carry = sh & 0xffff;                                             // the next hop depends on this one
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) keep += v[q];
    }
    if (keep == 12345.678) sink[0] = keep;
    if (t == 0) result[s] = carry;
}

int main(int argc, char **argv)
{
    const int H = 64, reps = 5;
    const int ncols = 32;
    for (size_t colbytes : {(size_t)0, (size_t)4 << 20}) {
        for (int NW : {32, 64, 128, 245}) {
            const size_t colwords = colbytes / 8;
            if (colwords / NW > 4096) continue;                          // 16 loads of 256 doubles per thread and hop at most
            U *slots, *result;
            double *col, *sink;
            int *err;
            hipMalloc(&slots, sizeof(U) * H * NW);
            U *fslots;
            hipExtMallocWithFlags((void **)&fslots, sizeof(U) * H * NW, hipDeviceMallocFinegrained);
            hipMalloc(&result, sizeof(U) * NW);
            hipMalloc(&col, colbytes * ncols + 64);
            hipMemset(col, 0, colbytes * ncols + 64);
            hipMalloc(&sink, 64);
            hipMalloc(&err, 4);
            hipMemset(err, 0, 4);
            for (int mode = 0; mode < 13; ++mode) {
                if (mode == 1 || mode == 3 || mode == 4 || mode == 5 || mode == 7 || mode == 9 || mode == 2 || mode == 8 || mode == 10) continue;      // measured: all of them time out (stale L1 lines)
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                float best = 1e30f;
                std::vector<U> r0(NW), r(NW);
                bool same = true;
                for (int rep = 0; rep < reps; ++rep) {
                    hipMemset(slots, 0xFF, sizeof(U) * H * NW);
                    hipMemset(fslots, 0xFF, sizeof(U) * H * NW);
                    hipDeviceSynchronize();
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k_hops<0>, dim3(NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 1) hipLaunchKernelGGL(k_hops<1>, dim3(8 * NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 2) hipLaunchKernelGGL(k_hops<2>, dim3(8 * NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 3) hipLaunchKernelGGL(k_hops<3>, dim3(8 * NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 4) hipLaunchKernelGGL(k_hops<4>, dim3(8 * NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 5) hipLaunchKernelGGL(k_hops<5>, dim3(8 * NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 6) hipLaunchKernelGGL(k_hops<6>, dim3(8 * NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 7) hipLaunchKernelGGL(k_hops<7>, dim3(8 * NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 8) hipLaunchKernelGGL(k_hops<8>, dim3(NW), dim3(256), 0, 0, fslots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 9) hipLaunchKernelGGL(k_hops<9>, dim3(NW), dim3(256), 0, 0, fslots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 10) hipLaunchKernelGGL(k_hops<10>, dim3(NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    else if (mode == 11) hipLaunchKernelGGL(k_hops<11>, dim3(NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    else hipLaunchKernelGGL(k_hops<12>, dim3(NW), dim3(256), 0, 0, slots, NW, H, col, colwords, ncols, sink, result, err);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                    hipMemcpy(r.data(), result, sizeof(U) * NW, hipMemcpyDeviceToHost);
                    if (rep == 0 && mode == 0) r0 = r;
                    for (int i = 0; i < NW; ++i) if (r[i] != r[0]) same = false;
                }
                int herr = 0;
                hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
                static U ref = 0;
                if (mode == 0) ref = r[0];
                printf("column %7zu B, %3d workgroups, mode %d: %6.2f us per launch of %d hops = %5.2f us per hop; all workgroups agree: %s; equals mode 0: %s; timeout: %d\n",
                       colbytes, NW, mode, best * 1e3, H, best * 1e3 / H, same ? "yes" : "NO", r[0] == ref ? "yes" : "NO", herr);
                fflush(stdout);
                hipMemset(err, 0, 4);
            }
            hipFree(slots); hipFree(result); hipFree(col); hipFree(sink); hipFree(err);
        }
    }
    return 0;
}
