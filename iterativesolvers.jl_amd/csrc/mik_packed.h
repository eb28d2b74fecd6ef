// mik_packed.h -- dictionary-coded CSR ("CSR-VI/DU"): a lossless, opt-in second representation of
// the operator for matrices with few distinct values and few distinct (column - row) offsets --
// stencil operators such as the reference's own fixtures (test/laplace_matrix.jl: 2 values, 7
// offsets; benchmark/advection_diffusion.jl: 4 values, 7 offsets).
//
// Each stored entry becomes ONE 16-bit code (value index << 8 | offset index) instead of 8 + 4
// bytes; the two dictionaries (<= 256 entries each) sit in LDS.  y = A*x is still computed row by
// row, products in ascending column order with a rounded multiply and a rounded add each -- the
// result is bit-identical to the plain CSR kernel (and to the oracle); only the HBM traffic changes:
// 2 B instead of 12 B per entry.  mik_csr_pack() builds it; matrices that do not qualify keep the
// plain kernels.
//
// Kernel: one workgroup per 256-row block (one row per thread).  The block's codes are one contiguous
// range: staged into LDS with coalesced non-temporal 16-byte loads (8 codes per lane); every thread
// then walks its own row: code -> (value, column = row + offset) from the LDS dictionaries, gathers
// x[column] (8 gathers in flight), and folds the products in order.
#pragma once
#include "mik_internal.h"
#include "mik_spmv.h"

#ifdef __HIPCC__

constexpr int MIK_PACK_TILE = 8192;     // codes staged per pass (16 KB of LDS)

typedef unsigned int mik_u32x4 __attribute__((ext_vector_type(4)));

template <typename T, bool FUSE_DOT>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_packed(int n, int nb, int map_mode, const int *__restrict__ rowptr,
                                                           const unsigned short *__restrict__ codes, const T *__restrict__ vtab_g,
                                                           const int *__restrict__ dtab_g, int nv, int nd, const T *__restrict__ x,
                                                           T *__restrict__ y, T *__restrict__ seg_out, const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr int TILE = MIK_PACK_TILE;
    __shared__ __attribute__((aligned(16))) unsigned short lc[TILE];
    __shared__ T vtab[256];
    __shared__ int dtab[256];
    __shared__ T lds4[4];

    const int t = threadIdx.x;
    const int rb = spmv_block_map((int)blockIdx.x, nb, map_mode);
    const int r0 = rb * MIK_BLOCK;
    const int r = r0 + t;
    int ks = 0, ke = 0;
    if (r < n) { ks = rowptr[r]; ke = rowptr[r + 1]; }
    const int kb = rowptr[r0] & ~7;                    // 16-byte aligned start of the code range
    const int kend = rowptr[min(r0 + MIK_BLOCK, n)];
    if (t < nv) vtab[t] = vtab_g[t];
    if (t < nd) dtab[t] = dtab_g[t];

    T acc = T(0);
    for (int kc = kb; kc < kend; kc += TILE) {
        const int cnt = min(TILE, kend - kc);
#pragma unroll
        for (int i = 0; i < TILE / (MIK_BLOCK * 8); ++i) {
            const int j = 8 * (t + MIK_BLOCK * i);
            if (j < cnt)    // reads past kend stay inside the padded allocation
                *reinterpret_cast<mik_u32x4 *>(&lc[j]) = __builtin_nontemporal_load(reinterpret_cast<const mik_u32x4 *>(codes + kc + j));
        }
        __syncthreads();
        int a = max(ks, kc) - kc;
        int len = min(ke, kc + cnt) - kc - a;
        while (len > 0) {
            T p[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned code = lc[min(a + i, TILE - 1)];
                const T v = vtab[code >> 8];
                int c = r + dtab[code & 255u];
                c = (i < len) ? c : r;                 // keep the speculative gather inside x
                p[i] = v * x[c];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < len) acc = acc + p[i];
            a += 8;
            len -= 8;
        }
        __syncthreads();
    }
    if (r < n) __builtin_nontemporal_store(acc, y + r);
    if (FUSE_DOT) {
        T p = T(0);
        if (r < n) p = x[r] * acc;
        T tot = block_tree_256(p, lds4);
        if (t == 0) seg_out[rb] = tot;
    }
}

#endif  // __HIPCC__
