// mik_sell.h -- sliced-ELL device layout of the operator (slice = one 256-row block) and its SpMV kernel.
//
// mul!(y, A, x) for operators whose rows within a 256-row block have similar lengths (stencils, banded
// matrices).  mik_csr_create re-lays the CSR out per row-block in COLUMN-MAJOR slices: entry j of the block's
// 256 rows is contiguous, so thread t (= row r0 + t) streams val[base + 256 j + t], col[base + 256 j + t] with
// fully coalesced loads, the gather x[col] is contiguous across the wave whenever neighbouring rows reference
// neighbouring columns, and every thread adds its own row's products in ascending column order from +0 --
// exactly the order Julia's CSC column scatter reaches that row (SparseArrays mul!, called at src/cg.jl:54,
// src/gmres.jl:287), i.e. bit-identical to the row-block CSR kernel (mik_spmv.h), with no LDS staging and no
// barrier.  Slices are padded to the block's longest row (padding entries are never added: j < len[row]); the
// layout is only built when padding stays below ~12 % and no row was split off as "long" (mik_csr_create).
//
// Bytes per launch vs CSR: no row pointer (4 B/row) but one length byte per row and the padding
// (256^3 Laplacian: +0.6 % entries) -- ~1.70 GB instead of 1.74 GB.
#ifndef MIK_SELL_H
#define MIK_SELL_H

#include "mik_internal.h"
#include "mik_spmv.h"

#ifdef __HIPCC__

constexpr int MIK_SELL_U = 8;     // entries per thread in flight per pass (7-point stencil: one pass)

template <typename T, bool FUSE_DOT, bool NT>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_sell(int n, int nb, int map_mode, const int *__restrict__ blkptr,
                                                         const unsigned char *__restrict__ rlen, const int *__restrict__ col,
                                                         const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y,
                                                         T *__restrict__ seg_out, const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr int U = MIK_SELL_U;
    __shared__ T lds4[4];
    const int t = threadIdx.x;
    const int rb = spmv_block_map((int)blockIdx.x, nb, map_mode);
    const int r = rb * MIK_BLOCK + t;
    const int base = blkptr[rb];
    const int width = (blkptr[rb + 1] - base) / MIK_BLOCK;     // longest row of this block
    const int len = r < n ? (int)rlen[r] : 0;
    const T *__restrict__ vp = val + base + t;
    const int *__restrict__ cp = col + base + t;

    T acc = T(0);
    for (int j0 = 0; j0 < width; j0 += U) {
        T v[U];
        int c[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int jj = min(j0 + q, width - 1);               // clamp: all U loads are unconditional (batched)
            v[q] = ld_stream<NT>(vp + (size_t)jj * MIK_BLOCK);
            c[q] = ld_stream<NT>(cp + (size_t)jj * MIK_BLOCK);
        }
        T xv[U];
#pragma unroll
        for (int q = 0; q < U; ++q) xv[q] = x[c[q]];
#pragma unroll
        for (int q = 0; q < U; ++q)
            if (j0 + q < len) { T p = v[q] * xv[q]; acc = acc + p; }
    }
    if (r < n) st_stream<NT>(y + r, acc);
    if (FUSE_DOT) {
        T p = T(0);
        if (r < n) p = x[r] * acc;
        T tot = block_tree_256(p, lds4);
        if (t == 0) seg_out[rb] = tot;
    }
}

#endif  // __HIPCC__
#endif  // MIK_SELL_H
