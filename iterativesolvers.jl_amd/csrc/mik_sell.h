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

// Two rows per thread (128-thread workgroups): rows 2t and 2t+1 of the slice sit next to each other in every
// column-major line, so val / col stream as 16-byte / 8-byte loads -- half the vector-memory instructions.
// The fused dot keeps the (1, 1) shape of one value per ROW: a 64-row "virtual wave" is 32 lanes x 2 rows, the
// shuffle-down tree over rows (offsets 32..2) becomes offsets 16..1 over lanes on both values, the last step
// (row i += row i+1) adds the thread's own pair; then the 4 virtual-wave sums left to right, as block_tree_256.
template <typename T, bool FUSE_DOT, bool NT>
__global__ __launch_bounds__(MIK_BLOCK / 2) void k_spmv_sell2(int n, int nb, int map_mode, const int *__restrict__ blkptr,
                                                              const unsigned char *__restrict__ rlen, const int *__restrict__ col,
                                                              const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y,
                                                              T *__restrict__ seg_out, const int *__restrict__ done, int yvec)
{
    if (done && *done) return;
    constexpr int U = MIK_SELL_U;
    typedef T TV2 __attribute__((ext_vector_type(2)));
    typedef int IV2 __attribute__((ext_vector_type(2)));
    __shared__ T lds4[4];
    const int t = threadIdx.x;
    const int rb = spmv_block_map((int)blockIdx.x, nb, map_mode);
    const int r0 = rb * MIK_BLOCK + 2 * t;
    const int base = blkptr[rb];
    const int width = (blkptr[rb + 1] - base) / MIK_BLOCK;
    const int len0 = r0 < n ? (int)rlen[r0] : 0;
    const int len1 = r0 + 1 < n ? (int)rlen[r0 + 1] : 0;
    const T *__restrict__ vp = val + base + 2 * t;
    const int *__restrict__ cp = col + base + 2 * t;

    T acc0 = T(0), acc1 = T(0);
    for (int j0 = 0; j0 < width; j0 += U) {
        TV2 v[U];
        IV2 c[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int jj = min(j0 + q, width - 1);
            v[q] = ld_stream<NT>(reinterpret_cast<const TV2 *>(vp + (size_t)jj * MIK_BLOCK));
            c[q] = ld_stream<NT>(reinterpret_cast<const IV2 *>(cp + (size_t)jj * MIK_BLOCK));
        }
        T xa[U], xb[U];
#pragma unroll
        for (int q = 0; q < U; ++q) { xa[q] = x[c[q][0]]; xb[q] = x[c[q][1]]; }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (j0 + q < len0) { T p = v[q][0] * xa[q]; acc0 = acc0 + p; }
            if (j0 + q < len1) { T p = v[q][1] * xb[q]; acc1 = acc1 + p; }
        }
    }
    if (r0 + 1 < n && yvec) {
        TV2 o; o[0] = acc0; o[1] = acc1;
        st_stream<NT>(reinterpret_cast<TV2 *>(y + r0), o);
    } else {
        if (r0 < n) st_stream<NT>(y + r0, acc0);
        if (r0 + 1 < n) st_stream<NT>(y + r0 + 1, acc1);
    }
    if (FUSE_DOT) {
        T p0 = T(0), p1 = T(0);
        if (r0 < n) p0 = x[r0] * acc0;
        if (r0 + 1 < n) p1 = x[r0 + 1] * acc1;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
            p0 = p0 + __shfl_down(p0, off, 32);
            p1 = p1 + __shfl_down(p1, off, 32);
        }
        const T ws = p0 + p1;                      // valid in lane 0 of every 32-lane group = one 64-row virtual wave
        if ((t & 31) == 0) lds4[t >> 5] = ws;
        __syncthreads();
        if (t == 0) {
            T tot = lds4[0];
            tot = tot + lds4[1]; tot = tot + lds4[2]; tot = tot + lds4[3];
            seg_out[rb] = tot;
        }
    }
}

// Sliced-ELL with 8-bit column codes.  For banded / stencil operators the difference (column - row) takes few
// distinct values over the whole matrix; when there are at most 255 of them, every stored entry keeps its full
// value but its column index becomes a 1-byte code into a table of offsets (code 255 = padding).  Thread t of a
// slice owns 8-byte groups of codes ([row][j] layout, row length padded to a multiple of 8), so one 8-byte load
// brings the columns of 8 entries; the value lines are the ones of the plain sliced-ELL form.  Same products, same
// order, same bits -- 9 instead of 12 bytes per fp64 entry and no row-length array.
template <typename T, bool FUSE_DOT, bool NT>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_sell8(int n, int nb, int map_mode, const int *__restrict__ blkptr,
                                                          const int *__restrict__ cptr, const unsigned char *__restrict__ codes,
                                                          const int *__restrict__ dtab_g, int nd, const T *__restrict__ val,
                                                          const T *__restrict__ x, T *__restrict__ y, T *__restrict__ seg_out,
                                                          const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr int U = 8;
    __shared__ int dtab[256];
    __shared__ T lds4[4];
    const int t = threadIdx.x;
    dtab[t] = t < nd ? dtab_g[t] : 0;
    const int rb = spmv_block_map((int)blockIdx.x, nb, map_mode);
    const int r = rb * MIK_BLOCK + t;
    const int base = blkptr[rb];
    const int width = (blkptr[rb + 1] - base) / MIK_BLOCK;
    const int cb = cptr[rb];
    const int w8 = (cptr[rb + 1] - cb) / MIK_BLOCK;                // codes per row, a multiple of 8
    const T *__restrict__ vp = val + base + t;
    const unsigned long long *__restrict__ myc = reinterpret_cast<const unsigned long long *>(codes + (size_t)cb + (size_t)t * w8);

    // The streams of the first pass are issued BEFORE the barrier that publishes the offset table: they do not
    // depend on it, and for a stencil (width <= 8) they are the whole row.
    unsigned long long cw = 0;
    T v[U];
    if (width > 0) {
        cw = ld_stream<NT>(myc);
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = ld_stream<NT>(vp + (size_t)min(q, width - 1) * MIK_BLOCK);
    }
    __syncthreads();

    T acc = T(0);
    for (int j0 = 0; j0 < width;) {
        T xv[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int code = (int)((cw >> (8 * q)) & 255ull);
            xv[q] = x[code != 255 ? r + dtab[code] : 0];             // padding gathers from a valid address, never added
        }
#pragma unroll
        for (int q = 0; q < U; ++q)
            if (((cw >> (8 * q)) & 255ull) != 255ull) { T p = v[q] * xv[q]; acc = acc + p; }
        j0 += U;
        if (j0 < width) {
            cw = ld_stream<NT>(myc + (j0 >> 3));
#pragma unroll
            for (int q = 0; q < U; ++q) v[q] = ld_stream<NT>(vp + (size_t)min(j0 + q, width - 1) * MIK_BLOCK);
        }
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
