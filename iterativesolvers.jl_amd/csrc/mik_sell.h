// mik_sell.h -- sliced device layouts of banded / stencil operators (slice = one 256-row block) and their SpMV kernels:
//               per-slice offsets + row masks (k_spmv_sdia) and the slice-constant forms (k_spmv_sdiab, k_spmv_sdiab2; up to 32
//               offsets per slice: k_spmv_sdiaw, k_spmv_sdiaw2).  Long near-uniform rows: mik_jds.h.
//
// mul!(y, A, x) (SparseArrays mul!, called at src/cg.jl:54, src/gmres.jl:287).  mik_csr_create re-lays the CSR out per row-block
// in COLUMN-MAJOR slices: entry j of the block's 256 rows is contiguous, so thread t (= row r0 + t) streams
// val[base + 256 j + t] with fully coalesced loads, the gather of x is contiguous across the wave whenever neighbouring rows
// reference neighbouring columns, and every thread adds its own row's products in ascending column order from +0 -- exactly
// the order Julia's CSC column scatter reaches that row, i.e. bit-identical to the row-block CSR kernels (mik_spmv.h), with
// no LDS staging and no barrier.  Slices are padded to the block's longest row (padding entries are never added); the
// layouts are only built when padding stays below ~12 % and no row was split off as "long" (mik_csr_create).
// (Round 1's plain sliced-ELL kernel with 4-byte columns and its 8-bit-column-code variant were superseded in round 3 by the
// jagged slices of mik_jds.h -- faster on every FE operator measured -- and by the wide slice-constant form below.)
#ifndef MIK_SELL_H
#define MIK_SELL_H

#include "mik_internal.h"
#include "mik_spmv.h"

#ifdef __HIPCC__

// Sliced-ELL with PER-SLICE offsets and per-row presence masks ("sliced diagonal" form).  When the rows of a
// 256-row slice together use at most 8 distinct (column - row) offsets -- any stencil on a structured grid -- the
// slice stores those offsets once (in the order the rows sum them: a common super-sequence of the rows' entry
// orders, built at upload), every row keeps one value slot per offset (zero where the row has
// no such entry, e.g. at grid boundaries) and one mask byte saying which slots are real.  The column of slot q
// is then row + offs[q] with offs[q] wave-uniform: the gather of x no longer waits for a per-row index load (it is
// issued together with the value stream), needs no table lookup, and the index data shrinks from 8 bytes to 1
// byte per row.  A row's real slots are visited in its own entry order from +0 and absent slots are skipped,
// so the sum is bit-identical to the other layouts.
//
// Runs of offsets (o - 1, o, o + 1) in consecutive slots -- the line neighbours of a stencil -- are served by ONE
// gather: the centre value moves one lane up / down the wave (shuffles), and only the wave's first and last lane
// fetch their outer neighbour themselves.  The load path of the CU (64 B/clk), not HBM, is what the 7 gathers of a
// 7-point row cost (measured: gathers from one and the same address are as slow as the real ones), so every
// gather avoided counts.  `trio[slice]` = first slot of such a run, or -1.
// (Measured and dropped: 2 or 4 neighbouring slices per 512- / 1024-thread workgroup passing their centre values
// through LDS, which also removes the +-N_x gathers of the inner slices -- 225 / 255 us vs 219 us: the barrier and
// the larger workgroups cost more than the gathers they save.)
// One row of a slice; TRI (compile time) = first slot of the (o-1, o, o+1) run served by shuffles, -1 = none.
// CV = true: the slot's value is the same for every row of the slice that has it ("slice-constant slots", k_spmv_sdiac):
// vp then points at the slice's 8 values and v[q] is a wave-uniform scalar load instead of a per-row stream.
template <typename T, bool NT, int TRI, bool CV>
__device__ __forceinline__ T sdia_row(int r, int n, int ncols, int ns, const int *__restrict__ so, const T *__restrict__ vp,
                                      const unsigned char *__restrict__ mask, const T *__restrict__ x, T &xr, bool &have_xr)
{
    constexpr int U = 8;
    T v[U], xv[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
        v[q] = T(0);
        xv[q] = T(0);
        if (q < ns) {                                                // slice-uniform: slots the slice does not have cost nothing
            v[q] = CV ? vp[q] : ld_stream<NT>(vp + (size_t)q * MIK_BLOCK);
            if (TRI < 0 || (q != TRI && q != TRI + 2))
                xv[q] = x[min(max(r + so[q], 0), ncols - 1)];        // absent slots gather from a valid address
        }
    }
    if (TRI >= 0) {
        const int oc = so[TRI + 1];
        const T c = xv[TRI + 1];
        T lo = __shfl_up(c, 1), hi = __shfl_down(c, 1);
        const int lane = threadIdx.x & 63;
        if (lane == 0 || lane == 63) {
            const T e = x[min(max(r + oc + (lane == 0 ? -1 : 1), 0), ncols - 1)];
            if (lane == 0) lo = e; else hi = e;
        }
        xv[TRI >= 0 ? TRI : 0] = lo;
        xv[TRI >= 0 ? TRI + 2 : 0] = hi;
        if (oc == 0) { xr = c; have_xr = true; }
    }
    const int m = r < n ? (int)mask[r] : 0;
    T acc = T(0);
#pragma unroll
    for (int q = 0; q < U; ++q)
        if ((m >> q) & 1) { T p = v[q] * xv[q]; acc = acc + p; }
    return acc;
}

template <typename T, bool FUSE_DOT, bool NT, bool CV>
__device__ __forceinline__ void spmv_sdia_body(int n, int ncols, int rb0, int nb, int map_mode, const int *__restrict__ blkptr,
                                               const int *__restrict__ offs, const int *__restrict__ trio,
                                               const unsigned char *__restrict__ mask, const T *__restrict__ val,
                                               const T *__restrict__ x, T *__restrict__ y, T *__restrict__ seg_out)
{
    constexpr int U = 8;
    __shared__ T lds4[4];
    const int t = threadIdx.x;
    const int rb = rb0 + spmv_block_map((int)blockIdx.x, nb, map_mode);   // this launch covers row-blocks [rb0, rb0 + nb)
    const int r = rb * MIK_BLOCK + t;
    const int base = blkptr[rb];
    const int ns = (blkptr[rb + 1] - base) / MIK_BLOCK;            // offsets used by this slice, <= 8
    const int tri = trio[rb];
    const int *__restrict__ so = offs + (size_t)rb * U;
    const T *__restrict__ vp = CV ? val + (size_t)rb * U : val + base + t;

    T acc = T(0);
    T xr = T(0);
    bool have_xr = false;
    if (ns > 0) {
        switch (tri) {                                              // slice-uniform: one specialised path per run position
        case 0: acc = sdia_row<T, NT, 0, CV>(r, n, ncols, ns, so, vp, mask, x, xr, have_xr); break;
        case 1: acc = sdia_row<T, NT, 1, CV>(r, n, ncols, ns, so, vp, mask, x, xr, have_xr); break;
        case 2: acc = sdia_row<T, NT, 2, CV>(r, n, ncols, ns, so, vp, mask, x, xr, have_xr); break;
        case 3: acc = sdia_row<T, NT, 3, CV>(r, n, ncols, ns, so, vp, mask, x, xr, have_xr); break;
        case 4: acc = sdia_row<T, NT, 4, CV>(r, n, ncols, ns, so, vp, mask, x, xr, have_xr); break;
        case 5: acc = sdia_row<T, NT, 5, CV>(r, n, ncols, ns, so, vp, mask, x, xr, have_xr); break;
        default: acc = sdia_row<T, NT, -1, CV>(r, n, ncols, ns, so, vp, mask, x, xr, have_xr); break;
        }
    }
    if (r < n) st_stream<NT>(y + r, acc);
    if (FUSE_DOT) {
        T p = T(0);
        if (r < n) p = (have_xr ? xr : x[r]) * acc;
        T tot = block_tree_256(p, lds4);
        if (t == 0) seg_out[rb] = tot;
    }
}

template <typename T, bool FUSE_DOT, bool NT>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_sdia(int n, int ncols, int rb0, int nb, int map_mode, const int *__restrict__ blkptr,
                                                         const int *__restrict__ offs, const int *__restrict__ trio,
                                                         const unsigned char *__restrict__ mask, const T *__restrict__ val,
                                                         const T *__restrict__ x, T *__restrict__ y, T *__restrict__ seg_out,
                                                         const int *__restrict__ done)
{
    if (done && *done) return;
    spmv_sdia_body<T, FUSE_DOT, NT, false>(n, ncols, rb0, nb, map_mode, blkptr, offs, trio, mask, val, x, y, seg_out);
}

// Slice-CONSTANT slots: in every 256-row slice, all rows that have slot q carry the same value bits there -- any
// constant-coefficient stencil (the Laplacian and advection-diffusion fixtures of the reference; a rank's block of them).
// A slice is then described by a PATTERN {number of slots, run position, <= 8 offsets, <= 8 values}; equal patterns are
// stored once (a 256^3 Laplacian has 9: interior lines and the faces / edges of the grid) and a slice carries a 4-byte
// pattern index, a row one mask byte: the operator shrinks from 57 bytes to ~1 byte per row and the SpMV moves 17 B per
// row (mask + x + y) instead of 73.  The pattern table stays hot in the scalar cache, so the only per-slice metadata
// that comes from memory is its index (the per-slice offset / value arrays of a first version were two dependent HBM
// misses per workgroup: 117 us).  Same products (value x gathered x, rounded), same order, same bits as
// every other layout; constancy is decided at upload on value BIT patterns (not numbers), so -0.0 / NaN payloads cannot
// alias.  Anything with varying coefficients keeps the per-row value slots of k_spmv_sdia.
#ifndef MIK_SDIAC_G
#define MIK_SDIAC_G 2       // slices per workgroup of k_spmv_sdiac
#endif
template <typename T> struct SdiaPattern {
    int ns, tri;        // slots used; first slot of an (o - 1, o, o + 1) run or -1
    int cq, dfull;      // the slot with offset 0 (-1: none); 1 = every row of the slice has it
    int off[8];
    int soff[8];        // (off[q] + koff) * sizeof(T): the scalar byte offsets of k_spmv_sdiab
    T val[8];
};

// What k_spmv_sdiab needs to ISSUE a slice's gathers, per slice and in one 64-byte scalar load: the scalar byte offsets and the
// shape of the slice's pattern.  With only a pattern index per slice the gathers waited for two dependent scalar loads
// (index, then pattern); the kernel is latency-bound (a wave lives ~4 us, three memory round trips), so the level counts.
// The values are still read from the pattern table -- they are not needed before the gathers return.
struct SdiaSliceRec {
    int soff[8];
    int ns, cq, dfull, pid;
    int pad[4];
};

template <typename T, bool FUSE_DOT, bool NT, int G>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_sdiac(int n, int ncols, int rb0, int nb, int map_mode, const int *__restrict__ pat_id,
                                                          const SdiaPattern<T> *__restrict__ pats, const unsigned char *__restrict__ mask,
                                                          const T *__restrict__ x, T *__restrict__ y, T *__restrict__ seg_out,
                                                          const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr int U = 8;
    __shared__ T lds[G][4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // this workgroup takes G slices: virtual blocks blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x is a multiple of 8,
    // so all of them carry this workgroup's XCD in the strip map); this launch covers row-blocks [rb0, rb0 + nb)
    int rb[G], r[G], m[G];
    const SdiaPattern<T> *__restrict__ pt[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int vb = (int)blockIdx.x + g * (int)gridDim.x;
        rb[g] = vb < nb ? rb0 + spmv_block_map(vb, nb, map_mode) : -1;
        r[g] = rb[g] >= 0 ? rb[g] * MIK_BLOCK + t : n;
        pt[g] = pats + (rb[g] >= 0 ? pat_id[rb[g]] : 0);
        m[g] = r[g] < n ? (int)mask[r[g]] : 0;
    }
    T xv[G][U];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int q = 0; q < U; ++q) {
            xv[g][q] = T(0);
            if (rb[g] >= 0 && q < pt[g]->ns) xv[g][q] = x[min(max(r[g] + pt[g]->off[q], 0), ncols - 1)];   // absent slots gather a valid address
        }
    // Measured and dropped: serving the +-1 neighbours by wave shuffles with the run position read from the pattern (145 us
    // instead of 112 us: the selects cost more than two gathers); two rows per thread with 16-byte gathers of x, 2-byte mask
    // loads and 16-byte stores (332 us: the 8-byte-aligned 16-byte gathers are far slower than two 8-byte ones).
    T p[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        T acc = T(0), xr = T(0);
        bool have = false;
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (rb[g] >= 0 && q < pt[g]->ns && pt[g]->off[q] == 0) { xr = xv[g][q]; have = true; }
            if ((m[g] >> q) & 1) { T pr = pt[g]->val[q] * xv[g][q]; acc = acc + pr; }
        }
        if (r[g] < n) st_stream<NT>(y + r[g], acc);
        p[g] = T(0);
        if (FUSE_DOT && r[g] < n) p[g] = (have ? xr : x[r[g]]) * acc;
    }
    if (FUSE_DOT) {                                   // one partial per slice: wave tree, then the 4 wave sums left to right
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const T ws = wave_tree(p[g]);
            if (lane == 0) lds[g][w] = ws;
        }
        __syncthreads();
        if (t < G) {
            const int vb = (int)blockIdx.x + t * (int)gridDim.x;
            if (vb < nb) {
                T tot = lds[t][0];
                tot = tot + lds[t][1]; tot = tot + lds[t][2]; tot = tot + lds[t][3];
                seg_out[rb0 + spmv_block_map(vb, nb, map_mode)] = tot;
            }
        }
    }
}

// The slice-constant layout through BUFFER loads.  k_spmv_sdiac executes ~180 vector-ALU instructions per row for 7
// multiply-adds (address clamps so that absent slots gather a valid element, mask tests and selects around every add,
// 64-bit address arithmetic).
// Here x is read through a buffer descriptor: the hardware adds a per-slot SCALAR offset (the slot's column offset) to
// a per-row vector offset and returns ZERO for a vector offset outside the descriptor's range -- so an absent slot just
// ORs all-ones into its row offset (one v_bfe of the inverted mask, one v_or), reads 0.0, and contributes value * 0 =
// +-0, which leaves the sum bit for bit as it was (the sum starts from +0 and +0 + -0 = +0, any other acc + +-0 = acc).
// No clamp, no select, no 64-bit add: 4 VALU per slot (bfe, or, mul, add).  Used when every pattern value is finite
// (Inf * 0 would be NaN), and row and slot offsets fit the 32-bit fields (mik_csr_create decides: sdia_buf_ok).
// The descriptor's base is x - koff elements (koff = the most negative slot offset of the operator), so every scalar
// offset (off[q] + koff) * sizeof(T) is non-negative; addresses below x are formed but never read.
template <typename T> __device__ __forceinline__ T buffer_gather(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff);
template <> __device__ __forceinline__ double buffer_gather<double>(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, soff, 0));
}
template <> __device__ __forceinline__ float buffer_gather<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff)
{
    const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, soff, 0);
    return __builtin_bit_cast(float, v);
}

template <typename T> __device__ __forceinline__ void buffer_put(__amdgpu_buffer_rsrc_t rs, unsigned voff, T v, int aux);
template <> __device__ __forceinline__ void buffer_put<double>(__amdgpu_buffer_rsrc_t rs, unsigned voff, double v, int aux)
{
    typedef unsigned mik_u32x2 __attribute__((ext_vector_type(2)));
    if (aux) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(mik_u32x2, v), rs, (int)voff, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(mik_u32x2, v), rs, (int)voff, 0, 0);
}
template <> __device__ __forceinline__ void buffer_put<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, float v, int aux)
{
    if (aux) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)voff, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)voff, 0, 0);
}

// spmv_block_map for strips of a power-of-two number of row-blocks: P = 8 << sshift row-blocks per plane, `nfull` =
// the row-blocks in whole planes ((nb / P) * P, computed by the host); sshift < 0: identity
__device__ __forceinline__ int spmv_block_map_shift(int b, int nfull, int sshift)
{
    if (sshift >= 0 && b < nfull) {
        const int xcd = b & 7, q = b >> 3;
        return ((q >> sshift) << (sshift + 3)) + (xcd << sshift) + (q & ((1 << sshift) - 1));
    }
    return b;
}

// The slot-by-slot form executes ~400 scalar instructions per wave ("does the slice have slot q", "is it the diagonal",
// offset arithmetic).  The common (slots, centre slot) CLASS of the operator is therefore compiled in: template NS / CQ; a
// workgroup whose slices all have NS slots with the diagonal in slot CQ runs straight-line code (per slot: v_bfe, v_or,
// buffer_load, v_mul, v_add), any other workgroup the slot-by-slot path.  Worth 5 us of 91 at 256^3 -- the launch turned out
// to wait for its streamed mask bytes, not for its instruction stream.
constexpr int MIK_SDIAB_NCLS = 4;   // 0: none; 1: 7 slots, centre 3 (3-D 7-point); 2: 5 slots, centre 2 (2-D 5-point); 3: 3 slots, centre 1
__host__ __device__ constexpr int mik_sdiab_cls_ns(int c) { return c == 1 ? 7 : c == 2 ? 5 : c == 3 ? 3 : 0; }
__host__ __device__ constexpr int mik_sdiab_cls_cq(int c) { return c == 1 ? 3 : c == 2 ? 2 : c == 3 ? 1 : -1; }

template <typename T, bool FUSE_DOT, bool NT, int G, int NS, int CQ>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_sdiab(int n, int ncols, int koff, int rb0, int nb, int nfull, int sshift, int skip_at, int skip_len, const SdiaSliceRec *__restrict__ recs,
                                                          const SdiaPattern<T> *__restrict__ pats, const unsigned char *__restrict__ mask,
                                                          const T *__restrict__ x, T *__restrict__ y, T *__restrict__ seg_out,
                                                          const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr int U = 8;
    constexpr unsigned ES = (unsigned)sizeof(T);
    __shared__ T lds[G][4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // rows are addressed by a 32-bit byte offset; an offset of all ones is out of the descriptor's range (reads 0)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((uintptr_t)x - (uintptr_t)koff * ES), (short)0,
                                                                        (int)0xFFFFFFF0u, (int)0x00020000);
    // the row masks and y go through descriptors as well: a row past the end reads mask 0 (= no slot at all: every gather out
    // of range, sum +0) and its store is dropped by the range check -- no "row < n" branch anywhere on the main path
    const __amdgpu_buffer_rsrc_t ms = __builtin_amdgcn_make_buffer_rsrc((void *)mask, (short)0, n, (int)0x00020000);
    const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc((void *)y, (short)0, (int)((unsigned)n * ES), (int)0x00020000);
    int rb[G], minv[G], rr[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {                       // virtual blocks blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8)
        const int vb = min((int)blockIdx.x + g * (int)gridDim.x, nb - 1);   // past the end: the last block once more (same bits)
        rb[g] = rb0 + spmv_block_map_shift(vb, nfull, sshift);
        if (rb[g] >= skip_at) rb[g] += skip_len;        // a launch over the row-blocks OUTSIDE [skip_at, skip_at + skip_len): the boundary rows of a rank
        rr[g] = rb[g] * MIK_BLOCK + t;
        // (the masks are read with the default cache policy: 1 B per row stays in the Infinity Cache from one SpMV to the next;
        //  streamed past the caches they cost the launch 14 us)
        minv[g] = ~(int)__builtin_amdgcn_raw_buffer_load_b8(ms, rr[g], 0, 0);               // bit q set: this row has no slot q
    }
    SdiaSliceRec rc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) rc[g] = recs[rb[g]];
    const SdiaPattern<T> *__restrict__ pt[G];
#pragma unroll
    for (int g = 0; g < G; ++g) pt[g] = pats + rc[g].pid;
    // The classes also have the slots around the centre address the row's LANE neighbours (columns row - 1 and row + 1): the
    // wave then loads x[row] once, range-checked and whatever the masks say, and hands it to the lanes on either side (two DPP
    // moves per 32 bits); only lanes 0 and 63 fetch their outer neighbour -- 5 full gathers per row of the 3-D stencil, not 7.
    constexpr bool LN = NS >= 3 && CQ >= 1 && CQ + 1 < NS;
    bool fast = NS > 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        fast = fast & (rc[g].ns == NS) & (rc[g].cq == CQ);
        if (LN) fast = fast & (rc[g].soff[CQ > 0 ? CQ - 1 : 0] + (int)ES == rc[g].soff[CQ >= 0 ? CQ : 0]) & (rc[g].soff[CQ >= 0 ? CQ : 0] + (int)ES == rc[g].soff[CQ + 1 < U ? CQ + 1 : 0]);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) asm volatile("" : "+v"(minv[g]));       // all masks have arrived before the first gather is issued
    T acc[G], xr[G];
    if (fast) {
        constexpr int NQ = NS > 0 ? NS : 1, CC = CQ >= 0 ? CQ : 0;
        T xv[G][NQ];
        if (LN) {
            // x[row] through a descriptor over the ROWS (a row past the end reads +0), the outer neighbours through one over the
            // COLUMNS: a row-partitioned block has more columns than rows, and its last row's upper neighbour is a halo entry
            const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc((void *)x, (short)0, (int)((unsigned)n * ES), (int)0x00020000);
            const __amdgpu_buffer_rsrc_t xw = __builtin_amdgcn_make_buffer_rsrc((void *)x, (short)0, (int)((unsigned)ncols * ES), (int)0x00020000);
            T xc[G], xe[G], xf[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const unsigned rowoff = (unsigned)rr[g] * ES;
                xc[g] = buffer_gather<T>(xs, rowoff, 0);
                xe[g] = T(0);
                const bool top = lane == 63 || rr[g] + 1 >= n;               // the lane above is another wave's, or has no row
                if (lane == 0 || top) {                                      // fetch that neighbour (or none: out of range, +0)
                    const unsigned ab = (unsigned)(top ? __builtin_amdgcn_sbfe(minv[g], CC + 1, 1) : __builtin_amdgcn_sbfe(minv[g], CC - 1, 1));
                    xe[g] = buffer_gather<T>(xw, (top ? rowoff + ES : rowoff - ES) | ab, 0);
                }
                xf[g] = T(0);
                if (lane == 0 && top)                                        // (a wave whose first row is the last one needs both)
                    xf[g] = buffer_gather<T>(xw, (rowoff - ES) | (unsigned)__builtin_amdgcn_sbfe(minv[g], CC - 1, 1), 0);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (q < CC - 1 || q > CC + 1)
                        xv[g][q] = buffer_gather<T>(rs, rowoff | (unsigned)__builtin_amdgcn_sbfe(minv[g], q, 1), rc[g].soff[q]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                T lo = lane_next<true>(xc[g]), hi = lane_next<false>(xc[g]);
                const bool top = lane == 63 || rr[g] + 1 >= n;
                if (top) hi = xe[g];
                if (lane == 0) lo = top ? xf[g] : xe[g];
                xv[g][CC - 1] = ((minv[g] >> (CC - 1)) & 1) ? T(0) : lo;      // absent slots contribute value * +0, as through the gather
                xv[g][CC] = ((minv[g] >> CC) & 1) ? T(0) : xc[g];
                xv[g][CC + 1] = ((minv[g] >> (CC + 1)) & 1) ? T(0) : hi;
                xr[g] = xc[g];
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const unsigned rowoff = (unsigned)rr[g] * ES;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    xv[g][q] = buffer_gather<T>(rs, rowoff | (unsigned)__builtin_amdgcn_sbfe(minv[g], q, 1), rc[g].soff[q]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                xr[g] = xv[g][CC];
                if (FUSE_DOT && !rc[g].dfull && ((minv[g] >> CC) & 1) && rr[g] < n) xr[g] = x[rr[g]];   // a row without a diagonal entry
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            T a = T(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) { T pr = pt[g]->val[q] * xv[g][q]; a = a + pr; }
            acc[g] = a;
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned rowoff = (unsigned)rr[g] * ES;
            const int ns = rc[g].ns, cq = rc[g].cq;
            T xv[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                xv[q] = T(0);
                if (q < ns) xv[q] = buffer_gather<T>(rs, rowoff | (unsigned)__builtin_amdgcn_sbfe(minv[g], q, 1), rc[g].soff[q]);
            }
            T a = T(0), c = T(0);
#pragma unroll
            for (int q = 0; q < U; ++q) {
                if (q < ns) {
                    if (q == cq) c = xv[q];
                    T pr = pt[g]->val[q] * xv[q];
                    a = a + pr;
                }
            }
            acc[g] = a;
            xr[g] = c;
            if (FUSE_DOT && rr[g] < n && (cq < 0 || ((minv[g] >> cq) & 1))) xr[g] = x[rr[g]];
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        buffer_put<T>(ys, (unsigned)rr[g] * ES, acc[g], NT ? 2 : 0);
        if (FUSE_DOT) {
            const T pd = xr[g] * acc[g];                // a row past the end: 0 * +0
            const T ws = wave_tree(pd);
            if (lane == 0) lds[g][w] = ws;
        }
    }
    if (FUSE_DOT) {                                   // one partial per slice: the 4 wave sums left to right
        __syncthreads();
        if (t < G) {
            T tot = lds[t][0];
            tot = tot + lds[t][1]; tot = tot + lds[t][2]; tot = tot + lds[t][3];
            int rbt = rb[0];
#pragma unroll
            for (int g = 1; g < G; ++g)
                if (t == g) rbt = rb[g];
            seg_out[rbt] = tot;
        }
    }
}

// ---- two consecutive rows per lane ---------------------------------------------------------------------------------------
// A 64-lane gather is priced per INSTRUCTION on this GPU, not per byte (scripts/micro/gather_width.hip: y = sum of 5 stencil
// neighbours over 256^3 doubles, 50.8 us with one row per lane and 8-byte loads, 33.9 us -- the copy floor -- with two rows
// per lane and 16-byte loads).  Here lane l of a wave owns rows 2l and 2l + 1 of a 128-row piece: one 16-byte load per
// slot serves both rows (their columns are consecutive), the slots around the centre come from the centre load of the lane
// itself and of its neighbours (DPP), and a 16-byte store writes both sums.  A workgroup covers two consecutive slices (512
// rows); every wave lies in one slice.  Needs an even n; a wave whose slice is not of the compiled-in class, or in which a
// row pair differs in the presence of a gathered slot, runs the slot-by-slot path on its two rows.
// dot(x, y) partials keep the shape of k_spmv_sdiab bit for bit: per 64 rows the shuffle-down tree (row r + 32, + 16, ...
// + 1) -- rows r and r + 2k sit k lanes apart, rows r and r + 1 in one lane -- so lanes 0..31 and 32..63 each run the five
// upper levels on both of their values and add the two at the end; then the 4 sums of a slice left to right.
template <typename T> struct Pair2 { T a, b; };
template <typename T> __device__ __forceinline__ Pair2<T> buffer_gather2(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff);
template <> __device__ __forceinline__ Pair2<double> buffer_gather2<double>(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff)
{
    return __builtin_bit_cast(Pair2<double>, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0));
}
template <> __device__ __forceinline__ Pair2<float> buffer_gather2<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff)
{
    // (bit_cast of the builtin's own vector type: converting it to an ext_vector_type first and reading .x / .y gave the
    //  first dword twice with this compiler)
    return __builtin_bit_cast(Pair2<float>, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, soff, 0));
}
template <typename T> __device__ __forceinline__ void buffer_put2(__amdgpu_buffer_rsrc_t rs, unsigned voff, T a, T b, bool nt);
template <> __device__ __forceinline__ void buffer_put2<double>(__amdgpu_buffer_rsrc_t rs, unsigned voff, double a, double b, bool nt)
{
    typedef unsigned mik_gv4 __attribute__((__vector_size__(16)));         // the builtin's own vector type
    const Pair2<double> v = {a, b};
    if (nt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mik_gv4, v), rs, (int)voff, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mik_gv4, v), rs, (int)voff, 0, 0);
}
template <> __device__ __forceinline__ void buffer_put2<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, float a, float b, bool nt)
{
    typedef unsigned mik_gv2 __attribute__((__vector_size__(8)));
    const Pair2<float> v = {a, b};
    if (nt) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(mik_gv2, v), rs, (int)voff, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(mik_gv2, v), rs, (int)voff, 0, 0);
}

template <typename T, bool FUSE_DOT, bool NT, int NS, int CQ>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_sdiab2(int n, int ncols, int koff, int pb0, int np, int nfull, int sshift, int nslices,
                                                           const SdiaSliceRec *__restrict__ recs, const SdiaPattern<T> *__restrict__ pats,
                                                           const unsigned char *__restrict__ mask, const T *__restrict__ x, T *__restrict__ y,
                                                           T *__restrict__ seg_out, const int *__restrict__ done,
                                                           const T *__restrict__ ep_w = nullptr, const T *__restrict__ ep_c = nullptr,
                                                           const T *__restrict__ ep_z = nullptr)
{
    static_assert(NS >= 3 && CQ >= 1 && CQ + 1 < NS, "the class must have slots around the centre");
    if (done && *done) return;
    constexpr int U = 8;
    constexpr unsigned ES = (unsigned)sizeof(T);
    __shared__ T lds[8];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((uintptr_t)x - (uintptr_t)koff * ES), (short)0, (int)0xFFFFFFF0u, (int)0x00020000);
    // x[r0], x[r0 + 1] through a descriptor over the ROWS (n is even: a pair lies inside or outside), the outer neighbours through
    // one over the COLUMNS (a row-partitioned block has more columns than rows: its last row's upper neighbour is a halo entry)
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc((void *)x, (short)0, (int)((unsigned)n * ES), (int)0x00020000);
    const __amdgpu_buffer_rsrc_t xw = __builtin_amdgcn_make_buffer_rsrc((void *)x, (short)0, (int)((unsigned)ncols * ES), (int)0x00020000);
    const __amdgpu_buffer_rsrc_t ms = __builtin_amdgcn_make_buffer_rsrc((void *)mask, (short)0, n, (int)0x00020000);
    const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc((void *)y, (short)0, (int)((unsigned)n * ES), (int)0x00020000);
    const int vb = min((int)blockIdx.x, np - 1);                          // past the end: the last pair once more (same bits)
    const int pb = pb0 + spmv_block_map_shift(vb, nfull, sshift);         // slices 2 pb and 2 pb + 1
    const int sl = min(2 * pb + (w >> 1), nslices - 1);                   // this wave's slice (a pair may lack its second one)
    const int r0 = pb * (2 * MIK_BLOCK) + w * 128 + 2 * lane;             // rows r0, r0 + 1 (n is even: both in range or neither)
    const unsigned rowoff = (unsigned)r0 * ES;
    // the two mask bytes (temporal: they stay in the Infinity Cache from one SpMV to the next); past the end: 0 = no slot
    const unsigned m16 = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(ms, r0, 0, 0);
    int minv0 = ~(int)(m16 & 0xffu), minv1 = ~(int)((m16 >> 8) & 0xffu);
    const SdiaSliceRec rc = recs[sl];
    const SdiaPattern<T> *__restrict__ pt = pats + rc.pid;
    asm volatile("" : "+v"(minv0), "+v"(minv1));
    // slots other than CQ - 1, CQ, CQ + 1: both rows of every pair must agree on their presence
    constexpr int NLMASK = ((1 << NS) - 1) & ~(7 << (CQ - 1));
    const bool mixed = __builtin_amdgcn_ballot_w64(((minv0 ^ minv1) & NLMASK) != 0) != 0;
    const bool fast = (rc.ns == NS) & (rc.cq == CQ) & (rc.soff[CQ - 1] + (int)ES == rc.soff[CQ]) & (rc.soff[CQ] + (int)ES == rc.soff[CQ + 1]) & !mixed;
    T acc0, acc1, xr0, xr1;
    if (fast) {
        const Pair2<T> xc = buffer_gather2<T>(xs, rowoff, 0);                // x[r0], x[r0 + 1]; a pair past the end reads +0
        T xe = T(0);
        const bool top = lane == 63 || r0 + 2 >= n;                        // the lane above is another wave's, or has no rows
        if (lane == 0 || top) {                                            // the outer neighbour of the wave's first / last row
            const unsigned ab = (unsigned)(top ? __builtin_amdgcn_sbfe(minv1, CQ + 1, 1) : __builtin_amdgcn_sbfe(minv0, CQ - 1, 1));
            xe = buffer_gather<T>(xw, (top ? rowoff + 2 * ES : rowoff - ES) | ab, 0);
        }
        T xf = T(0);
        if (lane == 0 && top)                                              // (a wave whose first pair is the last one needs both)
            xf = buffer_gather<T>(xw, (rowoff - ES) | (unsigned)__builtin_amdgcn_sbfe(minv0, CQ - 1, 1), 0);
        Pair2<T> xg[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q)
            if (q < CQ - 1 || q > CQ + 1)
                xg[q] = buffer_gather2<T>(rs, rowoff | (unsigned)__builtin_amdgcn_sbfe(minv0, q, 1), rc.soff[q]);
        T below = lane_next<true>(xc.b), above = lane_next<false>(xc.a);   // x[r0 - 1] from the lane below, x[r0 + 2] from the lane above
        if (top) above = xe;
        if (lane == 0) below = top ? xf : xe;
        // absent slots contribute value * +0, as through the gather
        xg[CQ - 1] = {((minv0 >> (CQ - 1)) & 1) ? T(0) : below, ((minv1 >> (CQ - 1)) & 1) ? T(0) : xc.a};
        xg[CQ] = {((minv0 >> CQ) & 1) ? T(0) : xc.a, ((minv1 >> CQ) & 1) ? T(0) : xc.b};
        xg[CQ + 1] = {((minv0 >> (CQ + 1)) & 1) ? T(0) : xc.b, ((minv1 >> (CQ + 1)) & 1) ? T(0) : above};
        T a0 = T(0), a1 = T(0);
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const T v = pt->val[q];
            T p0 = v * xg[q].a; a0 = a0 + p0;
            T p1 = v * xg[q].b; a1 = a1 + p1;
        }
        acc0 = a0; acc1 = a1; xr0 = xc.a; xr1 = xc.b;
    } else {
        const int ns = rc.ns, cq = rc.cq;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned ro = rowoff + (unsigned)e * ES;
            const int mi = e ? minv1 : minv0;
            T xv[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                xv[q] = T(0);
                if (q < ns) xv[q] = buffer_gather<T>(rs, ro | (unsigned)__builtin_amdgcn_sbfe(mi, q, 1), rc.soff[q]);
            }
            T a = T(0), c = T(0);
#pragma unroll
            for (int q = 0; q < U; ++q) {
                if (q < ns) {
                    if (q == cq) c = xv[q];
                    T pr = pt->val[q] * xv[q];
                    a = a + pr;
                }
            }
            if (FUSE_DOT && r0 + e < n && (cq < 0 || ((mi >> cq) & 1))) c = x[r0 + e];
            if (e) { acc1 = a; xr1 = c; } else { acc0 = a; xr0 = c; }
        }
    }
    if (FUSE_DOT && ep_w) {
        // the Lanczos step of MINRES as the epilogue (src/minres.jl:102-107): y = A x + c w (a rounded multiply, a rounded add, as
        // axpy! after mul! gives them), and the dot below is dot(x, y) of THAT y -- one sweep over v_prev, v_next, v_curr less
        const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc((void *)ep_w, (short)0, (int)((unsigned)n * ES), (int)0x00020000);
        const Pair2<T> wv = buffer_gather2<T>(wsr, rowoff, 0);             // a pair past the end reads +0
        const T c = *ep_c;
        const T t0 = c * wv.a, t1 = c * wv.b;
        acc0 = acc0 + t0; acc1 = acc1 + t1;
    }
    buffer_put2<T>(ys, rowoff, acc0, acc1, NT);
    if (FUSE_DOT && ep_z) {
        // dot(z, y) instead of dot(x, y): sigma = dot(r_shadow, A u) and rho = dot(r_shadow, A r) of BiCGStab(l) (src/bicgstabl.jl:100, :89)
        const __amdgpu_buffer_rsrc_t zsr = __builtin_amdgcn_make_buffer_rsrc((void *)ep_z, (short)0, (int)((unsigned)n * ES), (int)0x00020000);
        const Pair2<T> zv = buffer_gather2<T>(zsr, rowoff, 0);             // a pair past the end reads +0
        xr0 = zv.a; xr1 = zv.b;
    }
    if (FUSE_DOT) {
        T s0 = xr0 * acc0, s1 = xr1 * acc1;                                // a pair past the end: 0 * +0
        s0 = s0 + lane_down<16>(s0); s1 = s1 + lane_down<16>(s1);
        s0 = s0 + lane_down<8>(s0);  s1 = s1 + lane_down<8>(s1);
        s0 = s0 + lane_down<4>(s0);  s1 = s1 + lane_down<4>(s1);
        s0 = s0 + lane_down<2>(s0);  s1 = s1 + lane_down<2>(s1);
        s0 = s0 + lane_down<1>(s0);  s1 = s1 + lane_down<1>(s1);
        const T ws = s0 + s1;                                             // lanes 0 and 32: the sums of rows 0..63 and 64..127 of the wave
        if ((lane & 31) == 0) lds[2 * w + (lane >> 5)] = ws;
        __syncthreads();
        if (t < 2 && 2 * pb + t < nslices) {                              // one partial per slice: its 4 sums left to right
            T tot = lds[4 * t];
            tot = tot + lds[4 * t + 1]; tot = tot + lds[4 * t + 2]; tot = tot + lds[4 * t + 3];
            seg_out[2 * pb + t] = tot;
        }
    }
}

// ---- slice-constant values with up to 32 offsets per slice (layout 6) -----------------------------------------------------------
// The per-slice-offset forms above stop at 8 offsets per slice (one mask BYTE per row): 3 / 5 / 7-point stencils.  9-point 2-D,
// 13 / 19 / 27-point 3-D constant-coefficient stencils get the same treatment with one mask WORD per row: a slice is a pattern
// {ns <= 32 offsets in ascending order = the order in which a row's entries are summed, their scalar byte offsets, their values},
// equal patterns are stored once, a row keeps a 32-bit presence mask.  x goes through the buffer descriptor of k_spmv_sdiab: the
// slot's offset is the instruction's SCALAR offset, the row's byte offset the vector offset, and a slot the row does not have
// ORs all-ones into the vector offset -- out of range, so the hardware returns 0.0 and the row adds value * 0 = +-0, which leaves
// the sum bit for bit unchanged (the sum starts from +0).  4 B + x + y per row instead of 12 B per entry: a 27-point operator
// moves 20 B per row instead of 340.  One row per lane, 8 gathers in flight; the kernel is priced by its ns gather instructions
// per 64 rows (scripts/micro/gather_width.hip).  Built on the host (csr_build_sdiaw) when every value is finite and the offsets fit.
// item of k_spmv_sdiaw2: the run (o - 1, o, o + 1), o even, as slot BITS (0: the slice has no such slot) and values -- everything
// a wave needs for the item in one scalar load, no decoding
template <typename T> struct SdiawItem {
    int off;               // o * sizeof(T)
    unsigned ba, bb, bc;   // 1 << slot of o - 1 / o / o + 1, or 0
    T va, vb, vc;          // their values (+0 where the bit is 0)
};
struct SdiawItemHead { int off; unsigned ba, bb, bc; };   // the first 16 bytes of an item (4-byte aligned in the fp32 table)
template <typename T> struct SdiawPattern {
    int ns, nitems;        // nitems > 0 (a multiple of 3): the slots decompose into items (0: k_spmv_sdiaw2 runs the slice slot by slot)
    unsigned exa, exc;     // union of the items' ba / bc: the slots whose value reaches a wave's first / last row from outside the wave
    unsigned fullbits;     // all slots, if every item is a full run (o - 1, o, o + 1) -- else bit 31, which no row mask has
    int pad_[3];
    int soff[32];          // (offset + koff) * sizeof(T), ascending; 0 beyond ns
    T val[32];             // the slice's value for that offset; +0 beyond ns
    SdiawItem<T> items[24];
};

// mik_sdiaw_finish (mik_core.hip) writes the table byte by byte
static_assert(sizeof(SdiawItem<double>) == 40 && sizeof(SdiawItem<float>) == 28, "SdiawItem layout");
static_assert(sizeof(SdiawPattern<double>) == 160 + 256 + 24 * 40 && sizeof(SdiawPattern<float>) == 160 + 128 + 24 * 28, "SdiawPattern layout");

template <typename T, bool FUSE_DOT, bool NT>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_sdiaw(int n, int koff, int rb0, int nb, int map_mode, const int *__restrict__ pat_id,
                                                          const SdiawPattern<T> *__restrict__ pats, const unsigned *__restrict__ mask,
                                                          const T *__restrict__ x, T *__restrict__ y, T *__restrict__ seg_out,
                                                          const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr unsigned ES = (unsigned)sizeof(T);
    __shared__ T lds4[4];
    const int t = threadIdx.x;
    const int rb = rb0 + spmv_block_map((int)blockIdx.x, nb, map_mode);
    const int r = rb * MIK_BLOCK + t;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((uintptr_t)x - (uintptr_t)koff * ES), (short)0,
                                                                        (int)0xFFFFFFF0u, (int)0x00020000);
    const unsigned m = r < n ? mask[r] : 0u;                 // (cached: 4 B per row stay in the Infinity Cache from one SpMV to the next)
    const SdiawPattern<T> *__restrict__ p = pats + pat_id[rb];
    const int ns = p->ns;
    const unsigned rowoff = (unsigned)r * ES;
    T acc = T(0);
    for (int q0 = 0; q0 < ns; q0 += 8) {
        T xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u;                             // q >= ns: mask bit 0, value +0 -- adds +0
            const unsigned absent = ((m >> q) & 1u) ? 0u : 0xFFFFFFFFu;
            xv[u] = buffer_gather<T>(rs, rowoff | absent, p->soff[q]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const T pr = p->val[q0 + u] * xv[u]; acc = acc + pr; }
    }
    if (r < n) st_stream<NT>(y + r, acc);
    if (FUSE_DOT) {
        T pp = T(0);
        if (r < n) pp = x[r] * acc;
        const T tot = block_tree_256(pp, lds4);
        if (t == 0) seg_out[rb] = tot;
    }
}

// Per 128-row chunk (= the rows of one wave of k_spmv_sdiaw2) of the row masks: {U0, Z0, U1, Z1} = the slots ALL / ANY of its even
// rows have, and the same for its odd rows (rows >= n count as rows without slots).  The first row of the chunk is left out of U0
// for the slots that reach it from outside the wave (pattern.exa), the last row out of U1 for pattern.exc: those values come
// from the kernel's sparse edge load, which reads 0 when the row lacks the slot -- so the rows on the x faces of a grid do not
// make their whole wave take the per-lane path.  Built once at upload.
static __global__ __launch_bounds__(MIK_BLOCK) void k_sdiaw_chunk_bits(int n, int nchunks, int nslices, const unsigned *__restrict__ mask,
                                                                       const int *__restrict__ pat_id, const unsigned char *__restrict__ pats,
                                                                       int pat_bytes, uint4 *__restrict__ uz)
{
    const int c = (int)blockIdx.x * (MIK_BLOCK / 64) + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (c >= nchunks) return;
    const unsigned *ph = reinterpret_cast<const unsigned *>(pats + (size_t)pat_id[min(c / 2, nslices - 1)] * (size_t)pat_bytes);
    const unsigned exa = ph[2], exc = ph[3];
    const int64_t r0 = (int64_t)c * 128 + 2 * lane;
    const unsigned m0 = r0 < n ? mask[r0] : 0u, m1 = r0 + 1 < n ? mask[r0 + 1] : 0u;
    unsigned a0 = lane == 0 && r0 < n ? (m0 | exa) : m0, a1 = lane == 63 && r0 + 1 < n ? (m1 | exc) : m1, o0 = m0, o1 = m1;
    for (int d = 32; d >= 1; d >>= 1) {
        a0 &= (unsigned)__shfl_xor((int)a0, d); o0 |= (unsigned)__shfl_xor((int)o0, d);
        a1 &= (unsigned)__shfl_xor((int)a1, d); o1 |= (unsigned)__shfl_xor((int)o1, d);
    }
    if (lane == 0) uz[c] = make_uint4(a0, o0, a1, o1);
}

// One product of k_spmv_sdiaw2.  U / Z: the slots all / any of the wave's rows of this parity have (wave-uniform, from
// k_sdiaw_chunk_bits): a slot every row has costs a multiply and an add, a slot no row has nothing, and only a slot SOME rows
// have -- the rows on a face of the grid -- takes the per-lane test (absent: value * +0 = +-0, and acc + +-0 = acc since acc
// starts at +0 and never becomes -0).  The kernel is bound by vector-ALU issue, not by memory: the per-lane test and select
// tripled the instructions of a product.
template <typename T> __device__ __forceinline__ void sdiaw_term(T &acc, unsigned U, unsigned Z, unsigned m, unsigned bit, T val, T xv)
{
    if (U & bit) {
        const T pr = val * xv;
        acc = acc + pr;
    } else if (Z & bit) {
        const T pr = val * ((m & bit) ? xv : T(0));
        acc = acc + pr;
    }
}
// the neighbouring lane's value, the lane without a neighbour (0 / 63) keeps `old`
template <bool BELOW> __device__ __forceinline__ unsigned lane_next_old_u32(unsigned v, unsigned old)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, BELOW ? 0x138 : 0x130, 0xf, 0xf, false);
}
template <bool BELOW> __device__ __forceinline__ double lane_next_old(double v, double old)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v), o = __builtin_bit_cast(unsigned long long, old);
    const unsigned lo = lane_next_old_u32<BELOW>((unsigned)b, (unsigned)o), hi = lane_next_old_u32<BELOW>((unsigned)(b >> 32), (unsigned)(o >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <bool BELOW> __device__ __forceinline__ float lane_next_old(float v, float old)
{
    return __builtin_bit_cast(float, lane_next_old_u32<BELOW>(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, old)));
}

// Items of a wave whose rows all have every slot of a pattern made of full runs: B items' loads in flight, then 6 multiplies and
// 6 adds per item (a multiple of B items).  B = 9 -- a 27-point stencil's rows in ONE round trip -- costs registers (6 instead
// of 8 waves per SIMD) and still wins: the kernel waits for x, and what is in flight per CU is what counts.
template <typename T, int B>
__device__ __forceinline__ void sdiaw_full_runs(const SdiawPattern<T> *__restrict__ p, int nitems, __amdgpu_buffer_rsrc_t xw, unsigned rowoff,
                                                unsigned voe, unsigned pe, T &acc0, T &acc1)
{
    constexpr unsigned ES = (unsigned)sizeof(T);
    for (int i0 = 0; i0 < nitems; i0 += B) {
        Pair2<T> P[B];
        T E[B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const SdiawItemHead hd = *reinterpret_cast<const SdiawItemHead *>(&p->items[i0 + i]);   // {off, ba, bb, bc}: one scalar load
            P[i] = buffer_gather2<T>(xw, rowoff + (unsigned)hd.off, 0);                 // a negative column wraps beyond the descriptor's range: reads 0
            E[i] = buffer_gather<T>(xw, (pe & (hd.ba | hd.bc)) ? voe + (unsigned)hd.off : 0xFFFFFFFFu, 0);
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const T va = p->items[i0 + i].va, vb = p->items[i0 + i].vb, vc = p->items[i0 + i].vc;
            const T below = lane_next_old<true>(P[i].b, E[i]), above = lane_next_old<false>(P[i].a, E[i]);
            { const T pr = va * below; acc0 = acc0 + pr; }
            { const T pr = vb * P[i].a; acc0 = acc0 + pr; }
            { const T pr = vc * P[i].b; acc0 = acc0 + pr; }
            { const T pr = va * P[i].a; acc1 = acc1 + pr; }
            { const T pr = vb * P[i].b; acc1 = acc1 + pr; }
            { const T pr = vc * above; acc1 = acc1 + pr; }
        }
    }
    (void)ES;
}

// The same layout with TWO consecutive rows per lane (n even), for the reason k_spmv_sdiab2 exists: a 64-lane gather is priced per
// instruction, so a lane should take what one 16-byte load brings.  A slice's sorted offsets are decomposed at upload into ITEMS:
// a run (o - 1, o, o + 1) with o even -- the line neighbours of a stencil -- is served by ONE 16-byte gather of x[r + o], x[r + 1 + o]
// for the lane's rows r, r + 1: row r's three values are {the lane below's second value, a, b}, row r + 1's {a, b, the lane above's
// first value} (whole-wave DPP moves; lane 0 and lane 63 fetch their outer value with one sparse load and the DPP move leaves it
// in place), and a lone even offset by one 16-byte gather.  A 27-point row pair costs 9 + 9 vector-memory instructions per 128
// rows instead of 54.
// x is read through a descriptor over the columns: a column outside [0, ncols) reads 0.0 (so the pair of lanes past the last row
// delivers the right outer value to the last row), and a slot a row does not have is replaced by +0 before the multiply (sdiaw_term).
// A slice whose offsets do not decompose (an odd lone offset: odd grid sizes) is summed slot by slot by the same launch.
// dot(u, c) partials: the tree of the one-row kernels, as in k_spmv_sdiab2.
template <typename T, bool FUSE_DOT, bool NT>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_sdiaw2(int n, int ncols, int koff, int pb0, int np, int pmode, int nslices, const int *__restrict__ pat_id,
                                                           const SdiawPattern<T> *__restrict__ pats, const unsigned *__restrict__ mask,
                                                           const uint4 *__restrict__ chunk_bits, const T *__restrict__ x, T *__restrict__ y,
                                                           T *__restrict__ seg_out, const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr unsigned ES = (unsigned)sizeof(T);
    __shared__ T lds[8];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const __amdgpu_buffer_rsrc_t xw = __builtin_amdgcn_make_buffer_rsrc((void *)x, (short)0, (int)((unsigned)ncols * ES), (int)0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((uintptr_t)x - (uintptr_t)koff * ES), (short)0, (int)0xFFFFFFF0u, (int)0x00020000);
    const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc((void *)y, (short)0, (int)((unsigned)n * ES), (int)0x00020000);
    const int pb = pb0 + spmv_block_map(min((int)blockIdx.x, np - 1), np, pmode);   // slices 2 pb and 2 pb + 1; XCD strips as the other banded kernels
    const int sl = min(2 * pb + (w >> 1), nslices - 1);
    const int r0 = pb * (2 * MIK_BLOCK) + w * 128 + 2 * lane;             // rows r0, r0 + 1 (n is even: both in range or neither)
    unsigned m0 = 0u, m1 = 0u;
    if (r0 < n) { const uint2 mm = *reinterpret_cast<const uint2 *>(mask + r0); m0 = mm.x; m1 = mm.y; }
    const SdiawPattern<T> *__restrict__ p = pats + pat_id[sl];
    const int ns = p->ns, nitems = p->nitems;
    const unsigned rowoff = (unsigned)r0 * ES;
    Pair2<T> xc{T(0), T(0)};
    if (FUSE_DOT) xc = buffer_gather2<T>(xw, r0 < n ? rowoff : 0xFFFFFFFFu, 0);      // x[r0], x[r0 + 1] for the dot, in flight with the rest
    T acc0 = T(0), acc1 = T(0);
    if (nitems > 0) {
        // Every item has the same shape -- a lone even offset o is the run (o - 1, o, o + 1) whose outer slots nobody has (slot 31)
        // -- and the item list is padded to a multiple of B2 with items of three absent slots.
        const uint4 uz = chunk_bits[pb * 4 + w];
        const unsigned U0 = uz.x, Z0 = uz.y, U1 = uz.z, Z1 = uz.w;
        // the sparse load of the outer values: lane 0 fetches column r0 - 1 + o if its first row has the slot, lane 63 column
        // r0 + 2 + o if its second row has it -- everybody else, and a row without the slot, an out-of-range offset (reads 0; no
        // branch: exec-masked loads make the compiler drain the memory queue).  exa and exc are disjoint, so one test serves both.
        const unsigned pe = lane == 0 ? (m0 & p->exa) : (lane == 63 ? (m1 & p->exc) : 0u);
        const unsigned voe = rowoff + (lane ? 2 * ES : 0u - ES);
        constexpr int B2 = 3;
        const unsigned full = p->fullbits;
        if ((U0 & U1 & full) == full) {
            // every row of the wave has every slot and all items are full runs: 6 multiplies and 6 adds per item and row pair,
            // nothing to test (the kernel is bound by instruction issue, vector and scalar, not by memory)
            if (nitems % 9 == 0) sdiaw_full_runs<T, 9>(p, nitems, xw, rowoff, voe, pe, acc0, acc1);
            else sdiaw_full_runs<T, 3>(p, nitems, xw, rowoff, voe, pe, acc0, acc1);
        } else {
            for (int i0 = 0; i0 < nitems; i0 += B2) {
                Pair2<T> P[B2];
                T E[B2];
                SdiawItem<T> its[B2];
#pragma unroll
                for (int i = 0; i < B2; ++i) its[i] = p->items[i0 + i];
#pragma unroll
                for (int i = 0; i < B2; ++i) {
                    P[i] = buffer_gather2<T>(xw, rowoff + (unsigned)its[i].off, 0);
                    E[i] = buffer_gather<T>(xw, (pe & (its[i].ba | its[i].bc)) ? voe + (unsigned)its[i].off : 0xFFFFFFFFu, 0);
                }
#pragma unroll
                for (int i = 0; i < B2; ++i) {
                    const SdiawItem<T> &it = its[i];
                    if (Z0 & it.ba) sdiaw_term<T>(acc0, U0, Z0, m0, it.ba, it.va, lane_next_old<true>(P[i].b, E[i]));
                    sdiaw_term<T>(acc0, U0, Z0, m0, it.bb, it.vb, P[i].a);
                    sdiaw_term<T>(acc0, U0, Z0, m0, it.bc, it.vc, P[i].b);
                    sdiaw_term<T>(acc1, U1, Z1, m1, it.ba, it.va, P[i].a);
                    sdiaw_term<T>(acc1, U1, Z1, m1, it.bb, it.vb, P[i].b);
                    if (Z1 & it.bc) sdiaw_term<T>(acc1, U1, Z1, m1, it.bc, it.vc, lane_next_old<false>(P[i].a, E[i]));
                }
            }
        }
    } else {
        // slot by slot, each row on its own (k_spmv_sdiaw's body twice)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned ro = rowoff + (unsigned)e * ES, m = e ? m1 : m0;
            T a = T(0);
            for (int q0 = 0; q0 < ns; q0 += 8) {
                T xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = q0 + u;
                    const unsigned absent = ((m >> q) & 1u) ? 0u : 0xFFFFFFFFu;
                    xv[u] = buffer_gather<T>(rs, ro | absent, p->soff[q]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const T pr = p->val[q0 + u] * xv[u]; a = a + pr; }
            }
            if (e) acc1 = a; else acc0 = a;
        }
    }
    buffer_put2<T>(ys, rowoff, acc0, acc1, NT);
    if (FUSE_DOT) {
        T s0 = xc.a * acc0, s1 = xc.b * acc1;                              // a pair past the end: 0 * +0
        s0 = s0 + lane_down<16>(s0); s1 = s1 + lane_down<16>(s1);
        s0 = s0 + lane_down<8>(s0);  s1 = s1 + lane_down<8>(s1);
        s0 = s0 + lane_down<4>(s0);  s1 = s1 + lane_down<4>(s1);
        s0 = s0 + lane_down<2>(s0);  s1 = s1 + lane_down<2>(s1);
        s0 = s0 + lane_down<1>(s0);  s1 = s1 + lane_down<1>(s1);
        const T ws = s0 + s1;                                             // lanes 0 and 32: the sums of rows 0..63 and 64..127 of the wave
        if ((lane & 31) == 0) lds[2 * w + (lane >> 5)] = ws;
        __syncthreads();
        if (t < 2 && 2 * pb + t < nslices) {                              // one partial per slice: its 4 sums left to right
            T tot = lds[4 * t];
            tot = tot + lds[4 * t + 1]; tot = tot + lds[4 * t + 2]; tot = tot + lds[4 * t + 3];
            seg_out[2 * pb + t] = tot;
        }
    }
}

#endif  // __HIPCC__
#endif  // MIK_SELL_H
