// mik_spmv.h -- CSR SpMV for gfx950: row-block workgroups, LDS-staged products, serial row sums.
//
// y = A*x replaces SparseArrays' mul!(y, A::SparseMatrixCSC, x) at src/cg.jl:54,137 and
// src/gmres.jl:245,287 (the reference's bottleneck: a serial column scatter over Int64 indices).
//
// Layout.  A workgroup of 256 threads owns one row-block of 256 consecutive rows (one row per
// thread).  The nonzeros of a row-block are one contiguous range of the CSR arrays; the workgroup
// streams it in tiles of MIK_SPMV_TILE entries with fully coalesced loads of val[] and col[],
// gathers x[col] (L1/L2/Infinity-Cache hits for stencil matrices) and parks the products in LDS.
// Each thread then adds up its own row's products from LDS in ascending column order -- the order
// in which the reference's CSC column scatter reaches that row -- so y is bit-identical to the
// oracle.  For the 7-point stencil the per-thread LDS stride is 7 doubles = 14 banks:
// conflict-free for ds_read_b64.
//
// What was measured on MI355X (256^3 Laplacian, profiles/ and DESIGN.md):
//  * one row-block per workgroup and ~65k workgroups beats persistent / software-pipelined
//    variants (the hardware's 8 resident workgroups per CU already overlap the load, gather and
//    row-sum phases of different row-blocks; a register-prefetching loop cut occupancy to 6 and
//    lost 15-25 %);
//  * the val/col/y streams are touched exactly once: non-temporal loads/stores (NT) keep them
//    from evicting x out of the 4 MiB L2s and cut fabric reads (-8 % time);
//  * WIDE: 16-byte loads of val and 8/16-byte loads of col (aligned tile start) halve the number
//    of vector-memory instructions per tile.
//
// FUSE_DOT adds CG's dot(u, c) (src/cg.jl:55) as an epilogue: p = x[row] * y[row] per thread,
// block tree, one segment sum per row-block: the (W, L) = (1, 1) reduction shape of include/mik.h.
#pragma once
#include "mik_internal.h"

#ifdef __HIPCC__

// streamed-once data (val, col, y) can bypass cache retention so that x keeps its place in L2
template <bool NT, typename U> __device__ __forceinline__ U ld_stream(const U *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT, typename U> __device__ __forceinline__ void st_stream(U *p, U v)
{
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// native clang vectors (the non-temporal builtins reject HIP's struct-based double2 / int4)
typedef double mik_f64x2 __attribute__((ext_vector_type(2)));
typedef float mik_f32x4 __attribute__((ext_vector_type(4)));
typedef int mik_i32x2 __attribute__((ext_vector_type(2)));
typedef int mik_i32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct WideVec;
template <> struct WideVec<double> { using val = mik_f64x2; using idx = mik_i32x2; };
template <> struct WideVec<float>  { using val = mik_f32x4; using idx = mik_i32x4; };

// block -> row-block map.  mode 0: identity (block b runs on XCD b % 8, so the 8 XCDs interleave
// row-blocks; rows +-N^2 of a stencil then live in the SAME XCD whenever the plane size is a
// multiple of 8 row-blocks).  mode 1: contiguous range per XCD.
// mode P >= 8 (a multiple of 8): "strips" -- the row-blocks are taken in planes of P (P = row-blocks
// between a stencil row and its farthest neighbour row); inside every plane XCD k owns the contiguous
// strip [k*P/8, (k+1)*P/8), and walks strip k of plane 0, then of plane 1, ...  Both the +-1 line
// neighbours (same strip) and the +-plane neighbours (same XCD, P/8 workgroups earlier) then hit in
// that XCD's L2, so x is fetched from the fabric about once instead of three times.
__device__ __forceinline__ int spmv_block_map(int b, int nb, int mode)
{
    if (mode == 1) return xcd_remap(b, nb);
    if (mode >= 8) {
        const int P = mode, S = P >> 3;
        if (b < (nb / P) * P) {
            const int xcd = b & 7, q = b >> 3;
            return (q / S) * P + xcd * S + (q % S);
        }
    }
    return b;
}

// ---------------------------------------------------------------------------------------------
// long rows: one wave per row
// ---------------------------------------------------------------------------------------------
// Rows with more than MIK_LONG_ROW entries are stored behind the short part (mik_csr_create) and summed
// here -- a thread-per-row tile only pays while a 2048-entry tile covers many rows, and a single serial
// chain over a 20,000-entry row costs > 100 us however it is fed (measured: ~13 cycles per entry).
//
// Row-sum shape for these rows (part of the documented reduction semantics, include/mik.h; the oracle's
// TREE mode mirrors it): lane l of the wave sums the products of entries l, l+64, l+128, ... of the row
// in ascending order starting from +0, then the wave-64 shuffle-down tree (offsets 32..1) -- the same
// shape as one 64-thread segment of a dot product.  Rows up to MIK_LONG_ROW entries keep the reference's
// strictly sequential order.
//
// The wave streams the row in chunks of 64*U entries (coalesced val/col loads, gathered x) through a
// three-stage software pipeline: while chunk c is accumulated, the x-gather of chunk c+1 and the
// val/col stream of chunk c+2 are in flight.
constexpr int MIK_LONG_U = 8;                          // entries per lane per chunk
constexpr int MIK_LONG_CH = 64 * MIK_LONG_U;           // 512-entry chunks

// Rows with more than MIK_LONG_SEG entries are cut into SEGMENTS of MIK_LONG_SEG consecutive entries: every segment is
// summed by its own wave with the shape above (lane l: the segment's entries l, l + 64, ... in order; wave tree), and
// the segment sums are added left to right -- by whichever wave finishes the row's last outstanding segment (an integer
// ticket elects it; the sums themselves are stored individually and always added in segment order, so the result does
// not depend on the order in which the waves finish).  A guard against pathological rows: one wave per row makes a dense
// row of 10^6 entries a single serial chain of ~2000 chunks (milliseconds).  On the irregular configs[4] stand-in (rows up
// to 20 k entries) cutting changes nothing -- 196 us uncut, 198 us cut: that SpMV is bound by its 34 M random gathers of x
// (profiles/r02_c5_*), not by the length of any chain.
// The oracle's long-row mode mirrors the segments (orc.set_long_row(threshold, segment)).
constexpr int MIK_LONG_SEG = 2048;

// Tables of the long-row part (device, built at upload).  `rows[w]` >= 0: virtual row w is a whole row, its sum goes to
// y[rows[w]]; < 0: it is segment -(rows[w] + 1) of a cut row.
struct LongTab {
    const int *rows, *starts, *lens;   // per virtual row (whole rows and segments, longest first)
    const int *seg_row;                // per segment: index h of its cut row
    const int *cut_row, *cut_first, *cut_nseg;   // per cut row: matrix row, first segment, number of segments
    unsigned *tickets;                 // per cut row, zero between launches
    void *seg_sum;                     // per segment, dtype of the operator
    int nlong, nbig;
};

template <typename T> __device__ __forceinline__ void longrow_store(const LongTab &lt, int w, T acc, T *__restrict__ y)
{
    const int tgt = lt.rows[w];
    if (tgt >= 0) { y[tgt] = acc; return; }
    const int sg = -tgt - 1;
    T *ss = (T *)lt.seg_sum;
    // Hand-off without cache-wide fences (an acq_rel ticket costs a buffer_wbl2 + buffer_inv per segment: measured 402 us
    // instead of 196 us for the whole SpMV): the segment sum is ONE write-through (sc1) store, drained before the ticket
    // is taken; the wave that takes the last ticket reads the sums back with sc1 loads, which are served past its L1.
    __hip_atomic_store(&ss[sg], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int h = lt.seg_row[sg];
    const unsigned tk = __hip_atomic_fetch_add(&lt.tickets[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ns = lt.cut_nseg[h];
    if (tk == (unsigned)ns - 1u) {
        const int f = lt.cut_first[h];
        T t = __hip_atomic_load(&ss[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q = 1; q < ns; ++q) t = t + __hip_atomic_load(&ss[f + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&lt.tickets[h], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        y[lt.cut_row[h]] = t;
    }
}

template <typename T>
__device__ __forceinline__ void spmv_longrow_wave(int w, const LongTab &lt, const int *__restrict__ col,
                                                  const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y)
{
    constexpr int U = MIK_LONG_U, CH = MIK_LONG_CH;
    const int lane = threadIdx.x & 63;
    const int k0 = lt.starts[w], len = lt.lens[w];
    T acc = T(0);
    if (len <= CH) {
        // medium rows: one chunk, every load issued at once (no pipeline prologue / epilogue)
        T v[U], xv[U];
        int c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = u * 64 + lane;
            if (j < len) { v[u] = val[k0 + j]; c[u] = col[k0 + j]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u * 64 + lane < len) xv[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u * 64 + lane < len) { const T prod = v[u] * xv[u]; acc = acc + prod; }
        acc = wave_tree(acc);
        if (lane == 0) longrow_store<T>(lt, w, acc, y);
        return;
    }
    T vA[U], xA[U], vB[U], vC[U];
    int cB[U], cC[U];
    auto stream = [&](int base, T(&vv)[U], int(&cc)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base + u * 64 + lane;
            vv[u] = j < len ? val[k0 + j] : T(0);
            cc[u] = j < len ? col[k0 + j] : 0;       // padding gathers x[0]; its product is never added
        }
    };
    stream(0, vA, cB);                                  // chunk 0 (cB doubles as its column registers)
#pragma unroll
    for (int u = 0; u < U; ++u) xA[u] = x[cB[u]];
    stream(CH, vB, cB);                                 // chunk 1
    for (int base = 0; base < len; base += CH) {
        stream(base + 2 * CH, vC, cC);                  // chunk c+2: stream
        T xB[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xB[u] = x[cB[u]];   // chunk c+1: gather
#pragma unroll
        for (int u = 0; u < U; ++u) {                   // chunk c: this lane's entries, ascending
            const T prod = vA[u] * xA[u];
            if (base + u * 64 + lane < len) acc = acc + prod;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { vA[u] = vB[u]; xA[u] = xB[u]; vB[u] = vC[u]; cB[u] = cC[u]; }
    }
    acc = wave_tree(acc);
    if (lane == 0) longrow_store<T>(lt, w, acc, y);
}

// A wave takes MIK_LONG_R consecutive rows of the (longest-first) long-row list.  Medium rows (<= 256
// entries) are latency-bound one at a time -- ~1 KB in flight per wave -- so their loads are issued for
// all R rows before anything is waited for; longer rows go through the pipelined path one by one.  The
// per-row arithmetic (lane l: entries l, l+64, ... in order; wave tree) is identical either way.
constexpr int MIK_LONG_R = 4;

template <typename T>
__device__ __forceinline__ void spmv_longrow_group(int wv, const LongTab &lt, const int *__restrict__ col, const T *__restrict__ val,
                                                   const T *__restrict__ x, T *__restrict__ y)
{
    constexpr int R = MIK_LONG_R, U = 4;
    const int nlong = lt.nlong, nbig = lt.nbig;
    const int *__restrict__ starts = lt.starts, *__restrict__ lens = lt.lens;
    // waves [0, nbig): one virtual row each (more than 64*U entries, longest first: their pipelined sums are
    // the critical path); waves from nbig on: R medium rows each (never segments: those are longer)
    if (wv < nbig) { spmv_longrow_wave<T>(wv, lt, col, val, x, y); return; }
    const int w0 = nbig + (wv - nbig) * R;
    if (w0 >= nlong) return;
    const int lane = threadIdx.x & 63;
    int k0[R], len[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const bool ok = w0 + q < nlong;
        k0[q] = ok ? starts[w0 + q] : 0;
        len[q] = ok ? lens[w0 + q] : 0;
    }
    T v[R][U], xv[R][U];
    int c[R][U];
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = u * 64 + lane;
            if (j < len[q]) { v[q][u] = val[k0[q] + j]; c[q][u] = col[k0[q] + j]; }
        }
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u * 64 + lane < len[q]) xv[q][u] = x[c[q][u]];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        T acc = T(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u * 64 + lane < len[q]) { const T prod = v[q][u] * xv[q][u]; acc = acc + prod; }
        acc = wave_tree(acc);
        if (lane == 0 && w0 + q < nlong) longrow_store<T>(lt, w0 + q, acc, y);     // a cut row's short last segment lands here too
    }
}

template <typename T>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_longrows(LongTab lt, const int *__restrict__ col, const T *__restrict__ val,
                                                             const T *__restrict__ x, T *__restrict__ y, const int *__restrict__ done)
{
    if (done && *done) return;
    const int wv = blockIdx.x * (MIK_BLOCK / 64) + (threadIdx.x >> 6);     // whole waves work alone: no block-level barrier
    spmv_longrow_group<T>(wv, lt, col, val, x, y);
}

// MERGE_LONG: the first `nlong_blocks` workgroups of the launch are long-row workgroups (4 rows each,
// scheduled first so the longest chains start earliest); the rest are row-block workgroups.
template <typename T, bool FUSE_DOT, bool NT, bool WIDE, bool MERGE_LONG>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_rowblock(int n, int nb, int map_mode, const int *__restrict__ rowptr,
                                                             const int *__restrict__ col, const T *__restrict__ val,
                                                             const T *__restrict__ x, T *__restrict__ y,
                                                             T *__restrict__ seg_out, const int *__restrict__ done,
                                                             const unsigned char *__restrict__ is_long, int nlb, LongTab lt)
{
    if (done && *done) return;
    constexpr int TILE = MIK_SPMV_TILE * (int)(8 / sizeof(T));   // 16 KB of LDS: 2048 fp64 / 4096 fp32 products
    constexpr int VW = WIDE ? VT<T>::W : 1;            // elements per lane per load
    constexpr int PER = TILE / (MIK_BLOCK * VW);       // loads per lane per tile
    __shared__ __attribute__((aligned(16))) T prod[TILE];
    __shared__ T lds4[4];

    const int t = threadIdx.x;
    int bid = blockIdx.x;
    if (MERGE_LONG) {
        if (bid < nlb) {
            spmv_longrow_group<T>(bid * (MIK_BLOCK / 64) + (t >> 6), lt, col, val, x, y);
            return;
        }
        bid -= nlb;
    }
    const int rb = spmv_block_map(bid, nb, map_mode);
    const int r0 = rb * MIK_BLOCK;
    const int r = r0 + t;
    int ks = 0, ke = 0;
    if (r < n) { ks = rowptr[r]; ke = rowptr[r + 1]; }
    const int kb = rowptr[r0] & ~(VW - 1);             // tile start aligned for the wide loads
    const int kend = rowptr[min(r0 + MIK_BLOCK, n)];

    T acc = T(0);
    for (int kc = kb; kc < kend; kc += TILE) {
        const int cnt = min(TILE, kend - kc);
        // ---- stage: coalesced stream of val/col, gather of x, products into LDS ----
        if (WIDE) {
            using VV = typename WideVec<T>::val;
            using IV = typename WideVec<T>::idx;
            VV v[PER];
            IV c[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = VW * (t + MIK_BLOCK * i);
                if (j < cnt) {      // reads past kend stay inside the padded allocation
                    v[i] = ld_stream<NT>(reinterpret_cast<const VV *>(val + kc + j));
                    c[i] = ld_stream<NT>(reinterpret_cast<const IV *>(col + kc + j));
                }
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = VW * (t + MIK_BLOCK * i);
                if (j < cnt) {
                    T xv[VW];
#pragma unroll
                    for (int e = 0; e < VW; ++e) xv[e] = x[c[i][e]];   // padding cols are 0
#pragma unroll
                    for (int e = 0; e < VW; ++e) prod[j + e] = v[i][e] * xv[e];
                }
            }
        } else {
            T v[PER];
            int c[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = t + MIK_BLOCK * i;
                if (j < cnt) { v[i] = ld_stream<NT>(val + kc + j); c[i] = ld_stream<NT>(col + kc + j); }
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = t + MIK_BLOCK * i;
                if (j < cnt) prod[j] = v[i] * x[c[i]];
            }
        }
        __syncthreads();
        // ---- per-row serial sum, ascending column order ----
        int a = max(ks, kc) - kc;
        int len = min(ke, kc + cnt) - kc - a;
        while (len > 0) {
            T q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = prod[min(a + i, TILE - 1)];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < len) acc = acc + q[i];
            a += 8;
            len -= 8;
        }
        __syncthreads();
    }
    if (is_long && r < n && is_long[r]) acc = y[r];    // summed by k_spmv_longrows earlier on the stream
    else if (r < n) st_stream<NT>(y + r, acc);
    if (FUSE_DOT) {
        T p = T(0);
        if (r < n) p = x[r] * acc;
        T tot = block_tree_256(p, lds4);
        if (t == 0) seg_out[rb] = tot;
    }
}

// ---------------------------------------------------------------------------------------------
// the default CSR kernel: row-block tile filled by LDS-DMA, per-row gather
// ---------------------------------------------------------------------------------------------
// k_spmv_rowblock parks PRODUCTS in LDS: the lanes that stream consecutive ENTRIES also gather x for them, so one
// gather instruction of a wave touches ~10 cache lines for a 7-point row-block (entries of ~9 rows, 7 regions of x).
// Here the tile holds the operator's val[] / col[] AS THEY ARE; after the barrier every lane walks ITS row: column and
// value from LDS, x[column] from memory -- lanes l, l + 1 then read neighbouring x entries for banded operators
// (1-2 lines per wave-instruction) -- multiply, add, in ascending column order from +0: same products, same order,
// same bits.
// The tile is filled by global_load_lds_dwordx4 (gfx950 LDS-DMA: HBM -> LDS without passing through VGPRs; 1 KiB per
// wave-instruction, the LDS image is the memory image), so staging costs no vector registers, no ds_write and no
// VALU work; the stream is waited for with one vmcnt(0) before the barrier.
//
// Measured at 256^3 fp64 inside the CG loop (scripts/spmv_inloop.py, interleaved rounds): k_spmv_rowblock 302 us,
// this layout staged through registers 299 us, filled by LDS-DMA 291 us (278 us back to back = 6.25 TB/s).  Also
// built and measured: wave-private tiles (every wave stages its own 64 rows, no workgroup barrier at all) -- 354-380 us
// in all four variants (row / entry gather, padded or not): the barriers were not what the workgroup tile was
// waiting for, and per-element LDS writes cost more than they saved.
template <typename T, bool FUSE_DOT, bool NT>
__global__ __launch_bounds__(MIK_BLOCK, 6) void k_spmv_rowgather(int n, int rb0, int nb, int map_mode, const int *__restrict__ rowptr,
                                                              const int *__restrict__ col, const T *__restrict__ val,
                                                              const T *__restrict__ x, T *__restrict__ y, T *__restrict__ seg_out,
                                                              const int *__restrict__ done, const unsigned char *__restrict__ is_long)
{
    if (done && *done) return;
    constexpr int TILE = MIK_SPMV_TILE;                // entries per pass: 2048 (fp64: 16 KB values + 8 KB columns)
    constexpr int VW = VT<T>::W;
    constexpr int VP = 1024 / (int)sizeof(T);          // entries per 1-KiB DMA piece of val
    constexpr int CP = 256;                            // entries per 1-KiB DMA piece of col
    __shared__ __attribute__((aligned(16))) T sval[TILE];
    __shared__ __attribute__((aligned(16))) int scol[TILE];
    __shared__ T lds4[4];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int rb = rb0 + spmv_block_map((int)blockIdx.x, nb, map_mode);   // this launch covers row-blocks [rb0, rb0 + nb)
    const int r0 = rb * MIK_BLOCK;
    const int r = r0 + t;
    int ks = 0, ke = 0;
    if (r < n) { ks = rowptr[r]; ke = rowptr[r + 1]; }
    const int kb = rowptr[r0] & ~3;                    // 16-byte aligned start of both streams
    const int kend = rowptr[min(r0 + MIK_BLOCK, n)];

    T acc = T(0);
    for (int kc = kb; kc < kend; kc += TILE) {
        const int cnt = min(TILE, kend - kc);
        // wave wv issues pieces wv, wv + 4, ...; a piece is issued iff it holds an entry < cnt (wave-uniform);
        // reads past kend stay inside the padded allocation
#pragma unroll
        for (int p = 0; p < TILE / VP / 4; ++p) {
            const int piece = wv + 4 * p;
            if (piece * VP < cnt)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(val + kc + piece * VP + lane * VW),
                                                 (__attribute__((address_space(3))) void *)(sval + piece * VP), 16, 0, NT ? 2 : 0);
        }
#pragma unroll
        for (int p = 0; p < TILE / CP / 4; ++p) {
            const int piece = wv + 4 * p;
            if (piece * CP < cnt)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(col + kc + piece * CP + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(scol + piece * CP), 16, 0, NT ? 2 : 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- this lane's row, ascending column order ----
        int a = max(ks, kc) - kc;
        int len = min(ke, kc + cnt) - kc - a;
        while (len > 0) {
            T q[8];
            int cc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int s = min(a + i, TILE - 1);
                cc[i] = scol[s];
                q[i] = sval[s];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const T xv = x[i < len ? cc[i] : 0];              // slots past the row gather a valid address
                q[i] = q[i] * xv;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < len) acc = acc + q[i];
            a += 8;
            len -= 8;
        }
        if (kc + TILE < kend) __syncthreads();         // workgroup-uniform: another pass will overwrite the tile
    }
    if (is_long && r < n && is_long[r]) acc = y[r];    // summed by k_spmv_longrows earlier on the stream
    else if (r < n) st_stream<NT>(y + r, acc);
    if (FUSE_DOT) {
        T p = T(0);
        if (r < n) p = x[r] * acc;
        T tot = block_tree_256(p, lds4);
        if (t == 0) seg_out[rb] = tot;
    }
}

#endif  // __HIPCC__
