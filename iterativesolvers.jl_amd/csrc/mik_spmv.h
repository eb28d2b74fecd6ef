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
// long rows: one wave per segment
// ---------------------------------------------------------------------------------------------
// Rows with more than MIK_LONG_ROW entries are stored behind the short part (mik_csr_create) and summed
// here -- a thread-per-row tile only pays while a 2048-entry tile covers many rows, and a single serial
// chain over a 20,000-entry row costs > 100 us however it is fed (measured: ~13 cycles per entry).
//
// Row-sum shape for these rows (part of the documented reduction semantics, include/mik.h; the oracle's
// long-row mode mirrors it): a row is cut into SEGMENTS of MIK_LONG_SEG consecutive entries; inside a segment
// lane l of a wave sums, starting from +0 and in ascending order, the products of the GROUPS l, l + 64, l + 128, ...
// of MIK_LONG_G = 4 consecutive entries (entry e of the segment belongs to lane (e / 4) % 64) -- the shape of one
// 64-thread segment of a dot product with 16-byte loads (include/mik.h, level 1) -- then the wave-64 shuffle-down
// tree (offsets 32..1); the segment sums are added left to right.  Rows up to MIK_LONG_ROW entries keep the
// reference's strictly sequential order.
//
// Round 4: until round 3 a lane took single entries l, l + 64, ... with 4-byte loads through a
// three-stage chunk pipeline, 4 unrelated rows per workgroup, longest first: 40.7 us for the 12.5 M long-row entries
// of the banded configs[4] stand-in (2.4 TB/s; profiles/r04_c5_banded_*).  A vector-memory instruction is priced
// per instruction on this GPU (scripts/micro/gather_width.hip), so the operator streams now come as ONE 16-byte
// column load and one (fp64: two) 16-byte value load(s) per lane and group -- which is what fixes the group of 4
// in the shape above -- every row starts 16-byte aligned in the long part, a segment is ONE pass whose loads are all
// issued before anything is waited for (few registers: eight waves per SIMD), and the virtual rows are listed in
// (row, segment) order so that the four waves of a workgroup walk neighbouring pieces of ONE row's sorted column
// window (val / col streamed non-temporally, x gathered with the default policy).
constexpr int MIK_LONG_G = 4;                          // entries per lane and group (one 16-byte column load)

// Rows with more than MIK_LONG_SEG entries are cut into segments: every segment is summed by its own wave with the
// shape above and the segment sums are added left to right -- by whichever wave finishes the row's last outstanding
// segment (an integer ticket elects it; the sums themselves are stored individually and always added in segment order,
// so the result does not depend on the order in which the waves finish).
// The oracle's long-row mode mirrors the segments and the groups (orc.set_long_row(threshold, segment, group)).
constexpr int MIK_LONG_SEG = 1024;

// Tables of the long-row part (device, built at upload).  `rows[w]` >= 0: virtual row w is a whole row, its sum goes to
// y[rows[w]]; < 0: it is segment -(rows[w] + 1) of a cut row.
struct LongTab {
    const int *rows, *starts, *lens;   // per virtual row (whole rows and segments, in (row, segment) order); starts are multiples of 4
    const int *seg_row;                // per segment: index h of its cut row
    const int *cut_row, *cut_first, *cut_nseg;   // per cut row: matrix row, first segment, number of segments
    unsigned *tickets;                 // per cut row, zero between launches
    void *seg_sum;                     // per segment, dtype of the operator
    int nlong;
};

// the sum of virtual row w (in lane 0) to its place; called by the WHOLE wave
template <typename T> __device__ __forceinline__ void longrow_store(const LongTab &lt, int w, T acc, T *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int tgt = lt.rows[w];                         // wave-uniform
    if (tgt >= 0) { if (lane == 0) y[tgt] = acc; return; }
    const int sg = -tgt - 1;
    T *ss = (T *)lt.seg_sum;
    const int h = lt.seg_row[sg];
    const int ns = lt.cut_nseg[h];
    // Hand-off without cache-wide fences (an acq_rel ticket costs a buffer_wbl2 + buffer_inv per segment: measured 402 us
    // instead of 196 us for the whole SpMV): the segment sum is ONE write-through (sc1) store, drained before the ticket
    // is taken; the wave that takes the last ticket reads the sums back with sc1 loads, which are served past its L1 --
    // 64 of them at a time, one per lane (a 20,000-entry row has 20 segments: one round trip instead of 20 dependent ones),
    // and adds them left to right.
    int last = 0;
    if (lane == 0) {
        __hip_atomic_store(&ss[sg], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tk = __hip_atomic_fetch_add(&lt.tickets[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = tk == (unsigned)ns - 1u ? 1 : 0;
    }
    last = __shfl(last, 0);
    if (!last) return;
    const int f = lt.cut_first[h];
    T t = T(0);
    for (int q0 = 0; q0 < ns; q0 += 64) {
        const int m = min(64, ns - q0);
        T v = T(0);
        if (lane < m) v = __hip_atomic_load(&ss[f + q0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q = 0; q < m; ++q) { const T vq = __shfl(v, q); t = (q0 + q == 0) ? vq : t + vq; }
    }
    if (lane == 0) {
        __hip_atomic_store(&lt.tickets[h], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        y[lt.cut_row[h]] = t;
    }
}

// the 4 values of a group: one 16-byte load (fp32) or two (fp64)
template <typename T> struct LongVal;
template <> struct LongVal<float> {
    mik_f32x4 a;
    __device__ __forceinline__ void load(const float *p) { a = __builtin_nontemporal_load(reinterpret_cast<const mik_f32x4 *>(p)); }
    __device__ __forceinline__ float get(int e) const { return a[e]; }
};
template <> struct LongVal<double> {
    mik_f64x2 a, b;
    __device__ __forceinline__ void load(const double *p)
    {
        a = __builtin_nontemporal_load(reinterpret_cast<const mik_f64x2 *>(p));
        b = __builtin_nontemporal_load(reinterpret_cast<const mik_f64x2 *>(p) + 1);
    }
    __device__ __forceinline__ double get(int e) const { return e < 2 ? a[e] : b[e - 2]; }
};

// wave w of the long part: virtual row w (a whole row of at most one segment, or one segment of a cut row)
//
// xw / wlo / wlen (round 4): the workgroup's WINDOW of x in LDS.  The list of virtual rows is sorted by first column (mik_csr_create),
// so the four waves of a workgroup gather from nearly the same stretch of x: the workgroup copies x[wlo .. wlo + wlen) into LDS once
// (LDS-DMA, spmv_long_window) and a gather inside it is a ds_read -- a line of x then crosses L2 -> L1 once per workgroup instead of
// once per wave and 128 bytes at a time for 16 useful ones; a column outside the window is gathered from memory as before.
template <typename T>
__device__ __forceinline__ void spmv_longrow_wave(int w, const LongTab &lt, const int *__restrict__ col, const T *__restrict__ val,
                                                  const T *__restrict__ x, T *__restrict__ y, const T *xw = nullptr, int wlo = 0, int wlen = 0)
{
    if (w >= lt.nlong) return;                          // wave-uniform
    constexpr int G = MIK_LONG_G, U = sizeof(T) == 8 ? 2 : 4;   // groups per lane and pass (fp64 carries twice the value registers)
    const int lane = threadIdx.x & 63;
    const int k0 = lt.starts[w], len = lt.lens[w];      // k0 is a multiple of 4: 16-byte aligned streams
    const int ng = (len + G - 1) / G;                   // groups of this virtual row (the last one padded in storage)
    const int npass = (ng + 64 * U - 1) / (64 * U);
    // One pass = 64 U groups = 1024 entries at fp32 (512 at fp64): a whole default segment.  All of a pass's streams are issued, then all
    // of its gathers; nothing is carried from pass to pass but the accumulator -- 48 vector registers, eight waves per SIMD (the double-
    // buffered two-pass form of a 2048-entry segment needed ~100 registers: four waves per SIMD, 1,775 workgroups in two rounds, 42 us).
    T acc = T(0);
    for (int p = 0; p < npass; ++p) {
        mik_i32x4 cc[U];
        LongVal<T> vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = lane + 64 * (U * p + u);
            const int gg = g < ng ? g : 0;              // a lane without a group re-reads the first one; its products are never added
            cc[u] = __builtin_nontemporal_load(reinterpret_cast<const mik_i32x4 *>(col + k0 + G * gg));
            vv[u].load(val + k0 + G * gg);
        }
        T xv[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const int c = cc[u][e];
                const unsigned o = (unsigned)(c - wlo);
                xv[u][e] = o < (unsigned)wlen ? xw[o] : x[c];       // wlen = 0: no window
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < G; ++e) {               // this lane's entries of the pass, ascending
                const T prod = vv[u].get(e) * xv[u][e];
                const T s = acc + prod;
                acc = G * (lane + 64 * (U * p + u)) + e < len ? s : acc;
            }
    }
    acc = wave_tree(acc);
    longrow_store<T>(lt, w, acc, y);
}

// x[wlo .. wlo + wlen) -> LDS at `xw` by LDS-DMA, 1-KiB pieces dealt to the four waves; workgroup-uniform; ends with a barrier
template <typename T> __device__ __forceinline__ void spmv_long_window(const T *__restrict__ x, T *xw, int wlo, int wlen)
{
    constexpr int XP = 1024 / (int)sizeof(T);
    const int t = threadIdx.x;
    for (int piece = t >> 6; piece * XP < wlen; piece += MIK_BLOCK / 64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + wlo + piece * XP + (t & 63) * (16 / (int)sizeof(T))),
                                         (__attribute__((address_space(3))) void *)(xw + piece * XP), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// long_win[workgroup] = first element of the workgroup's window of x (16-byte aligned, inside x) or -1; lw = its length (whole 1-KiB pieces)
template <typename T>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_longrows(LongTab lt, const int *__restrict__ col, const T *__restrict__ val,
                                                             const T *__restrict__ x, T *__restrict__ y, const int *__restrict__ done,
                                                             const int *__restrict__ long_win, int lw)
{
    if (done && *done) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char mik_dyn_lds[];
    T *xw = reinterpret_cast<T *>(mik_dyn_lds);
    const int wlo = long_win ? long_win[blockIdx.x] : -1;
    if (wlo >= 0) spmv_long_window<T>(x, xw, wlo, lw);
    const int wv = blockIdx.x * (MIK_BLOCK / 64) + (threadIdx.x >> 6);     // after the window, whole waves work alone
    spmv_longrow_wave<T>(wv, lt, col, val, x, y, xw, wlo >= 0 ? wlo : 0, wlo >= 0 ? lw : 0);
}

// MERGE_LONG: the first `nlb` workgroups of the launch are long-row workgroups (4 virtual rows each,
// scheduled first); the rest are row-block workgroups.
//
// XWIN (round 4): the gather of x from an LDS WINDOW.  On an irregular matrix with locality (an RCM-ordered FE
// matrix, the banded configs[4] stand-in) the lanes that stream consecutive entries gather from all over the row-block's
// column band: every lane its own cache line, and the texture path prices a gather per distinct line -- ~1 lane per clock and
// CU, 66 us for 34 M entries whatever surrounds them (scripts/micro/gather_random.hip), more than the operator streams cost.
// But the band of a 256-row block is small (256 + 2 x 2000 columns = 17 KB of fp32): the workgroup copies x[win_lo[rb] ..
// + win_span) into LDS once with LDS-DMA (1 KiB per wave-instruction, no registers) and every gather becomes a ds_read.  The
// window table is built at upload (csr_build_xwin: enabled when the row-blocks holding three quarters of the entries span at most
// 32 KB of x each; a block that spans more -- rows that wrap around the matrix -- carries win_lo < 0 and gathers from memory);
// same products, same order, same bits.
//
// RPERM (round 4): rows of very different lengths share a wave -- 64 lanes run the serial sum of the LONGEST of their rows: with 10 %
// of the rows at 50-200 entries among rows of 5-15, every wave of the banded configs[4] stand-in spent ~20 loop trips where its
// median row needs 2, and the kernel was bound by vector-ALU issue (20 M wave-instructions per launch, 35 us of the 64 us of the short
// part; profiles/r04_c5_banded_pmc_summary.txt).  rperm[r0 + t] (one byte per row, built at upload) gives thread t of a row-block the
// row whose length has rank t among the block's 256 -- the 64 longest rows meet in one wave (wave rb & 3, so the heavy waves spread
// over the four SIMDs), the other three finish after two trips.  Which thread sums a row never shows in the result; the fused dot
// puts x[r] * y[r] back into row order (through LDS) before the block tree, whose shape is defined over rows.
template <typename T, bool FUSE_DOT, bool NT, bool WIDE, bool MERGE_LONG, bool XWIN = false, bool RPERM = false>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_rowblock(int n, int nb, int map_mode, const int *__restrict__ rowptr,
                                                             const int *__restrict__ col, const T *__restrict__ val,
                                                             const T *__restrict__ x, T *__restrict__ y,
                                                             T *__restrict__ seg_out, const int *__restrict__ done,
                                                             const unsigned char *__restrict__ is_long, int nlb, LongTab lt,
                                                             const int *__restrict__ win_lo = nullptr, int win_span = 0,
                                                             const unsigned char *__restrict__ rperm = nullptr,
                                                             const int *__restrict__ long_win = nullptr, int lw = 0,
                                                             const T *__restrict__ ep_w = nullptr, const T *__restrict__ ep_c = nullptr,
                                                             const T *__restrict__ ep_z = nullptr)
{
    if (done && *done) return;
    constexpr int TILE = MIK_SPMV_TILE * (int)(8 / sizeof(T));   // 16 KB of LDS: 2048 fp64 / 4096 fp32 products
    constexpr int VW = WIDE ? VT<T>::W : 1;            // elements per lane per load
    constexpr int PER = TILE / (MIK_BLOCK * VW);       // loads per lane per tile
    // all LDS of this kernel is dynamic (mik_spmv_rowblock_lds): [prod: TILE + 8][4 wave sums, padded to 16 B][XWIN: win_span elements of x];
    // a long-row workgroup of a MERGE_LONG launch uses the whole of it as ITS window of x
    extern __shared__ __attribute__((aligned(16))) unsigned char mik_dyn_lds[];
    T *prod = reinterpret_cast<T *>(mik_dyn_lds);      // + 8: the row sums read whole groups of 8 (the surplus is never added)
    T *lds4 = prod + TILE + 8;

    const int t = threadIdx.x;
    int bid = blockIdx.x;
    if (MERGE_LONG) {
        // The long-row workgroups LEAD the launch, or -- lw < 0: the operator's long rows gather from LDS windows -- are spread evenly over
        // it (workgroup b is a long-row one where ceil(b nlb / total) steps up).  Measured in round 4 on the configs[4] stand-ins: banded
        // (windows) 70.3 us leading, 65.7 us spread; random (no windows: both parts compete for the same L2 gather bandwidth all the way)
        // 192 us leading, 196 us spread.
        const bool spread = lw < 0;
        if (spread) lw = -lw;
        unsigned which = (unsigned)bid;
        bool is_long = bid < nlb;
        if (spread) {
            const unsigned total = gridDim.x;
            const unsigned before = (unsigned)(((unsigned long long)bid * (unsigned)nlb + total - 1) / total);
            const unsigned upto = (unsigned)(((unsigned long long)(bid + 1) * (unsigned)nlb + total - 1) / total);
            is_long = upto > before;
            which = is_long ? before : (unsigned)bid - before;
        } else if (!is_long) which = (unsigned)(bid - nlb);
        if (is_long) {
            const int lwlo = long_win ? long_win[which] : -1;
            if (lwlo >= 0) spmv_long_window<T>(x, prod, lwlo, lw);
            spmv_longrow_wave<T>((int)which * (MIK_BLOCK / 64) + (t >> 6), lt, col, val, x, y, prod, lwlo >= 0 ? lwlo : 0, lwlo >= 0 ? lw : 0);
            return;
        }
        bid = (int)which;
    }
    const int rb = spmv_block_map(bid, nb, map_mode);
    const int r0 = rb * MIK_BLOCK;
    const int tr = RPERM ? (int)rperm[r0 + t] : t;     // the row of this thread inside the block (rperm is padded to whole blocks)
    const int r = r0 + tr;
    int ks = 0, ke = 0;
    if (r < n) { ks = rowptr[r]; ke = rowptr[r + 1]; }
    const int kb = rowptr[r0] & ~(VW - 1);             // tile start aligned for the wide loads
    const int kend = rowptr[min(r0 + MIK_BLOCK, n)];
    T *xw = lds4 + 16 / (int)sizeof(T) * ((4 * (int)sizeof(T) + 15) / 16);     // behind the wave sums, 16-byte aligned
    int wlo = 0;
    if (XWIN) {
        // the row-block's window of x -> LDS: wave wv issues the 1-KiB pieces wv, wv + 4, ... (win_span is a multiple of a piece;
        // win_lo[rb] is 16-byte aligned and the window lies inside x); waited for together with the first tile's streams
        constexpr int XP = 1024 / (int)sizeof(T);
        wlo = win_lo[rb];                               // < 0: this block's columns span more than a window -- it gathers from memory
        if (kb < kend && wlo >= 0)
            for (int piece = t >> 6; piece * XP < win_span; piece += MIK_BLOCK / 64)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + wlo + piece * XP + (t & 63) * (16 / (int)sizeof(T))),
                                                 (__attribute__((address_space(3))) void *)(xw + piece * XP), 16, 0, 0);
    }

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
            if (XWIN && wlo >= 0) {                     // workgroup-uniform
                if (kc == kb) {                         // the window has landed once every wave's DMA has
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int j = VW * (t + MIK_BLOCK * i);
                    if (j < cnt) {
#pragma unroll
                        for (int e = 0; e < VW; ++e) {
                            // the aligned tile start may cover up to VW - 1 entries of the row-block before: clamped, never added
                            const unsigned o = min((unsigned)(c[i][e] - wlo), (unsigned)(win_span - 1));
                            prod[j + e] = v[i][e] * xw[o];
                        }
                    }
                }
            } else {
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
        // ---- per-row serial sum, ascending column order: whole groups of 8 without a test, then the rest ----
        int a = max(ks, kc) - kc;
        int len = min(ke, kc + cnt) - kc - a;
        for (; len >= 8; len -= 8, a += 8) {
            T q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = prod[a + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = acc + q[i];
        }
        if (len > 0) {
            T q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = prod[a + i];          // may run past the row (and, by up to 7, past the tile): never added
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const T s2 = acc + q[i];
                acc = i < len ? s2 : acc;
            }
        }
        __syncthreads();
    }
    if (FUSE_DOT && ep_w && r < n) { const T te = *ep_c * ep_w[r]; acc = acc + te; }    // y = A x + c w (the Lanczos step of MINRES; operators without long rows)
    if (is_long && r < n && is_long[r]) acc = y[r];    // summed by k_spmv_longrows earlier on the stream
    else if (r < n) st_stream<NT>(y + r, acc);
    if (FUSE_DOT) {
        T p = T(0);
        if (r < n) p = (ep_z ? ep_z[r] : x[r]) * acc;       // ep_z: dot(z, y) instead of dot(x, y) (BiCGStab(l): z = r_shadow)
        if (RPERM) {                                   // back into row order: thread t of the block tree holds row r0 + t
            prod[tr] = p;
            __syncthreads();
            p = prod[t];
        }
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
// Measured at 256^3 fp64 inside the CG loop (round 2, interleaved rounds in one process): k_spmv_rowblock 302 us,
// this layout staged through registers 299 us, filled by LDS-DMA 291 us (278 us back to back = 6.25 TB/s).  Also
// built and measured: wave-private tiles (every wave stages its own 64 rows, no workgroup barrier at all) -- 354-380 us
// in all four variants (row / entry gather, padded or not): the barriers were not what the workgroup tile was
// waiting for, and per-element LDS writes cost more than they saved.
template <typename T, bool FUSE_DOT, bool NT>
__global__ __launch_bounds__(MIK_BLOCK, 6) void k_spmv_rowgather(int n, int rb0, int nb, int map_mode, const int *__restrict__ rowptr,
                                                              const int *__restrict__ col, const T *__restrict__ val,
                                                              const T *__restrict__ x, T *__restrict__ y, T *__restrict__ seg_out,
                                                              const int *__restrict__ done, const unsigned char *__restrict__ is_long,
                                                              const T *__restrict__ ep_w = nullptr, const T *__restrict__ ep_c = nullptr,
                                                             const T *__restrict__ ep_z = nullptr)
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
    if (FUSE_DOT && ep_w && r < n) { const T te = *ep_c * ep_w[r]; acc = acc + te; }    // y = A x + c w (the Lanczos step of MINRES; operators without long rows)
    if (is_long && r < n && is_long[r]) acc = y[r];    // summed by k_spmv_longrows earlier on the stream
    else if (r < n) st_stream<NT>(y + r, acc);
    if (FUSE_DOT) {
        T p = T(0);
        if (r < n) p = (ep_z ? ep_z[r] : x[r]) * acc;       // ep_z: dot(z, y) instead of dot(x, y) (BiCGStab(l): z = r_shadow)
        T tot = block_tree_256(p, lds4);
        if (t == 0) seg_out[rb] = tot;
    }
}

#endif  // __HIPCC__
