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
// Rows with more than `long_row` entries (mik_csr_create; default MIK_LONG_ROW) are stored behind
// the short part and summed here -- a thread-per-row tile only pays while a 2048-entry tile covers
// many rows.  The sum stays SERIAL in ascending column order (bit-identical to the reference's
// scatter order): the wave streams the row in chunks of 64*U entries with coalesced loads, gathers
// x, parks the products in a wave-private LDS buffer, and lane 0 folds them into one accumulator with
// 16-byte LDS reads, software-pipelined one 32-value batch ahead of the dependent add chain.  The
// next chunk's val/col stream is issued before the chain starts; rows are ordered longest first
// (mik_csr_create) and different rows run concurrently on different waves, so the critical path is
// the longest single row at ~6 cycles per entry.
constexpr int MIK_LONG_U = 8;                          // entries per lane per chunk
constexpr int MIK_LONG_CH = 64 * MIK_LONG_U;           // 512-entry chunks

template <typename T>
__device__ __forceinline__ void spmv_longrow_wave(int w, T *__restrict__ wbuf, const int *__restrict__ rows,
                                                  const int *__restrict__ starts, const int *__restrict__ lens,
                                                  const int *__restrict__ col, const T *__restrict__ val,
                                                  const T *__restrict__ x, T *__restrict__ y)
{
    constexpr int U = MIK_LONG_U, CH = MIK_LONG_CH;
    constexpr int VW = VT<T>::W;                       // elements per 16-byte LDS read
    constexpr int B = 32 / VW;                         // 16-byte reads per 32-value batch
    using VV = typename WideVec<T>::val;
    const int lane = threadIdx.x & 63;
    const int k0 = starts[w], len = lens[w];
    T acc = T(0);
    // Three-stage software pipeline over chunks: while chunk c is multiplied and chained, the x-gather
    // of chunk c+1 and the val/col stream of chunk c+2 are in flight.
    T vA[U], xA[U], vB[U], vC[U];
    int cB[U], cC[U];
    auto stream = [&](int base, T(&vv)[U], int(&cc)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base + u * 64 + lane;
            vv[u] = j < len ? val[k0 + j] : T(0);
            cc[u] = j < len ? col[k0 + j] : 0;       // padding gathers x[0]; its product is replaced by +0 below
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
        for (int u = 0; u < U; ++u) {                   // chunk c: products (entries past the row end contribute +0)
            const T prod = vA[u] * xA[u];
            wbuf[u * 64 + lane] = (base + u * 64 + lane < len) ? prod : T(0);
        }
        // LDS operations of one wave execute in issue order, so lane 0's reads below see every lane's
        // writes above without a fence (a wavefront-scope fence would also drain vmcnt and with it the
        // loads in flight); the wave barrier only pins the compiler's instruction order.
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const VV *q = reinterpret_cast<const VV *>(wbuf);
            const int cnt = min(CH, len - base);                 // zero padding inside the last batch adds nothing
            // One v_add per entry plus one 16-byte LDS read per VW entries, fully unrolled: a single wave
            // issues roughly one instruction every 4-5 cycles, so instruction count and LDS latency are
            // the chain's cost.  Full chunks take the branch-free form, which lets the scheduler hoist
            // the next batches' LDS reads above the current adds.
            if (cnt == CH) {
#pragma unroll
                for (int g = 0; g < CH / 32; ++g) {
                    VV t[B];
#pragma unroll
                    for (int i = 0; i < B; ++i) t[i] = q[g * B + i];
#pragma unroll
                    for (int i = 0; i < B; ++i)
#pragma unroll
                        for (int e = 0; e < VW; ++e) acc = acc + t[i][e];
                }
            } else {
#pragma unroll
                for (int g = 0; g < CH / 32; ++g) {
                    if (g * 32 < cnt) {
                        VV t[B];
#pragma unroll
                        for (int i = 0; i < B; ++i) t[i] = q[g * B + i];
#pragma unroll
                        for (int i = 0; i < B; ++i)
#pragma unroll
                            for (int e = 0; e < VW; ++e) acc = acc + t[i][e];
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < U; ++u) { vA[u] = vB[u]; xA[u] = xB[u]; vB[u] = vC[u]; cB[u] = cC[u]; }
    }
    if (lane == 0) y[rows[w]] = acc;
}

template <typename T>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_longrows(int nlong, const int *__restrict__ rows, const int *__restrict__ starts,
                                                             const int *__restrict__ lens, const int *__restrict__ col,
                                                             const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y,
                                                             const int *__restrict__ done)
{
    if (done && *done) return;
    __shared__ __attribute__((aligned(16))) T buf[MIK_BLOCK / 64][MIK_LONG_CH];
    const int wv = threadIdx.x >> 6;
    const int w = blockIdx.x * (MIK_BLOCK / 64) + wv;
    if (w >= nlong) return;                            // whole waves leave: no block-level barrier is used
    spmv_longrow_wave<T>(w, buf[wv], rows, starts, lens, col, val, x, y);
}

// MERGE_LONG: the first `nlong_blocks` workgroups of the launch are long-row workgroups (4 rows each,
// scheduled first so the longest chains start earliest); the rest are row-block workgroups.
template <typename T, bool FUSE_DOT, bool NT, bool WIDE, bool MERGE_LONG>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_rowblock(int n, int nb, int map_mode, const int *__restrict__ rowptr,
                                                             const int *__restrict__ col, const T *__restrict__ val,
                                                             const T *__restrict__ x, T *__restrict__ y,
                                                             T *__restrict__ seg_out, const int *__restrict__ done,
                                                             const unsigned char *__restrict__ is_long, int nlong,
                                                             const int *__restrict__ long_tab)
{
    if (done && *done) return;
    constexpr int TILE = MIK_SPMV_TILE;
    constexpr int VW = WIDE ? VT<T>::W : 1;            // elements per lane per load
    constexpr int PER = TILE / (MIK_BLOCK * VW);       // loads per lane per tile
    static_assert(TILE >= (MIK_BLOCK / 64) * MIK_LONG_CH, "LDS tile doubles as the long-row wave buffers");
    __shared__ __attribute__((aligned(16))) T prod[TILE];
    __shared__ T lds4[4];

    const int t = threadIdx.x;
    int bid = blockIdx.x;
    if (MERGE_LONG) {
        const int nlb = (nlong + MIK_BLOCK / 64 - 1) / (MIK_BLOCK / 64);
        if (bid < nlb) {
            const int wv = t >> 6, w = bid * (MIK_BLOCK / 64) + wv;
            if (w < nlong) spmv_longrow_wave<T>(w, prod + wv * MIK_LONG_CH, long_tab, long_tab + nlong, long_tab + 2 * nlong, col, val, x, y);
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

#endif  // __HIPCC__
