// mik_jds.h -- jagged slices: the SpMV layout for operators with long, near-uniform rows that are too long for the row-block
//              CSR tile (finite-element matrices: 50-80 entries per row).
//
// mul!(y, A, x) (SparseArrays mul!, called at src/cg.jl:54, src/gmres.jl:287; the inputs of benchmark/matrixmarket.jl:5-22).
//
// Layout (built at upload, csr_build_jds in mik_core.hip).  64 consecutive rows form a SLICE = the work of one wave, one row
// per lane.  A row's entries are cut into GROUPS of W = 16 B / sizeof(T) consecutive entries (the last group padded); group
// g of all the slice's rows that have one is stored contiguously, lane order, after group g - 1 -- a jagged-diagonal layout in
// units of groups, no padding between rows of different length.  A lane therefore reads its row's next W columns with ONE
// 16-byte (fp64: 8-byte) load and its next W values with ONE 16-byte load, both fully coalesced across the wave, gathers x for
// them and adds the products IN ASCENDING COLUMN ORDER from +0 -- exactly the order in which the reference's CSC column
// scatter reaches that row.  No LDS, no barrier, no row pointer: a lane finds its group by counting the active lanes below
// it (ballot + mbcnt), the wave advances by the number of active lanes.
//
// Why: a 64-lane vector-memory instruction is priced per instruction on this GPU (scripts/micro/gather_width.hip), so the
// operator streams must come as 16 bytes per lane; and neighbouring rows of an FE matrix reference neighbouring columns, so
// with one row per lane a gather instruction touches 2-3 cache lines.  The thread-per-row CSR tile (mik_spmv.h) keeps 25-40
// lanes of 256 busy on a 54-entry-per-row matrix (3D hexahedra, 3 unknowns per node, 62 M entries, fp32: 233 us; this
// layout 82 us = 0.77 of 8 TB/s; the 8-bit-code sliced-ELL form, which stores 37 % fewer bytes, 136 us).
//
// Measured and dropped: rows SORTED by length inside windows of 4096 / 4352 rows (SELL-C-sigma) so that the rows of a wave
// have similar lengths -- on the irregular configs[4] stand-ins (rows of 5-15 and 50-200 entries) 108 us (banded) / 222 us
// (random columns) against 70 / ~140 us for the CSR product tile: a wave that owns 64 rows of 125 entries runs 8-13 dependent
// load -> gather -> add rounds of ~10 us each under load while its lanes' rows are spread over the window (no common cache
// lines), where the tile's entry-parallel gathers stay inside one row's band.  (With 16 workgroups per window every heavy
// workgroup landed on XCDs 0 and 1 -- 362 us.)  The layout is therefore built only where the natural row order keeps the
// lanes busy (csr_build_jds: wave iterations within 25 % of the ideal).
//
// Rows longer than mik_spmv_long_row() stay with the wave-per-row path of mik_spmv.h (their lanes would idle for thousands of
// iterations); their workgroups lead the same launch (MERGE_LONG).
#pragma once
#include "mik_internal.h"
#include "mik_spmv.h"

constexpr int MIK_JDS_LONG = 0xFFFF;    // jlen marker: the row is summed by the long-row path

#ifdef __HIPCC__

constexpr int MIK_JDS_U = 4;            // groups per lane in flight per pass (2 / 4 / 8 measured on the 62 M-entry FE operator: 98.6 / 102.0 / 99.5 us)

template <typename T, bool FUSE_DOT, bool NT, bool MERGE_LONG>
__global__ __launch_bounds__(MIK_BLOCK) void k_spmv_jds(int n, int rb0, const int *__restrict__ jptr, const unsigned short *__restrict__ jlen,
                                                        const typename WideVec<T>::idx *__restrict__ jcol,
                                                        const typename WideVec<T>::val *__restrict__ jval, const T *__restrict__ x,
                                                        T *__restrict__ y, T *__restrict__ seg_out, const int *__restrict__ done, int nlb,
                                                        LongTab lt, const int *__restrict__ col, const T *__restrict__ val,
                                                        const T *__restrict__ ep_w = nullptr, const T *__restrict__ ep_c = nullptr,
                                                        const T *__restrict__ ep_z = nullptr)
{
    if (done && *done) return;
    constexpr int W = VT<T>::W, U = MIK_JDS_U;
    using IV = typename WideVec<T>::idx;
    using VV = typename WideVec<T>::val;
    const int t = threadIdx.x;
    int bid = blockIdx.x;
    if (MERGE_LONG) {
        if (bid < nlb) {       // long-row workgroups first
            spmv_longrow_wave<T>(bid * (MIK_BLOCK / 64) + (t >> 6), lt, col, val, x, y);
            return;
        }
        bid -= nlb;
    }
    const int rb = rb0 + bid;
    const int pos = rb * MIK_BLOCK + t;                 // this lane's row; the wave = slice pos >> 6
    int len = pos < n ? (int)jlen[pos] : MIK_JDS_LONG;
    const bool store = len != MIK_JDS_LONG;
    if (!store) len = 0;
    const int ng = (len + W - 1) / W;
    int base = __builtin_amdgcn_readfirstlane(jptr[pos >> 6]);

    T acc = T(0);
    for (int g0 = 0;; g0 += U) {
        const unsigned long long m0 = __ballot(g0 < ng);
        if (m0 == 0ull) break;                          // wave-uniform
        // Branch-free body: a lane without a group in pass u re-reads the wave's first group of that pass (the same cache
        // lines an active lane fetches anyway; past the last slice: the padded tail) and its products are never added --
        // exec-masked loads would make the compiler drain the memory queue between the passes.
        IV c[U];
        VV v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool act = g0 + u < ng;
            const unsigned long long m = u == 0 ? m0 : __ballot(act);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const int idx = base + (act ? rank : 0);
            c[u] = ld_stream<NT>(jcol + idx);
            v[u] = ld_stream<NT>(jval + idx);
            base += __popcll(m);
        }
        T xv[U][W];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < W; ++e) xv[u][e] = x[c[u][e]];          // padding entries carry the row's last real column
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const T p = v[u][e] * xv[u][e];
                const T s = acc + p;
                acc = (g0 + u) * W + e < len ? s : acc;
            }
    }
    if (FUSE_DOT && ep_w && pos < n) { const T te = *ep_c * ep_w[pos]; acc = acc + te; }   // y = A x + c w (the Lanczos step of MINRES; FUSE_DOT: no long rows)
    if (store) st_stream<NT>(y + pos, acc);
    if (FUSE_DOT) {            // no long rows: thread t holds the sum of row r0 + t -- the (1, 1) shape of include/mik.h
        __shared__ T lds4[4];
        T p = T(0);
        if (pos < n) p = (ep_z ? ep_z[pos] : x[pos]) * acc;   // ep_z: dot(z, y) instead of dot(x, y) (BiCGStab(l): z = r_shadow)
        const T tot = block_tree_256(p, lds4);
        if (t == 0) seg_out[rb] = tot;
    }
}

// dot(x, y) partials in the shape of the dot fused into the SpMV (one row per thread, one segment sum per 256-row block)
// for the operators whose SpMV cannot form them itself (split-off long rows land in y from other workgroups): same products,
// same tree.
template <typename T>
__global__ __launch_bounds__(MIK_BLOCK) void k_rowdot(int n, const T *__restrict__ x, const T *__restrict__ y, T *__restrict__ seg_out,
                                                      const int *__restrict__ done)
{
    if (done && *done) return;
    __shared__ T lds4[4];
    const int r = blockIdx.x * MIK_BLOCK + threadIdx.x;
    T p = T(0);
    if (r < n) p = x[r] * y[r];
    const T tot = block_tree_256(p, lds4);
    if (threadIdx.x == 0) seg_out[blockIdx.x] = tot;
}

#endif  // __HIPCC__
