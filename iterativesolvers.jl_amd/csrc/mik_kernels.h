// mik_kernels.h -- hand-written gfx950 kernels of the Krylov inner loop (HBM-bound; no MFMA).
//
// Every kernel is a 256-thread (4 wave-64) workgroup.  Element-wise kernels stream 16 bytes per
// lane per load; reductions follow the fixed-shape tree documented in include/mik.h so that the
// CPU oracle can reproduce them bit for bit.  The library is compiled with -ffp-contract=off:
// `a + b * c` below is always a rounded multiply followed by a rounded add, like the reference's
// Julia broadcast (src/cg.jl:51,58-59) and SparseArrays' `y[i] += a * x` scatter.
#pragma once
#include <algorithm>

#include "mik_internal.h"

#ifdef __HIPCC__

// =============================================================================================
// element-wise map (+ optional level-1 reduction)
// =============================================================================================
//
// Op interface:
//   static constexpr bool REDUCE;
//   __device__ void apply(int64_t i, T &acc) const;            // one element
//   __device__ void apply_vec(int64_t i, T &acc) const;        // W elements starting at i (16 B)
//
// Segment s = 256*W*L elements; thread t owns, for l = 0..L-1, the W elements starting at
// s*SEG + l*256*W + W*t (coalesced 16-byte loads), accumulated in that order.

template <typename T, bool VEC, typename Op>
__global__ __launch_bounds__(MIK_BLOCK) void k_map(int64_t n, int64_t nseg, Op op, T *__restrict__ seg_out,
                                                    const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr int W = VT<T>::W;
    constexpr int L = MIK_RED_L;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    __shared__ T lds4[4];
    for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int64_t base = s * SEG + (int64_t)W * threadIdx.x;
        T acc = T(0);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
            if (VEC && i + W <= n) {
                op.apply_vec(i, acc);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i + e < n) op.apply(i + e, acc);
            }
        }
        if (Op::REDUCE) {
            T tot = block_tree_256(acc, lds4);
            if (threadIdx.x == 0) seg_out[s] = tot;
        }
    }
}

// k_map whose coefficient is the level-2 sum of a PRODUCER kernel's segment sums, finalised by every
// workgroup itself (m <= 1024 partials: 4 loads + 4 wave trees per thread) instead of by a separate
// single-workgroup launch.  PRO = 1: coef = sum (a projection h_i of Gram-Schmidt); PRO = 2:
// coef = 1 / sqrt(sum) (the closing normalisation), with sqrt(sum) stored next to it.  Workgroup 0
// publishes the value(s) at `coef_out` for the host.  Halves the launch count of the MGS chain.
// X: what happens to the finalised sum before it is used -- nothing on one GPU (NoExchange); over a row partition every workgroup swaps it for
// the sum over the ranks (MailSum, csrc/mik_comm.hip: workgroup 0 posts this rank's total to the peers' mailboxes, every workgroup collects the P
// totals from its own rank's mailbox and adds them in rank order).
struct NoExchange {
    template <typename T> __device__ __forceinline__ T operator()(T cf) const { return cf; }
    template <typename T> __device__ __forceinline__ T pass(T cf, int, bool) const { return cf; }      // (the single-launch Gram-Schmidt: per pass)
};
template <typename T, bool VEC, typename Op, int PRO, typename X = NoExchange>
__global__ __launch_bounds__(MIK_BLOCK) void k_map_pro(int64_t n, int64_t nseg, Op op, T *__restrict__ seg_out,
                                                        const T *__restrict__ prev_part, int prev_m, T *__restrict__ coef_out, X xch)
{
    constexpr int W = VT<T>::W;
    constexpr int L = MIK_RED_L;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    __shared__ T lds16[16];
    __shared__ T lds4[4];
    // One segment per workgroup (n up to ~1M, the launch-bound regime): issue the vector loads BEFORE the
    // prologue, so their latency overlaps the prologue's load -> wave trees -> LDS chain instead of following it.
    const bool ahead = nseg <= (int64_t)gridDim.x && (int64_t)blockIdx.x < nseg;
    typename Op::Regs rg[L];
    bool full[L];
    if (ahead) {
        const int64_t base = (int64_t)blockIdx.x * SEG + (int64_t)W * threadIdx.x;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
            full[l] = VEC && i + W <= n;
            if (full[l]) op.load_vec(i, rg[l]);
        }
    }
    T cf = block_level2_256(prev_part, prev_m, lds16);
    cf = xch(cf);
    if (PRO == 2) {
        T nrm = mik_sqrt(cf);
        const bool ok = mik_nrm_in_range(cf);       // outside the safe range: leave w alone (* 1), the host rescales
        cf = ok ? T(1) / nrm : T(1);
        if (!ok) nrm = __builtin_nan("");
        if (blockIdx.x == 0 && threadIdx.x == 0) { coef_out[0] = nrm; coef_out[1] = cf; }
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        coef_out[0] = cf;
    }
    op.set_coef(cf);
    if (ahead) {
        const int64_t s = blockIdx.x;
        const int64_t base = s * SEG + (int64_t)W * threadIdx.x;
        T acc = T(0);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
            if (full[l]) {
                op.compute_vec(i, rg[l], acc);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i + e < n) op.apply(i + e, acc);
            }
        }
        if (Op::REDUCE) {
            T tot = block_tree_256(acc, lds4);
            if (threadIdx.x == 0) seg_out[s] = tot;
        }
        return;
    }
    for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int64_t base = s * SEG + (int64_t)W * threadIdx.x;
        T acc = T(0);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
            if (VEC && i + W <= n) {
                op.apply_vec(i, acc);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i + e < n) op.apply(i + e, acc);
            }
        }
        if (Op::REDUCE) {
            T tot = block_tree_256(acc, lds4);
            if (threadIdx.x == 0) seg_out[s] = tot;
        }
    }
}

template <typename T, int PRO, typename Op, typename X = NoExchange>
static inline int launch_map_pro(mik_ctx *ctx, int64_t n, Op op, bool vec, T *seg_out, const T *prev_part, int prev_m, T *coef_out, X xch = X())
{
    const int64_t nseg = mik_nseg<T>(n);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(nseg, mik_max_grid(ctx)));   // >= 1: workgroup 0 publishes the coefficient
    if (vec)
        hipLaunchKernelGGL((k_map_pro<T, true, Op, PRO, X>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, op, seg_out, prev_part, prev_m, coef_out, xch);
    else
        hipLaunchKernelGGL((k_map_pro<T, false, Op, PRO, X>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, op, seg_out, prev_part, prev_m, coef_out, xch);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// host-side launcher of k_map: one workgroup per segment, capped grid with a grid-stride loop
template <typename T, typename Op>
static inline int launch_map(mik_ctx *ctx, int64_t n, Op op, bool vec, T *seg_out, const int *done)
{
    const int64_t nseg = mik_nseg<T>(n);
    if (nseg == 0) return MIK_OK;
    const int grid = (int)std::min<int64_t>(nseg, mik_max_grid(ctx));
    if (vec)
        hipLaunchKernelGGL((k_map<T, true, Op>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, op, seg_out, done);
    else
        hipLaunchKernelGGL((k_map<T, false, Op>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, op, seg_out, done);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// k_map with TWO reductions over the same sweep (Op::apply / apply_vec take two accumulators): both keep the segment / thread /
// tree shape of the single-reduction kernel, so each sum has the bits it would have in a sweep of its own.
template <typename T, bool VEC, typename Op>
__global__ __launch_bounds__(MIK_BLOCK) void k_map2(int64_t n, int64_t nseg, Op op, T *__restrict__ seg1, T *__restrict__ seg2,
                                                     const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr int W = VT<T>::W;
    constexpr int L = MIK_RED_L;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    __shared__ T lds4[4];
    for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int64_t base = s * SEG + (int64_t)W * threadIdx.x;
        T a1 = T(0), a2 = T(0);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
            if (VEC && i + W <= n) {
                op.apply_vec(i, a1, a2);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i + e < n) op.apply(i + e, a1, a2);
            }
        }
        const T t1 = block_tree_256(a1, lds4);
        const T t2 = block_tree_256(a2, lds4);
        if (threadIdx.x == 0) { seg1[s] = t1; seg2[s] = t2; }
    }
}

template <typename T, typename Op>
static inline int launch_map2(mik_ctx *ctx, int64_t n, Op op, bool vec, T *seg1, T *seg2, const int *done)
{
    const int64_t nseg = mik_nseg<T>(n);
    if (nseg == 0) return MIK_OK;
    const int grid = (int)std::min<int64_t>(nseg, mik_max_grid(ctx));
    if (vec) hipLaunchKernelGGL((k_map2<T, true, Op>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, op, seg1, seg2, done);
    else hipLaunchKernelGGL((k_map2<T, false, Op>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, op, seg1, seg2, done);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// k_map whose coefficient comes from the level-2 sum of the PRODUCER's segment sums, formed by every workgroup itself (m <= 1024
// partials: block_level2_256 = the tree of k_finalize_store) and turned into the sweep's coefficient by `pro` -- the scalar
// statement of the reference that sits between the reduction and the sweep; workgroup 0 publishes what later kernels need.
// One launch instead of reduction finaliser + sweep: at the sizes where an iteration is launch-bound that is where its time goes.
template <typename T, bool VEC, typename Op, typename Pro>
__global__ __launch_bounds__(MIK_BLOCK) void k_map_with(int64_t n, int64_t nseg, Op op, Pro pro, const T *__restrict__ part, int m, T *__restrict__ seg_out)
{
    constexpr int W = VT<T>::W;
    constexpr int L = MIK_RED_L;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    __shared__ T lds16[16];
    __shared__ T lds4[4];
    const T tot = block_level2_256(part, m, lds16);
    pro(tot, op, blockIdx.x == 0 && threadIdx.x == 0);
    for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int64_t base = s * SEG + (int64_t)W * threadIdx.x;
        T acc = T(0);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
            if (VEC && i + W <= n) {
                op.apply_vec(i, acc);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i + e < n) op.apply(i + e, acc);
            }
        }
        if (Op::REDUCE) {
            const T t2 = block_tree_256(acc, lds4);
            if (threadIdx.x == 0) seg_out[s] = t2;
        }
    }
}

template <typename T, typename Op, typename Pro>
static inline int launch_map_with(mik_ctx *ctx, int64_t n, Op op, Pro pro, bool vec, const T *part, int m, T *seg_out)
{
    const int64_t nseg = mik_nseg<T>(n);
    if (nseg == 0) return MIK_OK;
    const int grid = (int)std::min<int64_t>(nseg, mik_max_grid(ctx));
    if (vec) hipLaunchKernelGGL((k_map_with<T, true, Op, Pro>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, op, pro, part, m, seg_out);
    else hipLaunchKernelGGL((k_map_with<T, false, Op, Pro>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, op, pro, part, m, seg_out);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

template <typename T> __device__ __forceinline__ typename VT<T>::vec vload(const T *p)
{
    return *reinterpret_cast<const typename VT<T>::vec *>(p);
}
template <typename T> __device__ __forceinline__ void vstore(T *p, typename VT<T>::vec v)
{
    *reinterpret_cast<typename VT<T>::vec *>(p) = v;
}
template <typename T> __device__ __forceinline__ T &el(typename VT<T>::vec &v, int e)
{
    return reinterpret_cast<T *>(&v)[e];
}
// non-temporal 16-byte accesses for operands that are not needed again before they would be evicted anyway
// (keeps the vectors that ARE re-read by the next kernel in L2 / Infinity Cache)
template <typename T> struct NativeVec;
template <> struct NativeVec<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct NativeVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <typename T> __device__ __forceinline__ typename VT<T>::vec vload_nt(const T *p)
{
    using NV = typename NativeVec<T>::type;
    NV v = __builtin_nontemporal_load(reinterpret_cast<const NV *>(p));
    typename VT<T>::vec out;
#pragma unroll
    for (int e = 0; e < VT<T>::W; ++e) reinterpret_cast<T *>(&out)[e] = v[e];
    return out;
}
template <typename T> __device__ __forceinline__ void vstore_nt(T *p, typename VT<T>::vec v)
{
    using NV = typename NativeVec<T>::type;
    NV o;
#pragma unroll
    for (int e = 0; e < VT<T>::W; ++e) o[e] = reinterpret_cast<const T *>(&v)[e];
    __builtin_nontemporal_store(o, reinterpret_cast<NV *>(p));
}

// Helper macro: define apply_vec in terms of a per-element lambda over loaded vectors is not
// possible generically, so each op spells out its loads (all issued before the arithmetic).

// y .= x .+ beta .* y            -- src/cg.jl:51  (u .= r .+ beta .* u), :86
template <typename T> struct OpXpby {
    static constexpr bool REDUCE = false;
    const T *__restrict__ x; T *__restrict__ y; Coef<T> beta; int nt = 0;   // nt & 1: x is streamed (non-temporal load)
    __device__ __forceinline__ void apply(int64_t i, T &) const { T t = beta.get() * y[i]; y[i] = x[i] + t; }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        const T b = beta.get();
        auto xv = (nt & 1) ? vload_nt(x + i) : vload(x + i);
        auto yv = (nt & 2) ? vload_nt<T>(y + i) : vload<T>(y + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T t = b * el<T>(yv, e); el<T>(yv, e) = el<T>(xv, e) + t; }
        if (nt & 4) vstore_nt(y + i, yv); else vstore(y + i, yv);
    }
};

// y .= x .+ beta .* y; partial sums of y.^2      -- the bidiagonalisation updates of LSQR / LSMR with their norms:
//   u .= -alpha .* u .+ tmpm; norm(u) (src/lsqr.jl:151-152), v .= -beta .* v .+ tmpn; norm(v) (:159-160); src/lsmr.jl:161-162, :167-168
template <typename T> struct OpXpbyNrm {
    static constexpr bool REDUCE = true;
    const T *__restrict__ x; T *__restrict__ y; T beta;
    __device__ __forceinline__ void apply(int64_t i, T &acc) const { T t = beta * y[i]; const T v = x[i] + t; y[i] = v; T p = v * v; acc = acc + p; }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        auto xv = vload_nt(x + i); auto yv = vload<T>(y + i);          // x (the product just formed) is dead afterwards
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T t = beta * el<T>(yv, e); el<T>(yv, e) = el<T>(xv, e) + t; }
        vstore(y + i, yv);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T p = el<T>(yv, e) * el<T>(yv, e); acc = acc + p; }
    }
};

// LSQR, the tail of an iteration in one sweep        -- src/lsqr.jl:189-192
//   x .+= t1*w;  w = t2 .* w .+ v;  wrho .= w .* inv(rho) (not stored);  partial sums of wrho.^2
template <typename T> struct OpLsqrUpdate {
    static constexpr bool REDUCE = true;
    T *__restrict__ x; T *__restrict__ w; const T *__restrict__ v; T t1, t2, inv_rho;
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        const T wo = w[i];
        T a = t1 * wo; x[i] = x[i] + a;
        T b = t2 * wo; const T wn = b + v[i]; w[i] = wn;
        const T r = wn * inv_rho;
        T p = r * r; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        constexpr int W = VT<T>::W;
        auto xv = vload_nt<T>(x + i); auto wv = vload<T>(w + i); auto vv = vload(v + i);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const T wo = el<T>(wv, e);
            T a = t1 * wo; el<T>(xv, e) = el<T>(xv, e) + a;
            T b = t2 * wo; el<T>(wv, e) = b + el<T>(vv, e);
        }
        vstore_nt(x + i, xv); vstore(w + i, wv);                         // x is touched once per iteration: streamed both ways
#pragma unroll
        for (int e = 0; e < W; ++e) { const T r = el<T>(wv, e) * inv_rho; T p = r * r; acc = acc + p; }
    }
};

// LSMR, the three vector updates of an iteration in one sweep, with norm(x)        -- src/lsmr.jl:199-201, :242
//   hbar .= hbar .* c1 .+ h;  x .+= c2 * hbar;  h .= h .* c3 .+ v;  partial sums of x.^2
template <typename T> struct OpLsmrUpdate {
    static constexpr bool REDUCE = true;
    T *__restrict__ hbar; T *__restrict__ h; T *__restrict__ x; const T *__restrict__ v; T c1, c2, c3;
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        const T ho = h[i];
        T a = hbar[i] * c1; const T hb = a + ho; hbar[i] = hb;
        T b = c2 * hb; const T xn = x[i] + b; x[i] = xn;
        T c = ho * c3; h[i] = c + v[i];
        T p = xn * xn; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        constexpr int W = VT<T>::W;
        auto hbv = vload<T>(hbar + i); auto hv = vload<T>(h + i); auto xv = vload<T>(x + i); auto vv = vload(v + i);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const T ho = el<T>(hv, e);
            T a = el<T>(hbv, e) * c1; el<T>(hbv, e) = a + ho;
            T b = c2 * el<T>(hbv, e); el<T>(xv, e) = el<T>(xv, e) + b;
            T c = ho * c3; el<T>(hv, e) = c + el<T>(vv, e);
        }
        vstore(hbar + i, hbv); vstore(h + i, hv); vstore(x + i, xv);
#pragma unroll
        for (int e = 0; e < W; ++e) { T p = el<T>(xv, e) * el<T>(xv, e); acc = acc + p; }
    }
};

// QMR's two-sided Lanczos step (src/qmr.jl:70-72, :76-78, :81): y .+= a .* x1; y .+= b .* x2 (skipped when x2 is null), then partial sums of
// z .* y (skipped when z is null) -- two axpy! and the dot that follows them in one sweep
template <typename T> struct OpAxpy2Dot {
    static constexpr bool REDUCE = true;
    T *__restrict__ y; const T *__restrict__ x1; const T *__restrict__ x2; const T *__restrict__ z; T a, b;
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        T t = a * x1[i]; T v = y[i] + t;
        if (x2) { T u = b * x2[i]; v = v + u; }
        y[i] = v;
        if (z) { T p = v * z[i]; acc = acc + p; }
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        constexpr int W = VT<T>::W;
        auto yv = vload<T>(y + i); auto xv = vload(x1 + i);
#pragma unroll
        for (int e = 0; e < W; ++e) { T t = a * el<T>(xv, e); el<T>(yv, e) = el<T>(yv, e) + t; }
        if (x2) {
            auto x2v = vload(x2 + i);
#pragma unroll
            for (int e = 0; e < W; ++e) { T u = b * el<T>(x2v, e); el<T>(yv, e) = el<T>(yv, e) + u; }
        }
        vstore(y + i, yv);
        if (z) {
            auto zv = vload(z + i);
#pragma unroll
            for (int e = 0; e < W; ++e) { T p = el<T>(yv, e) * el<T>(zv, e); acc = acc + p; }
        }
    }
};

// x .*= a; y .*= b      -- rmul!(v_next, inv(δ)); rmul!(w_next, inv(β)): src/qmr.jl:90-91
template <typename T> struct OpScal2 {
    static constexpr bool REDUCE = false;
    T *__restrict__ x; T *__restrict__ y; T a, b;
    __device__ __forceinline__ void apply(int64_t i, T &) const { x[i] = x[i] * a; y[i] = y[i] * b; }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        auto xv = vload<T>(x + i); auto yv = vload<T>(y + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { el<T>(xv, e) = el<T>(xv, e) * a; el<T>(yv, e) = el<T>(yv, e) * b; }
        vstore(x + i, xv); vstore(y + i, yv);
    }
};

// QMR, the tail of an iteration in one sweep        -- src/qmr.jl:188-197
//   p = v; p .+= (-h1) .* p_curr (if p_curr); p .+= (-h0) .* p_prev (if p_prev); p .*= inv; x .+= g .* p; the result is the next p_curr
//   (stored over p_out, which may be p_prev's storage: the caller rotates its names instead of the reference's two copies)
template <typename T> struct OpQmrUpdate {
    static constexpr bool REDUCE = false;
    const T *__restrict__ v; const T *p_curr; const T *p_prev; T *p_out; T *__restrict__ x; T neg_h1, neg_h0, inv, g;
    __device__ __forceinline__ void apply(int64_t i, T &) const
    {
        T p = v[i];
        if (p_curr) { T t = neg_h1 * p_curr[i]; p = p + t; }
        if (p_prev) { T t = neg_h0 * p_prev[i]; p = p + t; }
        p = p * inv;
        p_out[i] = p;
        T t = g * p; x[i] = x[i] + t;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        constexpr int W = VT<T>::W;
        auto pv = vload(v + i); auto xv = vload_nt<T>(x + i);
        if (p_curr) {
            auto cv = vload(p_curr + i);
#pragma unroll
            for (int e = 0; e < W; ++e) { T t = neg_h1 * el<T>(cv, e); el<T>(pv, e) = el<T>(pv, e) + t; }
        }
        if (p_prev) {
            auto qv = vload(p_prev + i);
#pragma unroll
            for (int e = 0; e < W; ++e) { T t = neg_h0 * el<T>(qv, e); el<T>(pv, e) = el<T>(pv, e) + t; }
        }
#pragma unroll
        for (int e = 0; e < W; ++e) { el<T>(pv, e) = el<T>(pv, e) * inv; T t = g * el<T>(pv, e); el<T>(xv, e) = el<T>(xv, e) + t; }
        vstore(p_out + i, pv); vstore_nt(x + i, xv);
    }
};

// The CG step with the update of x moved one sweep later (same operands, same rounding, so the same bits): the tail of
// step k only does r .-= alpha .* c and |r|^2 (OpCgUpdateR), and x .+= alpha_k .* u_k is applied by the sweep that reads
// u_k anyway -- u = r + beta u of step k + 1 -- saving one read of u (n s bytes, 19 us at 256^3) per iteration.  That sweep is
// enqueued before iterate(k) returns (cg_enqueue_head), so x is up to date for every stream-ordered reader.  `pending`
// (device flag, set by the tail's finaliser, cleared by the next finaliser) says whether the x update is due; when the
// iteration has stopped (`done`) the sweep applies it and leaves u alone.
template <typename T> struct OpXpbyX {
    static constexpr bool REDUCE = false;
    const T *__restrict__ r; T *__restrict__ u; T *__restrict__ x; Coef<T> beta, alpha;
    const int *__restrict__ done; const int *__restrict__ pending; int nt = 0;   // nt & 1: r streamed; nt & 8: x streamed
    __device__ __forceinline__ void apply(int64_t i, T &) const
    {
        const int dn = *done, pd = *pending;
        if (dn && !pd) return;
        const T uo = u[i];
        if (pd) { T t = alpha.get() * uo; x[i] = x[i] + t; }
        if (!dn) { T t = beta.get() * uo; u[i] = r[i] + t; }
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        const int dn = *done, pd = *pending;
        if (dn && !pd) return;
        auto uv = (nt & 2) ? vload_nt<T>(u + i) : vload<T>(u + i);
        if (pd) {
            const T a = alpha.get();
            auto xv = (nt & 8) ? vload_nt<T>(x + i) : vload<T>(x + i);
#pragma unroll
            for (int e = 0; e < VT<T>::W; ++e) { T t = a * el<T>(uv, e); el<T>(xv, e) = el<T>(xv, e) + t; }
            if (nt & 8) vstore_nt(x + i, xv); else vstore(x + i, xv);
        }
        if (!dn) {
            const T b = beta.get();
            auto rv = (nt & 1) ? vload_nt(r + i) : vload(r + i);
#pragma unroll
            for (int e = 0; e < VT<T>::W; ++e) { T t = b * el<T>(uv, e); el<T>(uv, e) = el<T>(rv, e) + t; }
            if (nt & 4) vstore_nt(u + i, uv); else vstore(u + i, uv);
        }
    }
};

// r .-= alpha .* c; partial sums of r.^2        -- src/cg.jl:59-62 (the x half of :58 rides on OpXpbyX)
// alpha of a row-partitioned step formed by every thread itself from the P per-rank partial sums of dot(u, c) (rank order) and
// the residual norm: alpha = res^2 / dot(u, c) (src/cg.jl:55) -- the arithmetic of k_cgd_alpha without its launch
template <typename T> struct CoefAlphaRanks {
    const T *all; int nranks; const T *res;
    __device__ __forceinline__ T get() const
    {
        T tot = all[0];
        for (int p = 1; p < nranks; ++p) tot = tot + all[p];
        const T r = *res;
        return (r * r) / tot;
    }
};
template <typename T, typename C = Coef<T>> struct OpCgUpdateR {
    static constexpr bool REDUCE = true;
    T *__restrict__ r; const T *__restrict__ c; C alpha; int nt = 0;              // nt & 2: c streamed; nt & 8 / 16: r load / store streamed
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        T s = alpha.get() * c[i]; T rn = r[i] - s; r[i] = rn;
        T p = rn * rn; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        const T a = alpha.get();
        auto rv = (nt & 8) ? vload_nt<T>(r + i) : vload<T>(r + i);
        auto cv = (nt & 2) ? vload_nt(c + i) : vload(c + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T s = a * el<T>(cv, e); el<T>(rv, e) = el<T>(rv, e) - s; }
        if (nt & 16) vstore_nt(r + i, rv); else vstore(r + i, rv);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T p = el<T>(rv, e) * el<T>(rv, e); acc = acc + p; }
    }
};

// The PCG step with a diagonal Pl (src/cg.jl:72-100) in TWO vector sweeps instead of three: the tail forms c = Pl \\ r of the NEXT
// step on the r it has just updated -- stored over c = A u, which is dead by then -- and reduces dot(c, r) next to |r|^2
// (OpPcgUpdateR, two reductions); the head is the plain OpXpbyX on that c.  The same quotients, products and orders as
// OpJacobiDot / OpCgUpdateR, so the same bits; 10 n instead of 11 n scalars per iteration and two launches less.
// (Recomputing r ./ d in the head instead of storing it -- 10 n too, one stream less -- was measured: 408 us against 315 us per
// iteration at 256^3; two correctly rounded fp64 divisions per element make both sweeps instruction-bound.)
template <typename T> struct OpPcgUpdateR {
    static constexpr bool REDUCE = true;
    T *__restrict__ r; T *__restrict__ c; const T *__restrict__ d; Coef<T> alpha; int nt = 0;   // nt & 8 / 16: r load / store streamed; 32 / 64: c store / load streamed
    __device__ __forceinline__ void apply(int64_t i, T &a1, T &a2) const
    {
        T s = alpha.get() * c[i]; T rn = r[i] - s; r[i] = rn;
        T p = rn * rn; a1 = a1 + p;
        T v = rn / d[i]; c[i] = v;
        T q = v * rn; a2 = a2 + q;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &a1, T &a2) const
    {
        const T a = alpha.get();
        auto rv = (nt & 8) ? vload_nt<T>(r + i) : vload<T>(r + i);
        auto cv = (nt & 64) ? vload_nt<T>(c + i) : vload<T>(c + i);
        auto dv = vload_nt(d + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T s = a * el<T>(cv, e); el<T>(rv, e) = el<T>(rv, e) - s; }
        if (nt & 16) vstore_nt(r + i, rv); else vstore(r + i, rv);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T p = el<T>(rv, e) * el<T>(rv, e); a1 = a1 + p; }
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) el<T>(cv, e) = el<T>(rv, e) / el<T>(dv, e);
        if (nt & 32) vstore_nt(c + i, cv); else vstore(c + i, cv);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T q = el<T>(cv, e) * el<T>(rv, e); a2 = a2 + q; }
    }
};

// x .+= alpha .* u if the update is still due (no sweep over u followed the last step)
template <typename T> struct OpXFlush {
    static constexpr bool REDUCE = false;
    const T *__restrict__ u; T *__restrict__ x; Coef<T> alpha; const int *__restrict__ pending;
    __device__ __forceinline__ void apply(int64_t i, T &) const
    {
        if (!*pending) return;
        T t = alpha.get() * u[i]; x[i] = x[i] + t;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        if (!*pending) return;
        const T a = alpha.get();
        auto uv = vload(u + i); auto xv = vload<T>(x + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T t = a * el<T>(uv, e); el<T>(xv, e) = el<T>(xv, e) + t; }
        vstore(x + i, xv);
    }
};

// y .+= alpha .* x               -- src/cg.jl:58
template <typename T> struct OpAxpy {
    static constexpr bool REDUCE = false;
    const T *__restrict__ x; T *__restrict__ y; Coef<T> alpha;
    __device__ __forceinline__ void apply(int64_t i, T &) const { T t = alpha.get() * x[i]; y[i] = y[i] + t; }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        const T a = alpha.get();
        auto xv = vload(x + i); auto yv = vload<T>(y + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T t = a * el<T>(xv, e); el<T>(yv, e) = el<T>(yv, e) + t; }
        vstore(y + i, yv);
    }
};

// BiCGStab(l), the two block updates of the BiCG part in one launch each (mik_bicgstab_step) -- per element exactly OpXpby /
// OpAxpy column by column:
//   us[:, 1:j] .= rs[:, 1:j] .- beta .* us[:, 1:j]                                  -- src/bicgstabl.jl:93   (neg_beta = -beta)
//   rs[:, 1:j] .-= alpha .* us[:, 2:j+1];  x .+= alpha .* us[:, 1]                   -- :103, :111 (x does not depend on :107)
template <typename T> struct OpBicgU {
    static constexpr bool REDUCE = false;
    T *__restrict__ us; int64_t ldu; const T *__restrict__ rs; int64_t ldr; int ncols; Coef<T> neg_beta;
    int nt = 0;    // bits: 1 loads streamed, 2 stores of all but the last column streamed, 4 store of the last column streamed (mik_bicgstab_step: 7).  (3 =
                   // everything streamed except the store of the LAST column -- the input of the SpMV that follows, which the Infinity Cache
                   // should keep (the SpMV of the CG loop takes 47 us on a cached input, 75 us behind a sweep that left its tails there) -- was the first
                   // guess; measured, 7 is 1 % faster.)
    __device__ __forceinline__ void apply(int64_t i, T &) const
    {
        const T b = neg_beta.get();
        for (int q = 0; q < ncols; ++q) { T *u = us + q * ldu; T t = b * u[i]; u[i] = rs[q * ldr + i] + t; }
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        const T b = neg_beta.get();
        for (int q = 0; q < ncols; ++q) {
            T *u = us + q * ldu;
            auto xv = (nt & 1) ? vload_nt(rs + q * ldr + i) : vload(rs + q * ldr + i);
            auto yv = (nt & 1) ? vload_nt<T>(u + i) : vload<T>(u + i);
#pragma unroll
            for (int e = 0; e < VT<T>::W; ++e) { T t = b * el<T>(yv, e); el<T>(yv, e) = el<T>(xv, e) + t; }
            if (q + 1 < ncols ? (nt & 2) : (nt & 4)) vstore_nt(u + i, yv); else vstore(u + i, yv);
        }
    }
};
template <typename T> struct OpBicgR {
    static constexpr bool REDUCE = false;
    const T *__restrict__ us; int64_t ldu; T *__restrict__ rs; int64_t ldr; int ncols; T *__restrict__ x; Coef<T> neg_alpha, alpha;
    int nt = 0;    // the bits of OpBicgU (last column = the last residual column, the input of the SpMV that follows; x counts as another column)
    __device__ __forceinline__ void apply(int64_t i, T &) const
    {
        const T na = neg_alpha.get(), a = alpha.get();
        for (int q = 0; q < ncols; ++q) { T t = na * us[(q + 1) * ldu + i]; rs[q * ldr + i] = rs[q * ldr + i] + t; }
        { T t = a * us[i]; x[i] = x[i] + t; }
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        const T na = neg_alpha.get(), a = alpha.get();
        for (int q = 0; q < ncols; ++q) {
            auto xv = (nt & 1) ? vload_nt(us + (q + 1) * ldu + i) : vload(us + (q + 1) * ldu + i);
            auto yv = (nt & 1) ? vload_nt<T>(rs + q * ldr + i) : vload<T>(rs + q * ldr + i);
#pragma unroll
            for (int e = 0; e < VT<T>::W; ++e) { T t = na * el<T>(xv, e); el<T>(yv, e) = el<T>(yv, e) + t; }
            if (q + 1 < ncols ? (nt & 2) : (nt & 4)) vstore_nt(rs + q * ldr + i, yv); else vstore(rs + q * ldr + i, yv);
        }
        auto uv = (nt & 1) ? vload_nt(us + i) : vload(us + i);
        auto xx = (nt & 1) ? vload_nt<T>(x + i) : vload<T>(x + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T t = a * el<T>(uv, e); el<T>(xx, e) = el<T>(xx, e) + t; }
        if (nt & 2) vstore_nt(x + i, xx); else vstore(x + i, xx);
    }
};

// y .-= x                        -- src/cg.jl:138, src/gmres.jl:246
template <typename T> struct OpSub {
    static constexpr bool REDUCE = false;
    const T *__restrict__ x; T *__restrict__ y;
    __device__ __forceinline__ void apply(int64_t i, T &) const { y[i] = y[i] - x[i]; }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        auto xv = vload(x + i); auto yv = vload<T>(y + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) el<T>(yv, e) = el<T>(yv, e) - el<T>(xv, e);
        vstore(y + i, yv);
    }
};

// x .*= alpha                    -- src/orthogonalize.jl:76, src/gmres.jl:253
template <typename T> struct OpScal {
    static constexpr bool REDUCE = false;
    T *__restrict__ x; Coef<T> alpha;
    __device__ __forceinline__ void set_coef(T c) { alpha.ptr = nullptr; alpha.val = c; }
    struct Regs { typename VT<T>::vec xv; };                  // load / compute halves of apply_vec (k_map_pro)
    __device__ __forceinline__ void load_vec(int64_t i, Regs &r) const { r.xv = vload<T>(x + i); }
    __device__ __forceinline__ void compute_vec(int64_t i, Regs &r, T &) const
    {
        const T a = alpha.get();
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) el<T>(r.xv, e) = el<T>(r.xv, e) * a;
        vstore(x + i, r.xv);
    }
    __device__ __forceinline__ void apply(int64_t i, T &) const { x[i] = x[i] * alpha.get(); }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        const T a = alpha.get();
        auto xv = vload<T>(x + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) el<T>(xv, e) = el<T>(xv, e) * a;
        vstore(x + i, xv);
    }
};

// y .= x ./ d                    -- ldiv!(y, P::JacobiPrec, x), test/cg.jl:18
template <typename T> struct OpDivide {
    static constexpr bool REDUCE = false;
    const T *x; const T *d; T *y;   // y may alias x
    __device__ __forceinline__ void apply(int64_t i, T &) const { y[i] = x[i] / d[i]; }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        auto xv = vload(x + i); auto dv = vload(d + i); typename VT<T>::vec yv;
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) el<T>(yv, e) = el<T>(xv, e) / el<T>(dv, e);
        vstore(y + i, yv);
    }
};

// x .= value                     -- src/cg.jl:129 (u .= 0)
template <typename T> struct OpFill {
    static constexpr bool REDUCE = false;
    T *__restrict__ x; T value;
    __device__ __forceinline__ void apply(int64_t i, T &) const { x[i] = value; }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        typename VT<T>::vec v;
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) el<T>(v, e) = value;
        vstore(x + i, v);
    }
};

// partial sums of x .* y         -- dot(x, y): src/cg.jl:55, src/orthogonalize.jl:71
template <typename T> struct OpDot {
    static constexpr bool REDUCE = true;
    const T *__restrict__ x; const T *__restrict__ y;
    __device__ __forceinline__ void apply(int64_t i, T &acc) const { T p = x[i] * y[i]; acc = acc + p; }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        auto xv = vload(x + i); auto yv = vload(y + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T p = el<T>(xv, e) * el<T>(yv, e); acc = acc + p; }
    }
};

// out .= a .- b (b may be null: out .= a); partial sums of out.^2
//   -- r = b - A*x and norm(r): src/cg.jl:130,138,140; src/gmres.jl:241,246,252
template <typename T> struct OpSubNrm {
    static constexpr bool REDUCE = true;
    const T *__restrict__ a; const T *__restrict__ b; T *__restrict__ out;
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        T v = b ? a[i] - b[i] : a[i];
        out[i] = v;
        T p = v * v; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        auto av = vload(a + i);
        if (b) {
            auto bv = vload(b + i);
#pragma unroll
            for (int e = 0; e < VT<T>::W; ++e) el<T>(av, e) = el<T>(av, e) - el<T>(bv, e);
        }
        vstore(out + i, av);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T p = el<T>(av, e) * el<T>(av, e); acc = acc + p; }
    }
};

// partial sums of (x .* s).^2 -- the scaled pass of the over-/underflow-safe norm (s = exact power of two)
template <typename T> struct OpScaledSq {
    static constexpr bool REDUCE = true;
    const T *__restrict__ x; T s;
    __device__ __forceinline__ void apply(int64_t i, T &acc) const { T q = x[i] * s; T p = q * q; acc = acc + p; }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        auto xv = vload(x + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T q = el<T>(xv, e) * s; T p = q * q; acc = acc + p; }
    }
};

// max |x_i| with NaN propagation (order-independent, so any launch geometry gives the same value):
// per-workgroup maxima to part[], then one workgroup folds them (launched as <<<1, 256>>> with m partials).
template <typename T> __device__ __forceinline__ T amax_fold(T m, T a) { return (a > m || a != a) ? a : m; }
template <typename T>
__global__ __launch_bounds__(MIK_BLOCK) void k_amax(int64_t n, const T *__restrict__ x, T *__restrict__ part)
{
    __shared__ T sm[MIK_BLOCK];
    T m = T(0);
    for (int64_t i = (int64_t)blockIdx.x * MIK_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MIK_BLOCK) {
        const T v = x[i];
        m = amax_fold(m, v < T(0) ? -v : v);
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int off = MIK_BLOCK / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] = amax_fold(sm[threadIdx.x], sm[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

// fused CG update: x .+= alpha .* u; r .-= alpha .* c; partial sums of r.^2
//   -- src/cg.jl:58-59,62 (and :93-96)
template <typename T> struct OpCgUpdate {
    static constexpr bool REDUCE = true;
    T *__restrict__ x; T *__restrict__ r; const T *__restrict__ u; const T *__restrict__ c; Coef<T> alpha;
    int nt = 0;   // bit 0: x streamed (load + store), bit 1: c streamed, bit 2: u streamed
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        const T a = alpha.get();
        T t = a * u[i]; x[i] = x[i] + t;
        T s = a * c[i]; T rn = r[i] - s; r[i] = rn;
        T p = rn * rn; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        const T a = alpha.get();
        auto xv = (nt & 1) ? vload_nt<T>(x + i) : vload<T>(x + i);
        auto uv = (nt & 4) ? vload_nt(u + i) : vload(u + i);
        auto rv = (nt & 8) ? vload_nt<T>(r + i) : vload<T>(r + i);
        auto cv = (nt & 2) ? vload_nt(c + i) : vload(c + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) {
            T t = a * el<T>(uv, e); el<T>(xv, e) = el<T>(xv, e) + t;
            T s = a * el<T>(cv, e); el<T>(rv, e) = el<T>(rv, e) - s;
        }
        if (nt & 1) vstore_nt(x + i, xv); else vstore(x + i, xv);
        if (nt & 16) vstore_nt(r + i, rv); else vstore(r + i, rv);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T p = el<T>(rv, e) * el<T>(rv, e); acc = acc + p; }
    }
};

// y .+= alpha .* x (skipped when x is null); partial sums of z .* y, or of y.^2 when z is null
//   -- src/minres.jl:104+107 (Lanczos three-term step + projection) and :109+112 (orthogonalise + norm)
template <typename T> struct OpAxpyDot {
    static constexpr bool REDUCE = true;
    const T *x; T *y; const T *z; Coef<T> alpha;    // z may alias y's storage only as "null = y itself"
    int nt = 0;                                       // bit 0: x is streamed (not needed again soon)
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        T yv = y[i];
        if (x) { T t = alpha.get() * x[i]; yv = yv + t; y[i] = yv; }
        T p = (z ? z[i] : yv) * yv; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        auto yv = vload<T>(y + i);
        typename VT<T>::vec zv;
        if (z) zv = vload(z + i);
        if (x) {
            auto xv = (nt & 1) ? vload_nt(x + i) : vload(x + i);
            const T a = alpha.get();
#pragma unroll
            for (int e = 0; e < VT<T>::W; ++e) { T t = a * el<T>(xv, e); el<T>(yv, e) = el<T>(yv, e) + t; }
            vstore(y + i, yv);
        }
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T p = (z ? el<T>(zv, e) : el<T>(yv, e)) * el<T>(yv, e); acc = acc + p; }
    }
};

// MINRES tail of one iteration in one sweep                             -- src/minres.jl:113, :136-142
//   v_next .*= inv_h3;  w_next .= v_curr;  w_next .+= neg_h1 .* w_curr (if w_curr);  w_next .+= neg_h0 .* w_prev
//   (if w_prev);  w_next .*= inv_h2;  x .+= rhs0 .* w_next
template <typename T> struct OpMinresUpdate {
    static constexpr bool REDUCE = false;
    T *__restrict__ v_next; const T *__restrict__ v_curr; const T *__restrict__ w_curr; const T *__restrict__ w_prev;
    T *__restrict__ w_next; T *__restrict__ x;
    Coef<T> inv_h3, neg_h1, neg_h0, inv_h2, rhs0;     // host values, or left on the device by the iteration's own scalar kernel (mik_minres_step)
    int nt = 0;                                       // bit 0: x streamed, bit 1: w_prev streamed (dead afterwards), bit 2: w_next stored streamed and
                                                      // bit 3: w_curr loaded streamed (both idle until the next tail) -- the Infinity Cache then keeps v_next and
                                                      // v_curr, which the next SpMV (its input and its epilogue vector) reads
    __device__ __forceinline__ void apply(int64_t i, T &) const
    {
        v_next[i] = v_next[i] * inv_h3.get();
        T w = v_curr[i];
        if (w_curr) { T t = neg_h1.get() * w_curr[i]; w = w + t; }
        if (w_prev) { T t = neg_h0.get() * w_prev[i]; w = w + t; }
        w = w * inv_h2.get();
        w_next[i] = w;
        T t = rhs0.get() * w; x[i] = x[i] + t;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        auto vn = vload<T>(v_next + i); auto w = vload(v_curr + i);
        auto xv = (nt & 1) ? vload_nt<T>(x + i) : vload<T>(x + i);
        typename VT<T>::vec wc, wp;
        if (w_curr) wc = (nt & 8) ? vload_nt(w_curr + i) : vload(w_curr + i);
        if (w_prev) wp = (nt & 2) ? vload_nt(w_prev + i) : vload(w_prev + i);
        const T c3 = inv_h3.get(), c1 = w_curr ? neg_h1.get() : T(0), c0 = w_prev ? neg_h0.get() : T(0), c2 = inv_h2.get(), cr = rhs0.get();
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) {
            el<T>(vn, e) = el<T>(vn, e) * c3;
            T we = el<T>(w, e);
            if (w_curr) { T t = c1 * el<T>(wc, e); we = we + t; }
            if (w_prev) { T t = c0 * el<T>(wp, e); we = we + t; }
            we = we * c2;
            el<T>(w, e) = we;
            T t = cr * we; el<T>(xv, e) = el<T>(xv, e) + t;
        }
        vstore(v_next + i, vn);
        if (nt & 4) vstore_nt(w_next + i, w); else vstore(w_next + i, w);
        if (nt & 1) vstore_nt(x + i, xv); else vstore(x + i, xv);
    }
};

// Chebyshev search direction: c = Pl \ r (Identity or diagonal); u .= c (first) or u .= c .+ beta .* c
//   -- src/chebyshev.jl:35-45 as written (c itself is overwritten by the SpMV that follows, so it is not stored)
template <typename T> struct OpChebDirection {
    static constexpr bool REDUCE = false;
    const T *__restrict__ r; const T *__restrict__ d; T *__restrict__ u; T beta; int first;
    __device__ __forceinline__ void apply(int64_t i, T &) const
    {
        T c = d ? r[i] / d[i] : r[i];
        if (!first) { T t = beta * c; c = c + t; }
        u[i] = c;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        auto cv = vload(r + i);
        if (d) {
            auto dv = vload(d + i);
#pragma unroll
            for (int e = 0; e < VT<T>::W; ++e) el<T>(cv, e) = el<T>(cv, e) / el<T>(dv, e);
        }
        if (!first) {
#pragma unroll
            for (int e = 0; e < VT<T>::W; ++e) { T t = beta * el<T>(cv, e); el<T>(cv, e) = el<T>(cv, e) + t; }
        }
        vstore(u + i, cv);
    }
};

// PCG preconditioner application: c .= r ./ d; partial sums of c .* r   -- src/cg.jl:79,82
template <typename T> struct OpJacobiDot {
    static constexpr bool REDUCE = true;
    const T *__restrict__ r; const T *__restrict__ d; T *__restrict__ c;
    int nt = 0;    // 1: the diagonal (read once per iteration) is streamed non-temporally
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        T v = r[i] / d[i]; c[i] = v;
        T p = v * r[i]; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        auto rv = vload(r + i); auto dv = nt ? vload_nt(d + i) : vload(d + i); typename VT<T>::vec cv;
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) el<T>(cv, e) = el<T>(rv, e) / el<T>(dv, e);
        vstore(c + i, cv);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T p = el<T>(cv, e) * el<T>(rv, e); acc = acc + p; }
    }
};

// Modified Gram-Schmidt pass: w .-= h .* v; partial sums of z .* w (z = next column, or w itself
// for the closing norm)           -- src/orthogonalize.jl:71-72,75
template <typename T, bool SELF> struct OpMgsPass {
    static constexpr bool REDUCE = true;
    T *__restrict__ w; const T *__restrict__ v; const T *__restrict__ z; Coef<T> h;
    // cache hints (bits; results never depend on them): 1 = v (subtracted here, not needed again in this orthogonalisation) streamed
    // non-temporally, 2 = z streamed, 4 = w loaded non-temporally, 8 = w stored non-temporally
    int nt = 0;
    __device__ __forceinline__ void set_coef(T c) { h.ptr = nullptr; h.val = c; }
    struct Regs { typename VT<T>::vec wv, vv, zv; };          // load / compute halves of apply_vec (k_map_pro)
    __device__ __forceinline__ void load_vec(int64_t i, Regs &r) const
    {
        r.wv = vload<T>(w + i); r.vv = vload(v + i);
        if (!SELF) r.zv = vload(z + i);
    }
    __device__ __forceinline__ void compute_vec(int64_t i, Regs &r, T &acc) const
    {
        const T hh = h.get();
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T t = hh * el<T>(r.vv, e); el<T>(r.wv, e) = el<T>(r.wv, e) - t; }
        vstore(w + i, r.wv);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) {
            T p = (SELF ? el<T>(r.wv, e) : el<T>(r.zv, e)) * el<T>(r.wv, e);
            acc = acc + p;
        }
    }
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        T t = h.get() * v[i]; T wn = w[i] - t; w[i] = wn;
        T p = (SELF ? wn : z[i]) * wn; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        const T hh = h.get();
        auto wv = (nt & 4) ? vload_nt<T>(w + i) : vload<T>(w + i); auto vv = (nt & 1) ? vload_nt(v + i) : vload(v + i);
        typename VT<T>::vec zv;
        if (!SELF) zv = (nt & 2) ? vload_nt(z + i) : vload(z + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) { T t = hh * el<T>(vv, e); el<T>(wv, e) = el<T>(wv, e) - t; }
        if (nt & 8) vstore_nt(w + i, wv); else vstore(w + i, wv);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) {
            T p = (SELF ? el<T>(wv, e) : el<T>(zv, e)) * el<T>(wv, e);
            acc = acc + p;
        }
    }
};

// =============================================================================================
// Modified Gram-Schmidt as ONE launch (n up to 2048 reduction segments = 2.1 M fp64 / 4.2 M fp32 elements)
// =============================================================================================
// orthogonalize_and_normalize!(V[:, 1:k], w, h, ModifiedGramSchmidt()) -- src/orthogonalize.jl:67-79 -- is a chain of
// k + 1 grid-wide reductions (h_i = dot(v_i, w) needs the w of pass i - 1; then norm(w)).  As k + 2 launches each link
// costs a dependent kernel boundary plus a sweep (3.7-4.3 us at n = 125 k).  Here a workgroup owns G = 1, 2, 4 or 8 consecutive
// reduction segments (at most 256 workgroups: one per CU, all resident), keeps its G*256*W*L elements of w in REGISTERS across all
// passes, and the links are hand-offs through memory:
//   * a workgroup publishes its segment sum of pass i with one write-through (sc1) store into slot P[i][wg];
//   * every workgroup then reads ALL slots of pass i (one per thread, sc1 loads that are served past L1) until none
//     of them holds the "not yet written" pattern any more, and folds them with the same level-2 tree as everywhere else
//     (virtual thread vt = 0 + S[vt]; wave tree per 64; the 16 wave sums left to right) -- so every workgroup obtains
//     bit-identical h_i, and the values equal those of the multi-launch chain.
// No counter, no fence: the slot itself is the flag.  "Not yet written" = all bits set (a NaN no arithmetic produces);
// the slots of the other buffer are re-armed by their owners for the next launch (two buffers alternate, the kernel
// boundary in between orders re-arming against use).  Polling is bounded: a workgroup that never sees a slot filled
// (it cannot happen while all <= 256 workgroups are resident, which one workgroup per CU guarantees) raises `err` in
// the mirror instead of hanging the GPU.  v_{i+1} is loaded before the poll, so its latency hides behind the hand-off.
#ifndef MIK_MGS_SLEEP
#define MIK_MGS_SLEEP 0
#endif
template <typename T> struct MgsBits;
template <> struct MgsBits<double> { using U = unsigned long long; static constexpr U EMPTY = ~0ull; static constexpr U QNAN = 0x7ff8000000000000ull; };
template <> struct MgsBits<float>  { using U = unsigned int;       static constexpr U EMPTY = ~0u;   static constexpr U QNAN = 0x7fc00000u; };
// The bits a slot receives for the value v: a NaN is published as the canonical quiet NaN, so that no payload user data can
// carry (a vector filled with 0xFF bytes sums to the all-ones NaN) is ever mistaken for "not yet written".
template <typename T> __device__ __forceinline__ typename MgsBits<T>::U mgs_slot_bits(T v)
{
    typename MgsBits<T>::U bits;
    __builtin_memcpy(&bits, &v, sizeof(T));
    return v != v ? MgsBits<T>::QNAN : bits;
}

// slot accesses of the hand-off.  XL ("XCD-local", round 4): every participating workgroup runs on ONE XCD (the launch has 8 x as many
// workgroups and only those with blockIdx % 8 == 0 -- the ones the dispatcher deals to the first XCD -- take part), so the slots only
// have to be coherent in that XCD's L2: a workgroup-scope (sc0) store, which is written through the L1 into the L2, and a NON-TEMPORAL
// load, which the L1 does not keep -- every poll is served by the L2.  (An ordinary or sc0 load hits the L1 line the first poll left
// there for good, with or without `buffer_inv sc0` in front of it; an agent-scope load is served past the L2 as well.)  One hop then
// costs 1.1-1.2 us instead of the 2.4-2.8 us of the device-wide hand-off (scripts/micro/xl_hop.hip: the skeleton of this kernel).
// Should a non-temporal load ever be served from a stale line, it can only show "not yet written" once more: the poll goes on.
template <bool XL> __device__ __forceinline__ void mgs_slot_store(unsigned long long *p, unsigned long long bits)
{
    if (XL) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(bits) : "memory");
    else __hip_atomic_store(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool XL> __device__ __forceinline__ void mgs_slot_store(unsigned *p, unsigned bits)
{
    if (XL) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(bits) : "memory");
    else __hip_atomic_store(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool XL> __device__ __forceinline__ unsigned long long mgs_slot_load(const unsigned long long *p)
{
    if (XL) {
        unsigned long long v;
        asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool XL> __device__ __forceinline__ unsigned mgs_slot_load(const unsigned *p)
{
    if (XL) {
        unsigned v;
        asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct MgsMirror {             // host-mapped; h[] follows (restart + 2 scalars of the handle's dtype)
    unsigned long long seq;
    int err, pad;
};

// level 2 of a pass over `ns` slots (ns <= 2048 segment sums, published by the workgroups): the fixed 1024-virtual-thread shape
// of level2_sum evaluated by a 256-thread workgroup -- real thread (wave w, lane l) plays the virtual threads (w + 4 j) * 64 + l,
// j = 0..3; a virtual thread adds its slots vt, vt + 1024 in ascending order from +0; wave tree per 64; the 16 wave sums left to
// right.  Every slot is polled until it no longer holds the "not yet written" pattern (bounded).
template <typename T, bool XL = false>
__device__ __forceinline__ T mgs_grid_sum(const T *__restrict__ slots, int ns, T *lds16, int *err)
{
    using U = typename MgsBits<T>::U;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int jmax = ns <= 256 ? 1 : 4;                    // ns <= 256: virtual threads 256..1023 hold +0 (their wave sums are +0)
    for (int j = 0; j < jmax; ++j) {
        const int vt = (w + 4 * j) * 64 + lane;
        T v = T(0);
        for (int q = vt; q < ns; q += MIK_FIN_THREADS) {
            U bits = MgsBits<T>::EMPTY;
            for (int spin = 0; spin < (1 << 18); ++spin) {
                bits = mgs_slot_load<XL>(reinterpret_cast<const U *>(slots) + q);
                if (bits != MgsBits<T>::EMPTY) break;
                if (MIK_MGS_SLEEP) __builtin_amdgcn_s_sleep(1);
            }
            if (bits == MgsBits<T>::EMPTY) *err = 1;      // timed out: never hang the device
            T val;
            __builtin_memcpy(&val, &bits, sizeof(T));
            v = v + val;                                   // 0 + S[vt] (+ S[vt + 1024]), as level2_sum
        }
        v = wave_tree(v);
        if (lane == 0) lds16[w + 4 * j] = v;
    }
    if (jmax == 1 && t >= 4 && t < 16) lds16[t] = T(0);
    __syncthreads();
    T tot = lds16[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) tot = tot + lds16[q];
    __syncthreads();
    return tot;
}

// G consecutive reduction segments per workgroup (G = 1, 2, 4, 8: n up to 2048 segments = 4.2 M fp32 / 2.1 M fp64 elements with at
// most 256 workgroups -- one per CU, all resident, which the slot hand-off needs): a thread keeps G x L x W elements of w in
// registers; segment sums, slots and trees are those of G = 1, so the bits are those of the multi-launch chain.
// XL (only with G = 1, at most 128 segments and columns of at most ~1.5 MB -- one XCD's share of the fabric has to feed them): see
// mgs_slot_store.  xl_chk[parity] receives the XCC id of the first participant; one that finds another id raises err = 2 and the host
// switches the handle to the all-XCD form for good (the slots of workgroups on different XCDs would never become visible).
// X (row partitions, csrc/mik_mail.h MailSumPass): the grid-wide sum of a pass is this RANK's total; xch.pass() swaps it for the sum over the ranks
// in rank order, inside the launch -- every workgroup obtains the same bits, and they are those of the partition-aware chains.
template <typename T, bool VEC, int G, bool XL = false, typename X = NoExchange>
__global__ __launch_bounds__(MIK_BLOCK, 1) void k_mgs_fused(int64_t n, int k, const T *__restrict__ V, int64_t ldv, T *__restrict__ w,
                                                            T *__restrict__ P /* [2][kmax + 1][stride] */, int kmax, int stride, int nseg, int parity,
                                                            MgsMirror *mirror, unsigned long long seq, unsigned *__restrict__ xl_chk = nullptr, X xch = X())
{
    using U = typename MgsBits<T>::U;
    constexpr int W = VT<T>::W, L = MIK_RED_L;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    __shared__ T lds16[16];
    __shared__ T ldsg[G][4];
    __shared__ int s_err;
    if (XL && (blockIdx.x & 7u)) return;                   // the workgroups of the other seven XCDs
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, s = XL ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (t == 0) {
        s_err = 0;
        if (XL) {
            const unsigned me = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) + 1u;      // HW_REG_XCC_ID[3:0] + 1
            unsigned seen = 0u;
            if (!__hip_atomic_compare_exchange_strong(&xl_chk[parity], &seen, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) && seen != me) s_err = 2;
            if (s == 0) __hip_atomic_store(&xl_chk[parity ^ 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    T *cur = P + (size_t)parity * (size_t)(kmax + 1) * stride;
    T *oth = P + (size_t)(parity ^ 1) * (size_t)(kmax + 1) * stride;
    for (int q = t; q < (kmax + 1) * G; q += MIK_BLOCK)     // re-arm this workgroup's slots of the other buffer
        mgs_slot_store<XL>(reinterpret_cast<U *>(oth + (size_t)(q / G) * stride) + s * G + q % G, MgsBits<T>::EMPTY);
    const int64_t base = (int64_t)s * G * SEG + (int64_t)W * t;
    // this thread's elements of w and of the column in flight: segment g, load l: base + g SEG + l 256 W .. + W - 1, valid where < n
    T wr[G][L][W], zr[G][L][W], vr[G][L][W];
    auto load = [&](const T *__restrict__ p, T(&dst)[G][L][W]) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int64_t i = base + g * SEG + (int64_t)l * MIK_BLOCK * W;
                if (VEC && i + W <= n) {
                    auto v = vload(p + i);
#pragma unroll
                    for (int e = 0; e < W; ++e) dst[g][l][e] = el<T>(v, e);
                } else {
#pragma unroll
                    for (int e = 0; e < W; ++e) dst[g][l][e] = (i + e < n) ? p[i + e] : T(0);
                }
            }
    };
    auto publish = [&](int pass, T(&acc)[G]) {             // block tree per segment: wave tree, then the 4 wave sums left to right
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const T ws = wave_tree(acc[g]);
            if (lane == 0) ldsg[g][wv] = ws;
        }
        __syncthreads();
        if (t < G && s * G + t < nseg) {
            T tot = ldsg[t][0];
            tot = tot + ldsg[t][1]; tot = tot + ldsg[t][2]; tot = tot + ldsg[t][3];
            mgs_slot_store<XL>(reinterpret_cast<U *>(cur + (size_t)pass * stride) + s * G + t, mgs_slot_bits<T>(tot));
        }
        __syncthreads();
    };
    load(w, wr);
    T *hout = reinterpret_cast<T *>(mirror + 1);
    T acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = T(0);
    if (k > 0) {
        load(V, zr);
        // dot(v_1, w)                                                    src/orthogonalize.jl:71 (i = 1)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (base + g * SEG + (int64_t)l * MIK_BLOCK * W + e < n) { T p = zr[g][l][e] * wr[g][l][e]; acc[g] = acc[g] + p; }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (base + g * SEG + (int64_t)l * MIK_BLOCK * W + e < n) { T p = wr[g][l][e] * wr[g][l][e]; acc[g] = acc[g] + p; }
    }
    publish(0, acc);
    for (int i = 0; i < k; ++i) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int e = 0; e < W; ++e) vr[g][l][e] = zr[g][l][e];        // v_i: subtracted in this pass
        const bool last = i + 1 == k;
        if (!last) load(V + (int64_t)(i + 1) * ldv, zr);                   // v_{i+1}: in flight during the hand-off
        const T h = xch.pass(mgs_grid_sum<T, XL>(cur + (size_t)i * stride, nseg, lds16, &s_err), i, s == 0);
        if (s == 0 && t == 0) hout[i] = h;
        // w .-= h[i] .* v_i; then dot(v_{i+1}, w) or norm(w)^2            :72, :71 / :75 -- per 16-byte group: all W
        // elements updated, then their products added in element order (the order of OpMgsPass::compute_vec)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acc[g] = T(0);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int64_t i0 = base + g * SEG + (int64_t)l * MIK_BLOCK * W;
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i0 + e < n) { T tt = h * vr[g][l][e]; wr[g][l][e] = wr[g][l][e] - tt; }
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i0 + e < n) { T p = (last ? wr[g][l][e] : zr[g][l][e]) * wr[g][l][e]; acc[g] = acc[g] + p; }
            }
        }
        publish(i + 1, acc);
    }
    const T ss = xch.pass(mgs_grid_sum<T, XL>(cur + (size_t)k * stride, nseg, lds16, &s_err), k, s == 0);
    T nrm = mik_sqrt(ss);
    const bool ok = mik_nrm_in_range(ss);          // outside the safe range: leave w unscaled, the host rescales
    const T inv = ok ? T(1) / nrm : T(1);
    if (!ok) nrm = __builtin_nan("");
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int l = 0; l < L; ++l) {              // w .*= inv(nrm)                                :76
            const int64_t i0 = base + g * SEG + (int64_t)l * MIK_BLOCK * W;
            if (VEC && i0 + W <= n) {
                typename VT<T>::vec o;
#pragma unroll
                for (int e = 0; e < W; ++e) el<T>(o, e) = wr[g][l][e] * inv;
                vstore(w + i0, o);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i0 + e < n) w[i0 + e] = wr[g][l][e] * inv;
            }
        }
    if (t == 0 && s_err) __hip_atomic_store(&mirror->err, s_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (s == 0 && t == 0) {
        hout[k] = nrm;
        __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Classical Gram-Schmidt (and DGKS) as ONE launch (same sizes, same hand-off mechanism as k_mgs_fused):
//   h = V' w  (all k dots on the SAME w: one batch);  w -= V h;  nrm = norm(w);  w *= inv(nrm)      src/orthogonalize.jl:15-17,75-76
// Three dependent grid-wide steps instead of k + 1:  (1) every workgroup publishes the k sums of each of its segments (rows
// 0..k-1 of the slot buffer);  (2) column j is reduced by ONE workgroup (j mod m) with the usual level-2 tree and published as a
// final value (row kmax + 1), which every workgroup then picks up -- k x nseg slot reads per workgroup would cost more than the
// second hand-off;  (3) the norm goes through row k like the last pass of k_mgs_fused.  Products, per-thread order, block and
// level-2 trees are those of k_multidot / k_gemv_n / OpDot + k_finalize_*: bit-identical to the multi-launch chain.
// DGKS (src/orthogonalize.jl:20-36): the same round is repeated while nrm < eta * norm(correction), every round in its own
// block of slot rows; all workgroups evaluate the condition on identical values.  After `rounds` rounds (or when the sum of
// squares leaves the safe range) the kernel stops WITHOUT scaling w and reports {h, nrm, projection size, more = 1}: the host
// continues the loop with the multi-launch chain (it "typically runs once", ibid.).
// X (row partitions, MailSumPass): the reducer workgroup of column j swaps the rank's total for the sum over the ranks before it publishes the final
// value (vector slot round (kmax + 1) + j); the norm is exchanged by every workgroup like a pass of k_mgs_fused (slot round (kmax + 1) + k).
template <typename T, bool VEC, bool DGKS, int G, typename X = NoExchange>
__global__ __launch_bounds__(MIK_BLOCK, 1) void k_cgs_fused(int64_t n, int k, const T *__restrict__ V, int64_t ldv, T *__restrict__ w,
                                                            T *__restrict__ P /* [2][rounds][kmax + 2][stride] */, int kmax, int stride, int nseg, int rounds,
                                                            int parity, MgsMirror *mirror, unsigned long long seq, X xch = X())
{
    using U = typename MgsBits<T>::U;
    constexpr int W = VT<T>::W, L = MIK_RED_L;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    __shared__ T lds16[16];
    __shared__ T ldsg[G][4];
    __shared__ T wsum[256][G][4];                      // wave sums of the k column dots per segment; then this round's h / correction in wsum[j][0][0]
    __shared__ T hacc[256];                            // h, summed over the rounds
    __shared__ T s_proj;
    __shared__ int s_err;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, s = blockIdx.x, m = gridDim.x;
    if (t == 0) s_err = 0;
    const int rows = kmax + 2;                          // per round: k column rows, the norm row (index k <= kmax), the finals row
    T *cur = P + (size_t)parity * (size_t)rounds * rows * stride;
    T *oth = P + (size_t)(parity ^ 1) * (size_t)rounds * rows * stride;
    for (int q = t; q < rounds * rows * G; q += MIK_BLOCK)  // re-arm the other buffer: this workgroup's slots of every row ...
        __hip_atomic_store(reinterpret_cast<U *>(oth + (size_t)(q / G) * stride) + s * G + q % G, MgsBits<T>::EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s == 0)                                         // ... and all 256 entries of the finals rows
        for (int r = 0; r < rounds; ++r)
            __hip_atomic_store(reinterpret_cast<U *>(oth + ((size_t)r * rows + kmax + 1) * stride) + t, MgsBits<T>::EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t base = (int64_t)s * G * SEG + (int64_t)W * t;
    T wr[G][L][W];
    auto load = [&](const T *__restrict__ p, T(&dst)[G][L][W]) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int64_t i = base + g * SEG + (int64_t)l * MIK_BLOCK * W;
                if (VEC && i + W <= n) {
                    auto v = vload(p + i);
#pragma unroll
                    for (int e = 0; e < W; ++e) dst[g][l][e] = el<T>(v, e);
                } else {
#pragma unroll
                    for (int e = 0; e < W; ++e) dst[g][l][e] = (i + e < n) ? p[i + e] : T(0);
                }
            }
    };
    load(w, wr);
    T *hout = reinterpret_cast<T *>(mirror + 1);
    const T eta = T(1) / mik_sqrt(T(2));                // the constant of src/orthogonalize.jl:20
    T nrm = T(0);
    bool ok = true, more = false;
    for (int round = 0;; ++round) {
        T *rb = cur + (size_t)round * rows * stride;
        T *fin = rb + (size_t)(kmax + 1) * stride;
        // (1) segment sums of V[:, j] .* w, j = 0..k-1                                     k_multidot
        for (int j = 0; j < k; ++j) {
            T vr[G][L][W];
            load(V + (int64_t)j * ldv, vr);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                T acc = T(0);
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int e = 0; e < W; ++e)
                        if (base + g * SEG + (int64_t)l * MIK_BLOCK * W + e < n) { T p = vr[g][l][e] * wr[g][l][e]; acc = acc + p; }
                const T ws = wave_tree(acc);
                if (lane == 0) wsum[j][g][wv] = ws;
            }
        }
        __syncthreads();
        for (int q = t; q < k * G; q += MIK_BLOCK) {
            const int j = q / G, g = q % G;
            if (s * G + g < nseg) {
                T tot_t = wsum[j][g][0];
                tot_t = tot_t + wsum[j][g][1]; tot_t = tot_t + wsum[j][g][2]; tot_t = tot_t + wsum[j][g][3];
                __hip_atomic_store(reinterpret_cast<U *>(rb + (size_t)j * stride) + s * G + g, mgs_slot_bits<T>(tot_t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        // (2) level 2 of column j by workgroup j mod m; everybody picks the finals up          k_finalize_store
        for (int j = s; j < k; j += m) {
            const T h = xch.pass(mgs_grid_sum<T>(rb + (size_t)j * stride, nseg, lds16, &s_err), round * (kmax + 1) + j, true);
            if (t == 0)
                __hip_atomic_store(reinterpret_cast<U *>(fin) + j, mgs_slot_bits<T>(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (t < k) {
            U bits = MgsBits<T>::EMPTY;
            for (int spin = 0; spin < (1 << 18); ++spin) {
                bits = __hip_atomic_load(reinterpret_cast<const U *>(fin) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (bits != MgsBits<T>::EMPTY) break;
            }
            if (bits == MgsBits<T>::EMPTY) s_err = 1;
            T hv;
            __builtin_memcpy(&hv, &bits, sizeof(T));
            wsum[t][0][0] = hv;                                                    // this round's h / correction
            hacc[t] = round == 0 ? hv : hacc[t] + hv;                              // h .+= correction               :31
        }
        __syncthreads();
        // w += (-1 * c[j]) * V[:, j], j ascending                                              k_gemv_n
        for (int j = 0; j < k; ++j) {
            T vr[G][L][W];
            load(V + (int64_t)j * ldv, vr);
            const T temp = T(-1) * wsum[j][0][0];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int e = 0; e < W; ++e)
                        if (base + g * SEG + (int64_t)l * MIK_BLOCK * W + e < n) { T p = temp * vr[g][l][e]; wr[g][l][e] = wr[g][l][e] + p; }
        }
        // (3) norm(w)                                                                          OpDot{w, w} + k_finalize_nrm_inv
#pragma unroll
        for (int g = 0; g < G; ++g) {
            T acc = T(0);
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (base + g * SEG + (int64_t)l * MIK_BLOCK * W + e < n) { T p = wr[g][l][e] * wr[g][l][e]; acc = acc + p; }
            const T ws = wave_tree(acc);
            if (lane == 0) ldsg[g][wv] = ws;
        }
        __syncthreads();
        if (t < G && s * G + t < nseg) {
            T tot = ldsg[t][0];
            tot = tot + ldsg[t][1]; tot = tot + ldsg[t][2]; tot = tot + ldsg[t][3];
            __hip_atomic_store(reinterpret_cast<U *>(rb + (size_t)k * stride) + s * G + t, mgs_slot_bits<T>(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const T ss = xch.pass(mgs_grid_sum<T>(rb + (size_t)k * stride, nseg, lds16, &s_err), round * (kmax + 1) + k, s == 0);
        nrm = mik_sqrt(ss);
        ok = mik_nrm_in_range(ss);
        if (!DGKS) break;
        if (t == 0) {                                   // norm(h) / norm(correction): serial, like the host's small_norm    :22, :28
            T q = T(0);
            for (int j = 0; j < k; ++j) { T p = wsum[j][0][0] * wsum[j][0][0]; q = q + p; }
            s_proj = mik_sqrt(q);
        }
        __syncthreads();
        if (!ok) { more = true; break; }                // the host recomputes the norm with scaling and goes on
        if (!(nrm < eta * s_proj)) break;               // :26
        if (round + 1 >= rounds) { more = true; break; }
        __syncthreads();
    }
    // outside the safe range (or DGKS handed back): leave w unscaled, the host goes on
    const T inv = (ok && !more) ? T(1) / nrm : T(1);
    if (!ok) nrm = __builtin_nan("");
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int l = 0; l < L; ++l) {                  // w .*= inv(nrm)
            const int64_t i0 = base + g * SEG + (int64_t)l * MIK_BLOCK * W;
            if (VEC && i0 + W <= n) {
                typename VT<T>::vec o;
#pragma unroll
                for (int e = 0; e < W; ++e) el<T>(o, e) = wr[g][l][e] * inv;
                vstore(w + i0, o);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i0 + e < n) w[i0 + e] = wr[g][l][e] * inv;
            }
        }
    __syncthreads();
    if (t == 0 && s_err) __hip_atomic_store(&mirror->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (s == 0 && t == 0) {                        // one writer for the whole mirror: h, nrm, (projection size), then seq
        for (int j = 0; j < k; ++j) hout[j] = hacc[j];
        hout[k] = nrm;
        if (DGKS) hout[k + 1] = s_proj;
        mirror->pad = more ? 1 : 0;
        __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// =============================================================================================
// level-2 finalise kernels (one 1024-thread workgroup per reduced column)
// =============================================================================================

// out[col] = sum of column col's segment sums
template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_finalize_store(const T *__restrict__ S, int64_t m,
                                                                     int64_t col_stride, T *__restrict__ out,
                                                                     const int *__restrict__ done)
{
    if (done && *done) return;
    __shared__ T lds16[16];
    T tot = level2_sum(S + (int64_t)blockIdx.x * col_stride, m, lds16);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// nrm = sqrt(sum); out[0] = nrm; out[1] = 1 / nrm   -- norm(w); inv(nrm): src/orthogonalize.jl:75-76
template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_finalize_nrm_inv(const T *__restrict__ S, int64_t m,
                                                                       T *__restrict__ out)
{
    __shared__ T lds16[16];
    T tot = level2_sum(S, m, lds16);
    if (threadIdx.x == 0) {
        T nrm = mik_sqrt(tot);
        T inv = T(1) / nrm;
        if (!mik_nrm_in_range(tot)) { nrm = __builtin_nan(""); inv = T(1); }   // host: scaled recomputation (mik_safe_norm_slow)
        out[0] = nrm;
        out[1] = inv;
    }
}

// =============================================================================================
// batched dot (gemv-T) and axpy sweep (gemv-N) over the Krylov basis
// =============================================================================================

// partial sums of V[:, j] .* w for j = 0..k-1, one segment per workgroup; w is read once.
//   -- mul!(h, adjoint(V), w): src/orthogonalize.jl:15,27,43
template <typename T, bool VEC>
__global__ __launch_bounds__(MIK_BLOCK) void k_multidot(int64_t n, int64_t nseg, int k, const T *__restrict__ V,
                                                        int64_t ldv, const T *__restrict__ w,
                                                        T *__restrict__ seg_out /* [k][nseg] */, int nt)
{
    constexpr int W = VT<T>::W;
    constexpr int L = MIK_RED_L;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    __shared__ T lds4[4];
    for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int64_t base = s * SEG + (int64_t)W * threadIdx.x;
        T wr[L * W];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
            if (VEC && i + W <= n) {
                auto wv = vload(w + i);
#pragma unroll
                for (int e = 0; e < W; ++e) wr[l * W + e] = el<T>(wv, e);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e) wr[l * W + e] = (i + e < n) ? w[i + e] : T(0);
            }
        }
        for (int j = 0; j < k; ++j) {
            const T *__restrict__ col = V + (int64_t)j * ldv;
            T acc = T(0);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
                if (VEC && i + W <= n) {
                    auto cv = nt ? vload_nt(col + i) : vload(col + i);
#pragma unroll
                    for (int e = 0; e < W; ++e) { T p = el<T>(cv, e) * wr[l * W + e]; acc = acc + p; }
                } else {
#pragma unroll
                    for (int e = 0; e < W; ++e)
                        if (i + e < n) { T p = col[i + e] * wr[l * W + e]; acc = acc + p; }
                }
            }
            T tot = block_tree_256(acc, lds4);
            if (threadIdx.x == 0) seg_out[(int64_t)j * nseg + s] = tot;
        }
    }
}

// All pairwise dots of K columns in ONE pass: seg_out[p][s], p = index of the pair (r <= c) in row-major order of
// the upper triangle.  Each pair uses the same per-thread order, wave trees and 4-wave-sum order as k_multidot /
// OpDot, so M[r][c] equals dot(V[:, r], V[:, c]) of the column-by-column path bit for bit.
//   -- M = rs' * rs: src/bicgstabl.jl:120 (the column-by-column form reads every column K + 1 times)
template <typename T, bool VEC, int K>
__global__ __launch_bounds__(MIK_BLOCK) void k_gram(int64_t n, int64_t nseg, const T *__restrict__ V, int64_t ldv,
                                                    T *__restrict__ seg_out)
{
    constexpr int W = VT<T>::W;
    constexpr int L = MIK_RED_L;
    constexpr int NP = K * (K + 1) / 2;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    __shared__ T lds[NP][4];
    for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int64_t base = s * SEG + (int64_t)W * threadIdx.x;
        T cr[K][L * W];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const T *__restrict__ col = V + (int64_t)j * ldv;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
                if (VEC && i + W <= n) {
                    auto cv = vload(col + i);
#pragma unroll
                    for (int e = 0; e < W; ++e) cr[j][l * W + e] = el<T>(cv, e);
                } else {
#pragma unroll
                    for (int e = 0; e < W; ++e) cr[j][l * W + e] = (i + e < n) ? col[i + e] : T(0);
                }
            }
        }
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        int p = 0;
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int c = r; c < K; ++c) {
                T acc = T(0);
#pragma unroll
                for (int q = 0; q < L * W; ++q) {
                    const int64_t i = base + (int64_t)(q / W) * MIK_BLOCK * W + (q % W);
                    if (i < n) { T pr = cr[r][q] * cr[c][q]; acc = acc + pr; }
                }
                acc = wave_tree(acc);
                if (lane == 0) lds[p][w] = acc;
                ++p;
            }
        __syncthreads();
        if (threadIdx.x < NP) {
            T tot = lds[threadIdx.x][0];
            tot = tot + lds[threadIdx.x][1]; tot = tot + lds[threadIdx.x][2]; tot = tot + lds[threadIdx.x][3];
            seg_out[(int64_t)threadIdx.x * nseg + s] = tot;
        }
        __syncthreads();
    }
}

// BiCGStab(l) minimal-residual update in one sweep                      -- src/bicgstabl.jl:127-132
//   us_0 -= sum_{j=1..l} gamma_j us_j;  x += sum_{j=0..l-1} gamma_{j+1} rs_j;  rs_0 -= sum_{j=1..l} gamma_j rs_j;
//   partial sums of rs_0.^2.  Same per-element operations and order as the three mul!(y, V, c, alpha, 1) calls
//   (k_gemv_n: temp = alpha * c[j]; y = y + temp * V[:, j], j ascending); x uses rs_0 before its update.
//   sh != nullptr: also the segment sums of sh .* rs_0 (the new residual) at seg_out2 -- rho of the NEXT outer iteration's first
//   column (src/bicgstabl.jl:89), the products and the tree of OpDot, without the sweep over two vectors.
template <typename T> struct BicgGamma { T g[8]; };
template <typename T, bool VEC>
__global__ __launch_bounds__(MIK_BLOCK) void k_bicg_mr(int64_t n, int64_t nseg, int l, T *__restrict__ us, int64_t ldu,
                                                       T *__restrict__ rs, int64_t ldr, T *__restrict__ x, BicgGamma<T> gm,
                                                       T *__restrict__ seg_out, const T *__restrict__ gamma_dev,
                                                       const T *__restrict__ sh = nullptr, T *__restrict__ seg_out2 = nullptr, int nt = 0)
{   // nt (cache hints, results unchanged): 1 = the columns that are dead after this sweep (us_j, rs_j, j >= 1) and r_shadow streamed, 2 = x streamed both ways, 4 / 8 = the us_0 / rs_0 store streamed
    constexpr int W = VT<T>::W;
    constexpr int L = MIK_RED_L;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    __shared__ T lds4[4];
    if (gamma_dev) {               // gamma left on the device by the step's own LU solve (mik_bicgstab_step)
#pragma unroll
        for (int j = 0; j < 8; ++j) gm.g[j] = gamma_dev[j];
    }
    for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int64_t base = s * SEG + (int64_t)W * threadIdx.x;
        T acc = T(0), acc2 = T(0);
#pragma unroll
        for (int lq = 0; lq < L; ++lq) {
            const int64_t i0 = base + (int64_t)lq * MIK_BLOCK * W;
            const bool full = VEC && i0 + W <= n;
            T u0[W], xx[W], r0[W], rold[W];
            if (full) {
                auto a = vload<T>(us + i0); auto b = (nt & 2) ? vload_nt<T>(x + i0) : vload<T>(x + i0); auto c = vload<T>(rs + i0);
#pragma unroll
                for (int e = 0; e < W; ++e) { u0[e] = el<T>(a, e); xx[e] = el<T>(b, e); r0[e] = el<T>(c, e); }
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    const bool in = i0 + e < n;
                    u0[e] = in ? us[i0 + e] : T(0); xx[e] = in ? x[i0 + e] : T(0); r0[e] = in ? rs[i0 + e] : T(0);
                }
            }
#pragma unroll
            for (int e = 0; e < W; ++e) rold[e] = r0[e];
            {   // x += gamma_1 * rs_0 (the not yet updated residual)
                const T temp = T(1) * gm.g[0];
#pragma unroll
                for (int e = 0; e < W; ++e) { T p = temp * rold[e]; xx[e] = xx[e] + p; }
            }
            for (int j = 1; j <= l; ++j) {
                T uj[W], rj[W];
                if (full) {
                    auto a = (nt & 1) ? vload_nt(us + (int64_t)j * ldu + i0) : vload(us + (int64_t)j * ldu + i0);
                    auto c = (nt & 1) ? vload_nt(rs + (int64_t)j * ldr + i0) : vload(rs + (int64_t)j * ldr + i0);
#pragma unroll
                    for (int e = 0; e < W; ++e) { uj[e] = el<T>(a, e); rj[e] = el<T>(c, e); }
                } else {
#pragma unroll
                    for (int e = 0; e < W; ++e) {
                        const bool in = i0 + e < n;
                        uj[e] = in ? us[(int64_t)j * ldu + i0 + e] : T(0); rj[e] = in ? rs[(int64_t)j * ldr + i0 + e] : T(0);
                    }
                }
                const T tneg = T(-1) * gm.g[j - 1];
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    T p = tneg * uj[e]; u0[e] = u0[e] + p;
                    T q = tneg * rj[e]; r0[e] = r0[e] + q;
                }
                if (j < l) {
                    const T tpos = T(1) * gm.g[j];
#pragma unroll
                    for (int e = 0; e < W; ++e) { T p = tpos * rj[e]; xx[e] = xx[e] + p; }
                }
            }
            if (full) {
                typename VT<T>::vec a, b, c;
#pragma unroll
                for (int e = 0; e < W; ++e) { el<T>(a, e) = u0[e]; el<T>(b, e) = xx[e]; el<T>(c, e) = r0[e]; }
                if (nt & 4) vstore_nt(us + i0, a); else vstore(us + i0, a);
                if (nt & 8) vstore_nt(rs + i0, c); else vstore(rs + i0, c);
                if (nt & 2) vstore_nt(x + i0, b); else vstore(x + i0, b);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i0 + e < n) { us[i0 + e] = u0[e]; x[i0 + e] = xx[e]; rs[i0 + e] = r0[e]; }
            }
#pragma unroll
            for (int e = 0; e < W; ++e)
                if (i0 + e < n) { T p = r0[e] * r0[e]; acc = acc + p; }
            if (sh) {
                if (full) {
                    auto hv = (nt & 1) ? vload_nt(sh + i0) : vload(sh + i0);
#pragma unroll
                    for (int e = 0; e < W; ++e) { T p = el<T>(hv, e) * r0[e]; acc2 = acc2 + p; }
                } else {
#pragma unroll
                    for (int e = 0; e < W; ++e)
                        if (i0 + e < n) { T p = sh[i0 + e] * r0[e]; acc2 = acc2 + p; }
                }
            }
        }
        T tot = block_tree_256(acc, lds4);
        if (threadIdx.x == 0) seg_out[s] = tot;
        if (sh) {
            tot = block_tree_256(acc2, lds4);
            if (threadIdx.x == 0) seg_out2[s] = tot;
        }
    }
}

// y += sum_j (alpha * c[j]) * V[:, j], columns ascending (reference-BLAS dgemv 'N' order)
//   -- mul!(y, V, c, alpha, 1): src/orthogonalize.jl:16,30,44; src/gmres.jl:275
template <typename T, bool VEC>
__global__ __launch_bounds__(MIK_BLOCK) void k_gemv_n(int64_t n, int64_t nseg, int k, const T *__restrict__ V,
                                                      int64_t ldv, const T *__restrict__ cf, T alpha,
                                                      T *__restrict__ y, int nt)
{
    constexpr int W = VT<T>::W;
    constexpr int L = MIK_RED_L;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int64_t base = s * SEG + (int64_t)W * threadIdx.x;
        T yr[L * W];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
            if (VEC && i + W <= n) {
                auto yv = vload<T>(y + i);
#pragma unroll
                for (int e = 0; e < W; ++e) yr[l * W + e] = el<T>(yv, e);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e) yr[l * W + e] = (i + e < n) ? y[i + e] : T(0);
            }
        }
        for (int j = 0; j < k; ++j) {
            const T *__restrict__ col = V + (int64_t)j * ldv;
            const T temp = alpha * cf[j];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
                if (VEC && i + W <= n) {
                    auto cv = nt ? vload_nt(col + i) : vload(col + i);
#pragma unroll
                    for (int e = 0; e < W; ++e) { T p = temp * el<T>(cv, e); yr[l * W + e] = yr[l * W + e] + p; }
                } else {
#pragma unroll
                    for (int e = 0; e < W; ++e)
                        if (i + e < n) { T p = temp * col[i + e]; yr[l * W + e] = yr[l * W + e] + p; }
                }
            }
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int64_t i = base + (int64_t)l * MIK_BLOCK * W;
            if (VEC && i + W <= n) {
                typename VT<T>::vec yv;
#pragma unroll
                for (int e = 0; e < W; ++e) el<T>(yv, e) = yr[l * W + e];
                vstore(y + i, yv);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i + e < n) y[i + e] = yr[l * W + e];
            }
        }
    }
}

#endif  // __HIPCC__
