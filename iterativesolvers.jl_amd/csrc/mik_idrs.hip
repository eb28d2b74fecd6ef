// mik_idrs.hip -- IDR(s): one step of iterate(::IDRSIterable) (src/idrs.jl:164-272) per C call (mik_idrs_step).
//
// The reference walks through a cycle of s + 1 steps.  Steps 1..s build one more pair (U[k], G[k] = A U[k]) of the shadow-space basis,
// bi-orthogonalise it against P[1..k-1] and take the residual one dimension further down; step s + 1 is the polynomial step with its omega.
// Statement by statement the arithmetic below is the reference's (rounded multiply, rounded add, the library's fixed reduction trees), but the
// statements of a step are grouped into the fewest sweeps the data dependences allow:
//
//   :188-202   V, Q, ldiv!(Pl, V), U[k] = Q + omega V     ONE sweep over G[k..s], U[k..s], R (V and Q never exist in memory)
//   :203       G[k] = A U[k]                               the operator's SpMV kernel
//   :207-211   alpha_i = dot(P[i], G[k]) / M[i,i]; G[k] -= alpha_i G[i]; U[k] -= alpha_i U[i]
//                                                          a chain of k - 1 sweeps: sweep i finalises dot i itself (k_map_with, up to 1024 segments),
//                                                          divides by M[i,i], updates both vectors and forms the partial sums of dot i + 1
//   :215-217   M[k..s, k] = P[k..s]' G[k]                  one batched dot (k_multidot: G[k] read once)
//   :221-225   beta = f[k] / M[k,k]; R -= beta G[k]; X += beta U[k]; norm(R)
//                                                          ONE sweep, beta formed on the device from the batched dot's first total
//   :226-235   residual smoothing                          two sweeps (T_s = R_s - R is recomputed, never stored)
//
// so that the host waits ONCE per step (for M[k..s, k] and the norm; once more in step 1 for f = P' R and in step s + 1 for omega's three sums).
// The small triangular solve (:187), f (:237-239) and omega (:70-82) are host scalar work in the element type.
#include "mik_internal.h"
#include "mik_kernels.h"
#include "mik_iter.h"

#include <cmath>
#include <new>
#include <vector>

template <typename T>
int mik_spmv_launch(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done);   // mik_core.hip

namespace {

constexpr int IDRS_MAX_S = 32;        // shadow-space dimension the device path takes (the reference's default is 8)

template <typename T> struct IdrsCoef { T c[IDRS_MAX_S]; };

// a quotient formed on the device by every thread that needs it: num / *den (num a host value or a device scalar)
template <typename T> struct Quot {
    const T *num_ptr; T num_val; const T *den;
    __device__ __forceinline__ T get() const { const T a = num_ptr ? *num_ptr : num_val; return a / *den; }
};

// V = c[1] G[k] + ... ; Q = c[1] U[k] + ...; V = R - V; ldiv!(Pl, V); U[k] = Q + omega V                       -- src/idrs.jl:188-202
template <typename T> struct OpIdrsDir {
    static constexpr bool REDUCE = false;
    const T *G; const T *U; int64_t ldg, ldu; int cnt; IdrsCoef<T> c; const T *r; const T *d; T omega; T *uk;   // G, U: column k; uk == U
    int nt = 0;   // 1: everything that is only read here is streamed (non-temporal), so that U[k] -- the SpMV's input right after -- is what the Infinity Cache keeps
    __device__ __forceinline__ void apply(int64_t i, T &) const
    {
        T v = c.c[0] * G[i], q = c.c[0] * U[i];
        for (int j = 1; j < cnt; ++j) {
            T t = c.c[j] * G[(int64_t)j * ldg + i]; v = v + t;
            T u = c.c[j] * U[(int64_t)j * ldu + i]; q = q + u;
        }
        v = r[i] - v;
        if (d) v = v / d[i];
        T t = omega * v;
        uk[i] = q + t;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &) const
    {
        constexpr int W = VT<T>::W;
        auto gv = nt ? vload_nt(G + i) : vload(G + i); auto uv = nt ? vload_nt(U + i) : vload(U + i);
        T v[W], q[W];
#pragma unroll
        for (int e = 0; e < W; ++e) { v[e] = c.c[0] * el<T>(gv, e); q[e] = c.c[0] * el<T>(uv, e); }
        int j = 1;
        for (; j + 1 < cnt; j += 2) {                 // two columns' loads in flight per trip; the sums keep their order
            auto g0 = nt ? vload_nt(G + (int64_t)j * ldg + i) : vload(G + (int64_t)j * ldg + i);
            auto u0 = nt ? vload_nt(U + (int64_t)j * ldu + i) : vload(U + (int64_t)j * ldu + i);
            auto g1 = nt ? vload_nt(G + (int64_t)(j + 1) * ldg + i) : vload(G + (int64_t)(j + 1) * ldg + i);
            auto u1 = nt ? vload_nt(U + (int64_t)(j + 1) * ldu + i) : vload(U + (int64_t)(j + 1) * ldu + i);
            const T c0 = c.c[j], c1 = c.c[j + 1];
#pragma unroll
            for (int e = 0; e < W; ++e) {
                T t = c0 * el<T>(g0, e); v[e] = v[e] + t;
                T u = c0 * el<T>(u0, e); q[e] = q[e] + u;
                T t1 = c1 * el<T>(g1, e); v[e] = v[e] + t1;
                T u1v = c1 * el<T>(u1, e); q[e] = q[e] + u1v;
            }
        }
        for (; j < cnt; ++j) {
            gv = nt ? vload_nt(G + (int64_t)j * ldg + i) : vload(G + (int64_t)j * ldg + i);
            uv = nt ? vload_nt(U + (int64_t)j * ldu + i) : vload(U + (int64_t)j * ldu + i);
            const T cj = c.c[j];
#pragma unroll
            for (int e = 0; e < W; ++e) {
                T t = cj * el<T>(gv, e); v[e] = v[e] + t;
                T u = cj * el<T>(uv, e); q[e] = q[e] + u;
            }
        }
        auto rv = nt ? vload_nt(r + i) : vload(r + i);
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = el<T>(rv, e) - v[e];
        if (d) {
            auto dv = vload(d + i);
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] = v[e] / el<T>(dv, e);
        }
        typename VT<T>::vec o;
#pragma unroll
        for (int e = 0; e < W; ++e) { T t = omega * v[e]; el<T>(o, e) = q[e] + t; }
        vstore(uk + i, o);
    }
};

// G[k] -= alpha G[i]; U[k] -= alpha U[i]; partial sums of z .* G[k] (z = P[i + 1]: the next alpha's dot; null = none)   -- :207-211
template <typename T> struct OpIdrsBiorth {
    static constexpr bool REDUCE = true;
    T *__restrict__ gk; T *__restrict__ uk; const T *__restrict__ gi; const T *__restrict__ ui; const T *__restrict__ z; Coef<T> alpha;
    int nt = 0;   // 1: G[i], U[i] and P[i + 1] (read once per pass) are streamed: G[k] and U[k], which the next pass reads again, are what the cache keeps
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        const T a = alpha.get();
        T t = a * gi[i]; const T g = gk[i] - t; gk[i] = g;
        T u = a * ui[i]; uk[i] = uk[i] - u;
        if (z) { T p = z[i] * g; acc = acc + p; }
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        constexpr int W = VT<T>::W;
        const T a = alpha.get();
        auto g = vload<T>(gk + i); auto u = vload<T>(uk + i);
        auto gv = nt ? vload_nt(gi + i) : vload(gi + i); auto uv = nt ? vload_nt(ui + i) : vload(ui + i);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            T t = a * el<T>(gv, e); el<T>(g, e) = el<T>(g, e) - t;
            T s = a * el<T>(uv, e); el<T>(u, e) = el<T>(u, e) - s;
        }
        vstore(gk + i, g); vstore(uk + i, u);
        if (z) {
            auto zv = nt ? vload_nt(z + i) : vload(z + i);
#pragma unroll
            for (int e = 0; e < W; ++e) { T p = el<T>(zv, e) * el<T>(g, e); acc = acc + p; }
        }
    }
};

// alpha = dot(P[i], G[k]) / M[i, i] in front of the sweep that uses it (k_map_with)                                       -- :208
template <typename T> struct ProIdrsAlpha {
    T mii;
    __device__ __forceinline__ void operator()(T tot, OpIdrsBiorth<T> &op, bool) const { op.alpha = Coef<T>{nullptr, tot / mii}; }
};

template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_idrs_fin_alpha(const T *__restrict__ S, int64_t m, T mii, T *__restrict__ out)
{
    __shared__ T lds16[16];
    const T tot = level2_sum(S, m, lds16);
    if (threadIdx.x == 0) out[0] = tot / mii;
}

// beta = f[k] / M[k, k]; R -= beta G[k]; X += beta U[k]; partial sums of R.^2                                            -- :221-225
template <typename T> struct OpIdrsUpdate {
    static constexpr bool REDUCE = true;
    T *__restrict__ r; const T *__restrict__ g; T *__restrict__ x; const T *__restrict__ u; Quot<T> beta;
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        const T b = beta.get();
        T t = b * g[i]; const T rv = r[i] - t; r[i] = rv;
        T s = b * u[i]; x[i] = x[i] + s;
        T p = rv * rv; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        constexpr int W = VT<T>::W;
        const T b = beta.get();
        auto rv = vload<T>(r + i); auto gv = vload(g + i); auto xv = vload<T>(x + i); auto uv = vload(u + i);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            T t = b * el<T>(gv, e); el<T>(rv, e) = el<T>(rv, e) - t;
            T s = b * el<T>(uv, e); el<T>(xv, e) = el<T>(xv, e) + s;
        }
        vstore(r + i, rv); vstore(x + i, xv);
#pragma unroll
        for (int e = 0; e < W; ++e) { T p = el<T>(rv, e) * el<T>(rv, e); acc = acc + p; }
    }
};

// T_s = R_s - R (not stored); partial sums of R_s .* T_s and of T_s.^2                                                   -- :227-229, :258-260
template <typename T> struct OpIdrsSmoothDots {
    const T *__restrict__ rs; const T *__restrict__ r;
    __device__ __forceinline__ void apply(int64_t i, T &a1, T &a2) const
    {
        const T t = rs[i] - r[i];
        T p = rs[i] * t; a1 = a1 + p;
        T q = t * t; a2 = a2 + q;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &a1, T &a2) const
    {
        auto sv = vload(rs + i); auto rv = vload(r + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) {
            const T t = el<T>(sv, e) - el<T>(rv, e);
            T p = el<T>(sv, e) * t; a1 = a1 + p;
            T q = t * t; a2 = a2 + q;
        }
    }
};

// gamma = num / den; R_s -= gamma T_s; X_s -= gamma (X_s - X); partial sums of R_s.^2                                    -- :229-234
template <typename T> struct OpIdrsSmoothUpdate {
    static constexpr bool REDUCE = true;
    T *__restrict__ rs; const T *__restrict__ r; T *__restrict__ xs; const T *__restrict__ x; Quot<T> gamma;
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        const T g = gamma.get();
        const T t = rs[i] - r[i];
        T a = g * t; const T rn = rs[i] - a; rs[i] = rn;
        const T d = xs[i] - x[i];
        T b = g * d; xs[i] = xs[i] - b;
        T p = rn * rn; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        constexpr int W = VT<T>::W;
        const T g = gamma.get();
        auto sv = vload<T>(rs + i); auto rv = vload(r + i); auto xsv = vload<T>(xs + i); auto xv = vload(x + i);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const T t = el<T>(sv, e) - el<T>(rv, e);
            T a = g * t; el<T>(sv, e) = el<T>(sv, e) - a;
            const T d = el<T>(xsv, e) - el<T>(xv, e);
            T b = g * d; el<T>(xsv, e) = el<T>(xsv, e) - b;
        }
        vstore(rs + i, sv); vstore(xs + i, xsv);
#pragma unroll
        for (int e = 0; e < W; ++e) { T p = el<T>(sv, e) * el<T>(sv, e); acc = acc + p; }
    }
};

// omega(Q, R): partial sums of Q.^2 and of Q .* R                                                                        -- :72-74
template <typename T> struct OpIdrsOmegaDots {
    const T *__restrict__ q; const T *__restrict__ r;
    __device__ __forceinline__ void apply(int64_t i, T &a1, T &a2) const
    {
        T p = q[i] * q[i]; a1 = a1 + p;
        T s = q[i] * r[i]; a2 = a2 + s;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &a1, T &a2) const
    {
        auto qv = vload(q + i); auto rv = vload(r + i);
#pragma unroll
        for (int e = 0; e < VT<T>::W; ++e) {
            T p = el<T>(qv, e) * el<T>(qv, e); a1 = a1 + p;
            T s = el<T>(qv, e) * el<T>(rv, e); a2 = a2 + s;
        }
    }
};

// X += omega V (V = R before this statement when v is null); R -= omega Q; partial sums of R.^2                          -- :253-256
template <typename T> struct OpIdrsOmegaUpdate {
    static constexpr bool REDUCE = true;
    T *__restrict__ r; const T *__restrict__ q; T *__restrict__ x; const T *__restrict__ v; T omega;
    __device__ __forceinline__ void apply(int64_t i, T &acc) const
    {
        const T ro = r[i], vv = v ? v[i] : ro;
        T t = omega * q[i]; const T rn = ro - t; r[i] = rn;
        T s = omega * vv; x[i] = x[i] + s;
        T p = rn * rn; acc = acc + p;
    }
    __device__ __forceinline__ void apply_vec(int64_t i, T &acc) const
    {
        constexpr int W = VT<T>::W;
        auto rv = vload<T>(r + i); auto qv = vload(q + i); auto xv = vload<T>(x + i);
        auto vv = rv;
        if (v) vv = vload(v + i);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            T t = omega * el<T>(qv, e); el<T>(rv, e) = el<T>(rv, e) - t;
            T s = omega * el<T>(vv, e); el<T>(xv, e) = el<T>(xv, e) + s;
        }
        vstore(r + i, rv); vstore(x + i, xv);
#pragma unroll
        for (int e = 0; e < W; ++e) { T p = el<T>(rv, e) * el<T>(rv, e); acc = acc + p; }
    }
};

template <typename T> struct IdrsHost {
    std::vector<T> M, f;        // M column-major s x s (:142), f (:143)
    T omega = T(1);             // :146
    T normR = T(0);             // norm(R) as the last step left it (omega's norm(s) when there is no smoothing)
};

// device scalars of a step (elements of T): [0] norm, [1] 1 / norm, [2, 2 + s) a batched dot (f or M[k..s, k]), then two sums and alpha
constexpr int IDRS_SLOT_VEC = 2, IDRS_SLOT_SUMS = 2 + IDRS_MAX_S, IDRS_SLOT_ALPHA = 4 + IDRS_MAX_S, IDRS_SLOTS = 8 + IDRS_MAX_S;

}  // namespace

struct mik_idrs {
    mik_ctx *ctx = nullptr;
    const mik_csr *A = nullptr;
    int dtype = MIK_F64, s = 0;
    int64_t n = 0, ldp = 0, ldu = 0, ldg = 0;
    void *x = nullptr, *r = nullptr, *U = nullptr, *G = nullptr, *x_s = nullptr, *r_s = nullptr;
    const void *P = nullptr, *diag = nullptr;
    void *q = nullptr, *v = nullptr;      // Q of step s + 1; V = Pl \ R of step s + 1 (only with a preconditioner)
    void *dev = nullptr;                  // IDRS_SLOTS scalars
    void *host = nullptr;                 // IdrsHost<T>
};

static void idrs_disown(mik_ctx *ctx, void *h)
{
    for (size_t i = 0; i < ctx->owned.size(); ++i)
        if (ctx->owned[i].first == h) { ctx->owned.erase(ctx->owned.begin() + (long)i); return; }
}

extern "C" int mik_idrs_destroy(mik_idrs *it)
{
    if (!it) return MIK_OK;
    idrs_disown(it->ctx, it);
    (void)hipSetDevice(it->ctx->device);
    (void)hipStreamSynchronize(it->ctx->stream);
    if (it->q) (void)hipFree(it->q);
    if (it->v) (void)hipFree(it->v);
    if (it->dev) (void)hipFree(it->dev);
    if (it->host) { if (it->dtype == MIK_F64) delete (IdrsHost<double> *)it->host; else delete (IdrsHost<float> *)it->host; }
    delete it;
    return MIK_OK;
}

extern "C" int mik_idrs_create(mik_ctx *ctx, const mik_csr *A, int s, void *x, void *r, const void *P, int64_t ldp, void *U, int64_t ldu, void *G,
                               int64_t ldg, const void *pl_diag, void *x_s, void *r_s, double normR0, mik_idrs **out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    if (!A || A->ctx != ctx || A->n_rows != A->n_cols) return mik_fail(ctx, MIK_ERR_MISMATCH, "mik_idrs_create: A must be a square operator of this context");
    if (s < 1 || s > IDRS_MAX_S) return mik_fail(ctx, MIK_ERR_NOTIMPL, "mik_idrs_create: s = %d (the device path takes 1 ... %d)", s, IDRS_MAX_S);
    const int64_t n = A->n_rows;
    if (n && (!x || !r || !P || !U || !G)) return mik_fail(ctx, MIK_ERR_INVALID, "mik_idrs_create: NULL vector");
    if (ldp < n || ldu < n || ldg < n) return mik_fail(ctx, MIK_ERR_INVALID, "mik_idrs_create: leading dimension smaller than n");
    if ((x_s == nullptr) != (r_s == nullptr)) return mik_fail(ctx, MIK_ERR_INVALID, "mik_idrs_create: residual smoothing takes both X_s and R_s");
    mik_idrs *it = new (std::nothrow) mik_idrs();
    if (!it) return mik_fail(ctx, MIK_ERR_NOMEM, "mik_idrs_create: host allocation failed");
    it->ctx = ctx; it->A = A; it->dtype = A->dtype; it->s = s; it->n = n; it->x = x; it->r = r; it->P = P; it->ldp = ldp; it->U = U; it->ldu = ldu;
    it->G = G; it->ldg = ldg; it->diag = pl_diag; it->x_s = x_s; it->r_s = r_s;
    (void)hipSetDevice(ctx->device);
    const size_t es = A->dtype == MIK_F64 ? 8 : 4;
    hipError_t e;
    if ((e = hipMalloc(&it->dev, es * IDRS_SLOTS)) != hipSuccess || (e = hipMemset(it->dev, 0, es * IDRS_SLOTS)) != hipSuccess ||
        (e = hipMalloc(&it->q, es * (size_t)std::max<int64_t>(n, 1))) != hipSuccess ||
        (pl_diag && (e = hipMalloc(&it->v, es * (size_t)std::max<int64_t>(n, 1))) != hipSuccess)) {
        const int rc = mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_idrs_create: %s", hipGetErrorString(e));
        mik_idrs_destroy(it);
        return rc;
    }
    if (A->dtype == MIK_F64) {
        auto *h = new (std::nothrow) IdrsHost<double>();
        if (h) { h->M.assign((size_t)s * s, 0.0); h->f.assign((size_t)s, 0.0); for (int i = 0; i < s; ++i) h->M[(size_t)i * s + i] = 1.0; h->normR = normR0; }
        it->host = h;
    } else {
        auto *h = new (std::nothrow) IdrsHost<float>();
        if (h) { h->M.assign((size_t)s * s, 0.0f); h->f.assign((size_t)s, 0.0f); for (int i = 0; i < s; ++i) h->M[(size_t)i * s + i] = 1.0f; h->normR = (float)normR0; }
        it->host = h;
    }
    if (!it->host) { mik_idrs_destroy(it); return mik_fail(ctx, MIK_ERR_NOMEM, "mik_idrs_create: host allocation failed"); }
    ctx->owned.push_back({it, [](void *h) { return mik_idrs_destroy((mik_idrs *)h); }});
    *out = it;
    return MIK_OK;
}

namespace {

template <typename T> static int idrs_multidot(mik_ctx *ctx, int64_t n, int k, const T *V, int64_t ldv, const T *w, T *out_dev, int nt = 0)
{
    const int64_t nseg = mik_nseg<T>(n);
    if (k <= 0) return MIK_OK;
    if (nseg == 0) { MIK_HIP(ctx, hipMemsetAsync(out_dev, 0, sizeof(T) * k, ctx->stream)); return MIK_OK; }
    const int grid = (int)std::min<int64_t>(nseg, mik_max_grid(ctx));
    const bool vec = mik_aligned16(V) && mik_aligned16(w) && (ldv % VT<T>::W == 0);
    if (vec) hipLaunchKernelGGL((k_multidot<T, true>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, k, V, ldv, w, (T *)ctx->partials, nt);
    else hipLaunchKernelGGL((k_multidot<T, false>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, k, V, ldv, w, (T *)ctx->partials, 0);
    MIK_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL((k_finalize_store<T>), dim3(k), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, nseg, nseg, out_dev, (const int *)nullptr);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// level 2 of a norm's segment sums into dev[0] (norm; NaN outside the safe range) and dev[1]
template <typename T> static int idrs_fin_norm(mik_ctx *ctx, int64_t nseg, const T *part, T *dev)
{
    if (nseg == 0) { MIK_HIP(ctx, hipMemsetAsync(dev, 0, sizeof(T), ctx->stream)); return MIK_OK; }
    hipLaunchKernelGGL((k_finalize_nrm_inv<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, part, nseg, dev);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// residual smoothing (:226-235 / :257-266): leaves norm(R_s) in dev[0]
template <typename T> static int idrs_smooth(mik_idrs *it)
{
    mik_ctx *ctx = it->ctx;
    const int64_t n = it->n, nseg = mik_nseg<T>(n);
    T *dev = (T *)it->dev, *part = (T *)ctx->partials;
    T *rs = (T *)it->r_s, *xs = (T *)it->x_s, *r = (T *)it->r, *x = (T *)it->x;
    const bool vec = mik_aligned16(rs) && mik_aligned16(xs) && mik_aligned16(r) && mik_aligned16(x);
    if (nseg == 0) { MIK_HIP(ctx, hipMemsetAsync(dev, 0, sizeof(T), ctx->stream)); return MIK_OK; }
    OpIdrsSmoothDots<T> dots{rs, r};
    MIK_TRY((launch_map2<T>(ctx, n, dots, vec, part, part + nseg, nullptr)));
    hipLaunchKernelGGL((k_finalize_store<T>), dim3(2), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)part, nseg, nseg, dev + IDRS_SLOT_SUMS, (const int *)nullptr);
    MIK_LAUNCH_CHECK(ctx);
    OpIdrsSmoothUpdate<T> up{rs, r, xs, x, Quot<T>{dev + IDRS_SLOT_SUMS, T(0), dev + IDRS_SLOT_SUMS + 1}};
    MIK_TRY((launch_map<T>(ctx, n, up, vec, part, nullptr)));
    return idrs_fin_norm<T>(ctx, nseg, part, dev);
}

template <typename T> static int idrs_step_impl(mik_idrs *it, int step, T *normR_out)
{
    mik_ctx *ctx = it->ctx;
    IdrsHost<T> *h = (IdrsHost<T> *)it->host;
    const int s = it->s;
    const int64_t n = it->n, nseg = mik_nseg<T>(n);
    T *dev = (T *)it->dev;
    T *x = (T *)it->x, *r = (T *)it->r, *U = (T *)it->U, *G = (T *)it->G;
    const T *P = (const T *)it->P, *d = (const T *)it->diag;
    const bool smoothing = it->r_s != nullptr;
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * std::max<size_t>((size_t)std::max<int64_t>(nseg, 1) * (size_t)std::max(s, 2), 2048)));
    T *part = (T *)ctx->partials;
    T out[IDRS_MAX_S + 2];
    T nrm;
    // Cache hints (results never depend on them).  Vectors that cannot share the 256 MB Infinity Cache anyway: what a sweep reads ONCE (G[i], U[i], P[i], R)
    // is streamed non-temporally, so that what the next launch reads again (U[k] before the SpMV; G[k], U[k] between the passes of the
    // bi-orthogonalisation) is what the cache keeps.  Measured on one box, 256^3 fp64, s = 8, per step: 1,186 -> 1,048 us (default layout),
    // 1,421 -> 1,246 us (CSR arrays); same residual bits.  Bit 0: the sweeps, bit 1: the batched dots.
    const int nt = (double)n * sizeof(T) > 96.0e6 ? 3 : 0;
#define MM(i, j) h->M[(size_t)(j) * (size_t)s + (size_t)(i)]
    if (step <= s) {
        const int k = step - 1, cnt = s - k;
        if (k == 0) {                                                                               // f = P' R  :178-182
            MIK_TRY(idrs_multidot<T>(ctx, n, s, P, it->ldp, r, dev + IDRS_SLOT_VEC, (nt >> 1) & 1));
            MIK_TRY(mik_read_scalars<T>(ctx, dev + IDRS_SLOT_VEC, s, h->f.data()));
        }
        IdrsCoef<T> c;                                                                              // c = LowerTriangular(M[k:s,k:s]) \ f[k:s]  :187
        for (int i = 0; i < cnt; ++i) c.c[i] = h->f[(size_t)(k + i)];
        for (int j = k; j < s; ++j) {
            c.c[j - k] = c.c[j - k] / MM(j, j);
            const T cj = c.c[j - k];
            for (int i = j + 1; i < s; ++i) { T t = MM(i, j) * cj; c.c[i - k] = c.c[i - k] - t; }
        }
        T *gk = G + (int64_t)k * it->ldg, *uk = U + (int64_t)k * it->ldu;
        const bool vecb = mik_aligned16(G) && mik_aligned16(U) && (it->ldg % VT<T>::W == 0) && (it->ldu % VT<T>::W == 0);
        {
            OpIdrsDir<T> op{gk, uk, it->ldg, it->ldu, cnt, c, r, d, h->omega, uk, nt & 1};
            const bool vec = vecb && mik_aligned16(r) && (!d || mik_aligned16(d));
            MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)nullptr, nullptr)));                       // :188-202
        }
        MIK_TRY(mik_spmv_launch<T>(ctx, it->A, uk, gk, false, nullptr, nullptr));                   // :203
        if (k > 0 && nseg > 0) {                                                                    // :207-211
            const bool vecp = vecb && mik_aligned16(P) && (it->ldp % VT<T>::W == 0);
            const bool lean = nseg <= 1024;
            OpDot<T> d0{P, gk};
            MIK_TRY((launch_map<T>(ctx, n, d0, vecp, part, nullptr)));
            T *cur = part, *nxt = part + nseg;
            for (int i = 0; i < k; ++i) {
                const T *z = i + 1 < k ? P + (int64_t)(i + 1) * it->ldp : nullptr;
                OpIdrsBiorth<T> op{gk, uk, G + (int64_t)i * it->ldg, U + (int64_t)i * it->ldu, z, coef_ptr<T>(dev + IDRS_SLOT_ALPHA), nt & 1};
                if (lean) {
                    MIK_TRY((launch_map_with<T>(ctx, n, op, ProIdrsAlpha<T>{MM(i, i)}, vecp, (const T *)cur, (int)nseg, nxt)));
                } else {
                    hipLaunchKernelGGL((k_idrs_fin_alpha<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)cur, nseg, MM(i, i), dev + IDRS_SLOT_ALPHA);
                    MIK_LAUNCH_CHECK(ctx);
                    MIK_TRY((launch_map<T>(ctx, n, op, vecp, nxt, nullptr)));
                }
                T *t = cur; cur = nxt; nxt = t;
            }
        }
        MIK_TRY(idrs_multidot<T>(ctx, n, cnt, P + (int64_t)k * it->ldp, it->ldp, gk, dev + IDRS_SLOT_VEC, (nt >> 1) & 1));   // M[k..s, k]  :215-217
        {
            OpIdrsUpdate<T> op{r, gk, x, uk, Quot<T>{nullptr, h->f[(size_t)k], dev + IDRS_SLOT_VEC}};          // :221-225
            const bool vec = vecb && mik_aligned16(r) && mik_aligned16(x);
            MIK_TRY((launch_map<T>(ctx, n, op, vec, part, nullptr)));
            MIK_TRY(idrs_fin_norm<T>(ctx, nseg, part, dev));
        }
        if (smoothing) MIK_TRY(idrs_smooth<T>(it));                                                 // :226-235
        MIK_TRY(mik_read_scalars<T>(ctx, dev, 2 + cnt, out));
        for (int i = 0; i < cnt; ++i) MM(k + i, k) = out[2 + i];
        const T beta = h->f[(size_t)k] / MM(k, k);
        for (int i = k + 1; i < s; ++i) { T t = beta * MM(i, k); h->f[(size_t)i] = h->f[(size_t)i] - t; }   // :237-239
        nrm = out[0];
    } else {                                                                                        // step == s + 1  :242-266
        T *q = (T *)it->q, *v = (T *)it->v;
        if (d) {                                                                                    // V = Pl \ R  :246-249
            OpDivide<T> dv{r, d, v};
            MIK_TRY((launch_map<T>(ctx, n, dv, mik_aligned16(r) && mik_aligned16(d) && mik_aligned16(v), (T *)nullptr, nullptr)));
        }
        MIK_TRY(mik_spmv_launch<T>(ctx, it->A, d ? (const T *)v : (const T *)r, q, false, nullptr, nullptr));   // :251
        T ns = h->normR, nt = T(0), ts = T(0);
        if (nseg > 0) {
            const bool vec = mik_aligned16(q) && mik_aligned16(r);
            if (smoothing) {                                                                        // normR holds norm(R_s): norm(R) again
                OpDot<T> rr{r, r};
                MIK_TRY((launch_map<T>(ctx, n, rr, mik_aligned16(r), part, nullptr)));
                MIK_TRY(idrs_fin_norm<T>(ctx, nseg, part, dev));
            }
            OpIdrsOmegaDots<T> od{q, r};
            MIK_TRY((launch_map2<T>(ctx, n, od, vec, part, part + nseg, nullptr)));
            hipLaunchKernelGGL((k_finalize_store<T>), dim3(2), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)part, nseg, nseg, dev + IDRS_SLOT_SUMS, (const int *)nullptr);
            MIK_LAUNCH_CHECK(ctx);
            T sums[2];
            MIK_TRY(mik_read_scalars<T>(ctx, dev + IDRS_SLOT_SUMS, 2, sums));
            if (smoothing) {
                MIK_TRY(mik_read_scalars<T>(ctx, dev, 1, &ns));
                if (ns != ns) MIK_TRY(mik_safe_norm_slow<T>(ctx, n, r, &ns));
            }
            if (mik_nrm_in_range(sums[0])) nt = std::sqrt(sums[0]);
            else MIK_TRY(mik_safe_norm_slow<T>(ctx, n, q, &nt));
            ts = sums[1];
        } else if (smoothing) ns = T(0);
        {                                                                                           // omega(Q, R)  :70-82
            const double angle = std::sqrt(2.) / 2;
            const T qq = nt * ns;
            T rho = ts / qq;
            rho = std::fabs(rho);
            const T nn = nt * nt;
            T omega = ts / nn;
            if ((double)rho < angle) { T a = omega * (T)angle; omega = a / rho; }
            h->omega = omega;
        }
        {
            OpIdrsOmegaUpdate<T> op{r, q, x, d ? (const T *)v : nullptr, h->omega};                 // :253-256
            const bool vec = mik_aligned16(r) && mik_aligned16(q) && mik_aligned16(x) && (!d || mik_aligned16(v));
            MIK_TRY((launch_map<T>(ctx, n, op, vec, part, nullptr)));
            MIK_TRY(idrs_fin_norm<T>(ctx, nseg, part, dev));
        }
        if (smoothing) MIK_TRY(idrs_smooth<T>(it));                                                 // :257-266
        MIK_TRY(mik_read_scalars<T>(ctx, dev, 1, out));
        nrm = out[0];
    }
#undef MM
    if (nrm != nrm && nseg > 0) MIK_TRY(mik_safe_norm_slow<T>(ctx, n, smoothing ? (const T *)it->r_s : (const T *)r, &nrm));   // outside the range of a plain sum of squares
    if (!smoothing) h->normR = nrm;
    *normR_out = nrm;
    return MIK_OK;
}

}  // namespace

extern "C" int mik_idrs_step(mik_idrs *it, int step, void *normR)
{
    if (!it || !normR || step < 1 || step > it->s + 1) return MIK_ERR_INVALID;
    (void)hipSetDevice(it->ctx->device);
    return it->dtype == MIK_F64 ? idrs_step_impl<double>(it, step, (double *)normR) : idrs_step_impl<float>(it, step, (float *)normR);
}

/* omega and M as the host holds them (tests, and a caller that wants to restart from them): omega: one scalar; M: s x s column-major; f: s */
extern "C" int mik_idrs_state(const mik_idrs *it, void *omega, void *M, void *f)
{
    if (!it) return MIK_ERR_INVALID;
    const size_t s = (size_t)it->s;
    if (it->dtype == MIK_F64) {
        const auto *h = (const IdrsHost<double> *)it->host;
        if (omega) *(double *)omega = h->omega;
        if (M) memcpy(M, h->M.data(), sizeof(double) * s * s);
        if (f) memcpy(f, h->f.data(), sizeof(double) * s);
    } else {
        const auto *h = (const IdrsHost<float> *)it->host;
        if (omega) *(float *)omega = h->omega;
        if (M) memcpy(M, h->M.data(), sizeof(float) * s * s);
        if (f) memcpy(f, h->f.data(), sizeof(float) * s);
    }
    return MIK_OK;
}
