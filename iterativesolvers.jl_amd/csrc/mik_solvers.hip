// mik_solvers.hip -- whole iterations of the widened solvers enqueued by ONE call each, their scalar
//                    recurrences on the device: BiCGStab(l) (mik_bicgstab_step) and MINRES (mik_minres_step).
//
// ---- BiCGStab(l) ----------------------------------------------------------------------------------------------------------
//
// iterate(::BiCGStabIterable) at src/bicgstabl.jl:79-134, statement by statement with the kernels of the L1 entry points
// (mik_dot, mik_xpby, mik_spmv, mik_axpy, mik_gram, mik_bicgstab_mr_update): same sweeps, same reduction tree, same scalar
// arithmetic in the element type -- but rho, beta, sigma, alpha, M, gamma, omega never leave the device, so the host waits
// ONCE per outer iteration (for the residual norm) instead of 2 l + 3 times.  The reference authors' own benchmark of this
// solver is advection_dominated() with n = 125,000 (benchmark/benchmark-linear-systems.jl:68-77): there an outer iteration is
// ~20 launches of a few microseconds each, and the host round trips were most of its time.
#include "mik_internal.h"
#include "mik_kernels.h"
#include "mik_iter.h"

#include <cmath>
#include <limits>
#include <new>

template <typename T>
int mik_spmv_launch(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done);   // mik_core.hip

namespace {

template <typename T> struct BicgDev {
    T sigma, omega, rho, alpha, neg_alpha, neg_beta;
    T gamma[8];
};

struct BicgMirror {                 // pinned, written by the last kernel of a step
    double residual;
    int range;                      // |rs[:, 1]|^2 left the range of a safe sqrt(sum of squares): the host rescales (mik_safe_norm_slow)
    int singular;                   // lu! met an exactly singular pivot (the reference throws SingularException)
    unsigned long long seq;
};

// :89-90 in front of :93 -- what k_bicg_fin_rho does, per workgroup (reads sigma, omega; writes rho: no other workgroup reads it here)
template <typename T> struct ProBicgRho {
    BicgDev<T> *d; int first;
    __device__ __forceinline__ void operator()(T tot, OpBicgU<T> &op, bool publish) const
    {
        T sigma = d->sigma;
        if (first) { const T p = d->omega * sigma; sigma = -p; }
        const T beta = tot / sigma;
        op.neg_beta = Coef<T>{nullptr, -beta};
        if (publish) d->rho = tot;
    }
};
// :100-101 in front of :103, :111 -- k_bicg_fin_sigma per workgroup (reads rho; writes sigma for the next column)
template <typename T> struct ProBicgSigma {
    BicgDev<T> *d;
    __device__ __forceinline__ void operator()(T tot, OpBicgR<T> &op, bool publish) const
    {
        const T alpha = d->rho / tot;
        op.neg_alpha = Coef<T>{nullptr, -alpha};
        op.alpha = Coef<T>{nullptr, alpha};
        if (publish) { d->sigma = tot; d->alpha = alpha; d->neg_alpha = -alpha; }
    }
};

// rho = dot(r_shadow, rs[:, j]) (:89); beta = rho / sigma (:90); first: sigma = -omega * sigma before (:85)
template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_bicg_fin_rho(const T *__restrict__ S, int64_t m, BicgDev<T> *d, int first)
{
    __shared__ T lds16[16];
    const T tot = level2_sum(S, m, lds16);
    if (threadIdx.x == 0) {
        if (first) { const T p = d->omega * d->sigma; d->sigma = -p; }
        d->rho = tot;
        const T beta = tot / d->sigma;
        d->neg_beta = -beta;
    }
}

// sigma = dot(r_shadow, us[:, j + 1]) (:100); alpha = rho / sigma (:101)
template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_bicg_fin_sigma(const T *__restrict__ S, int64_t m, BicgDev<T> *d)
{
    __shared__ T lds16[16];
    const T tot = level2_sum(S, m, lds16);
    if (threadIdx.x == 0) {
        d->sigma = tot;
        const T alpha = d->rho / tot;
        d->alpha = alpha;
        d->neg_alpha = -alpha;
    }
}

// the same two over one partial per 256-row block (rho / sigma formed in the SpMV launch: 65,536 partials at 256^3): 16 single-wave
// workgroups and a ticket, as k_cg_fin_alpha
template <typename T>
__global__ __launch_bounds__(64) void k_bicg_fin_rho_spread(const T *__restrict__ S, int64_t m, BicgDev<T> *d, int first, FinScratch<T> *fs)
{
    T tot;
    if (level2_sum_spread(S, m, fs, tot)) {
        if (first) { const T p = d->omega * d->sigma; d->sigma = -p; }
        d->rho = tot;
        const T beta = tot / d->sigma;
        d->neg_beta = -beta;
    }
}
template <typename T>
__global__ __launch_bounds__(64) void k_bicg_fin_sigma_spread(const T *__restrict__ S, int64_t m, BicgDev<T> *d, FinScratch<T> *fs)
{
    T tot;
    if (level2_sum_spread(S, m, fs, tot)) {
        d->sigma = tot;
        const T alpha = d->rho / tot;
        d->alpha = alpha;
        d->neg_alpha = -alpha;
    }
}

// F = lu!(view(M, L, L)); ldiv!(gamma, F, view(M, L, 1)); omega = gamma[l]   (:123-125, :131) -- lu_solve of mik_krylov.hip, on
// the packed upper triangle the Gram finaliser left (row r, columns r..k-1, r ascending)
template <typename T> __device__ void bicg_gamma(const T *__restrict__ packed, int l, BicgDev<T> *d, BicgMirror *mirror)
{
    const int k = l + 1;
    T M[5][5], b[4];
    int p = 0;
    for (int r = 0; r < k; ++r)
        for (int c = r; c < k; ++c) { M[r][c] = packed[p]; M[c][r] = packed[p]; ++p; }
    // A = M[1:, 1:] (column-major in the host version; symmetric here), b = M[1:, 0]
    T A[4][4];
    for (int i = 0; i < l; ++i) {
        b[i] = M[i + 1][0];
        for (int j = 0; j < l; ++j) A[i][j] = M[i + 1][j + 1];
    }
    bool singular = false;
    for (int j = 0; j < l && !singular; ++j) {
        int pv = j;
        T best = fabs(A[j][j]);
        for (int i = j + 1; i < l; ++i) { const T a = fabs(A[i][j]); if (a > best) { best = a; pv = i; } }
        if (best == T(0)) { singular = true; break; }
        if (pv != j) {
            for (int c = 0; c < l; ++c) { const T t = A[j][c]; A[j][c] = A[pv][c]; A[pv][c] = t; }
            const T t = b[j]; b[j] = b[pv]; b[pv] = t;
        }
        const T piv = A[j][j];
        for (int i = j + 1; i < l; ++i) {
            const T mlt = A[i][j] / piv;
            A[i][j] = mlt;
            for (int c = j + 1; c < l; ++c) { const T t = mlt * A[j][c]; A[i][c] = A[i][c] - t; }
            const T t = mlt * b[j]; b[i] = b[i] - t;
        }
    }
    if (singular) { mirror->singular = 1; for (int j = 0; j < 8; ++j) d->gamma[j] = T(0); return; }
    for (int j = l - 1; j >= 0; --j) {
        b[j] = b[j] / A[j][j];
        const T t = b[j];
        for (int i = 0; i < j; ++i) { const T q = t * A[i][j]; b[i] = b[i] - q; }
    }
    for (int j = 0; j < 8; ++j) d->gamma[j] = j < l ? b[j] : T(0);
    d->omega = b[l - 1];
}
template <typename T> __global__ void k_bicg_gamma(const T *__restrict__ packed, int l, BicgDev<T> *d, BicgMirror *mirror)
{
    bicg_gamma<T>(packed, l, d, mirror);
}
// few segments: the np level-2 sums of the Gram matrix and the LU solve in ONE single-workgroup launch
template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_bicg_gram_gamma(const T *__restrict__ S, int64_t nseg, int np, int l, BicgDev<T> *d, BicgMirror *mirror)
{
    __shared__ T lds16[16];
    __shared__ T packed[15];
    for (int c = 0; c < np; ++c) {
        const T tot = level2_sum(S + (int64_t)c * nseg, nseg, lds16);
        if (threadIdx.x == 0) packed[c] = tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) bicg_gamma<T>(packed, l, d, mirror);
}

// residual = norm(rs[:, 1]) (:132) and the publication of the step
template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_bicg_fin_norm(const T *__restrict__ S, int64_t m, BicgMirror *mirror, unsigned long long seq)
{
    __shared__ T lds16[16];
    const T tot = level2_sum(S, m, lds16);
    if (threadIdx.x == 0) {
        if (mik_nrm_in_range(tot) || tot != tot) { mirror->residual = (double)mik_sqrt(tot); mirror->range = 0; }
        else mirror->range = 1;
        __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <typename T, int K> int gram_partials(mik_ctx *ctx, int64_t n, const T *V, int64_t ldv)
{
    const int64_t nseg = mik_nseg<T>(n);
    const int grid = (int)std::min<int64_t>(nseg, mik_max_grid(ctx));
    const bool vec = mik_aligned16(V) && (ldv % VT<T>::W == 0);
    if (vec) hipLaunchKernelGGL((k_gram<T, true, K>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, V, ldv, (T *)ctx->partials);
    else hipLaunchKernelGGL((k_gram<T, false, K>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, V, ldv, (T *)ctx->partials);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

}  // namespace

struct mik_bicgstab {
    mik_ctx *ctx = nullptr;
    const mik_csr *A = nullptr;
    int dtype = MIK_F64, l = 2;
    int64_t n = 0, ldr = 0, ldu = 0;
    void *x = nullptr, *rs = nullptr, *us = nullptr;
    const void *r_shadow = nullptr, *pl_diag = nullptr;
    bool failed = false;                       // a singular MR system was reported: every later step reports it again
    bool fuse = false;              // no Pl, aligned r_shadow: sigma (:100) and rho of j >= 2 (:89) may leave the SpMV launch in front of them (bicg_fuses)
    void *dev = nullptr;            // BicgDev<T>
    void *fin = nullptr;            // FinScratch<T> of the spread finalisers
    void *rho_part = nullptr;       // segment sums of dot(r_shadow, rs[:, 1]) left by the MR sweep of the step before (rho of the first column, :89)
    bool rho_ready = false;
    BicgMirror *mirror = nullptr;
    unsigned long long seq = 0;
};

static void ctx_disown(mik_ctx *ctx, void *h)
{
    for (size_t i = 0; i < ctx->owned.size(); ++i)
        if (ctx->owned[i].first == h) { ctx->owned.erase(ctx->owned.begin() + (long)i); return; }
}

extern "C" int mik_bicgstab_destroy(mik_bicgstab *it)
{
    if (!it) return MIK_OK;
    ctx_disown(it->ctx, it);
    (void)hipSetDevice(it->ctx->device);
    (void)hipStreamSynchronize(it->ctx->stream);
    if (it->dev) (void)hipFree(it->dev);
    if (it->fin) (void)hipFree(it->fin);
    if (it->rho_part) (void)hipFree(it->rho_part);
    if (it->mirror) (void)hipHostFree(it->mirror);
    delete it;
    return MIK_OK;
}

extern "C" int mik_bicgstab_create(mik_ctx *ctx, const mik_csr *A, int l, void *x, void *rs, int64_t ldr, void *us, int64_t ldu,
                                   const void *r_shadow, const void *pl_diag, mik_bicgstab **out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    if (!A || A->ctx != ctx || A->n_rows != A->n_cols) return mik_fail(ctx, MIK_ERR_MISMATCH, "mik_bicgstab_create: A must be a square operator of this context");
    if (l < 1 || l > 4) return mik_fail(ctx, MIK_ERR_NOTIMPL, "mik_bicgstab_create: l = %d (1 ... 4; drive larger l through the L1 entry points)", l);
    const int64_t n = A->n_rows;
    if (n && (!x || !rs || !us || !r_shadow || ldr < n || ldu < n)) return mik_fail(ctx, MIK_ERR_INVALID, "mik_bicgstab_create: NULL vector or leading dimension < n");
    mik_bicgstab *it = new (std::nothrow) mik_bicgstab();
    if (!it) return mik_fail(ctx, MIK_ERR_NOMEM, "mik_bicgstab_create: host allocation failed");
    it->ctx = ctx; it->A = A; it->dtype = A->dtype; it->l = l; it->n = n; it->ldr = ldr; it->ldu = ldu;
    it->x = x; it->rs = rs; it->us = us; it->r_shadow = r_shadow; it->pl_diag = pl_diag;
    (void)hipSetDevice(ctx->device);
    hipError_t e;
    const size_t db = A->dtype == MIK_F64 ? sizeof(BicgDev<double>) : sizeof(BicgDev<float>);
    it->fuse = !pl_diag && mik_aligned16(r_shadow);
    const size_t rb = (A->dtype == MIK_F64 ? sizeof(double) * (size_t)mik_nseg<double>(n) : sizeof(float) * (size_t)mik_nseg<float>(n)) + 16;
    if ((e = hipMalloc(&it->dev, db)) != hipSuccess || (e = hipMalloc(&it->fin, 512)) != hipSuccess || (e = hipMemset(it->fin, 0, 512)) != hipSuccess ||
        (e = hipMalloc(&it->rho_part, rb)) != hipSuccess ||
        (e = hipHostMalloc((void **)&it->mirror, sizeof(BicgMirror), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) {
        const int rc = mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_bicgstab_create: %s", hipGetErrorString(e));
        mik_bicgstab_destroy(it);
        return rc;
    }
    memset(it->mirror, 0, sizeof(BicgMirror));
    // omega = sigma = 1 (src/bicgstabl.jl:59)
    if (A->dtype == MIK_F64) { BicgDev<double> h{}; h.sigma = 1.0; h.omega = 1.0; e = hipMemcpy(it->dev, &h, sizeof(h), hipMemcpyHostToDevice); }
    else { BicgDev<float> h{}; h.sigma = 1.0f; h.omega = 1.0f; e = hipMemcpy(it->dev, &h, sizeof(h), hipMemcpyHostToDevice); }
    if (e != hipSuccess) {
        const int rc = mik_fail(ctx, MIK_ERR_HIP, "mik_bicgstab_create: %s", hipGetErrorString(e));
        mik_bicgstab_destroy(it);
        return rc;
    }
    ctx->owned.push_back({it, [](void *h) { return mik_bicgstab_destroy((mik_bicgstab *)h); }});
    *out = it;
    return MIK_OK;
}

// Do the steps of this handle form sigma and rho in the SpMV launches?  (The sweeps that finalise their producer's reduction themselves
// at launch-bound sizes take at most 1024 partials.)
template <typename T> static bool bicg_fuses(const mik_bicgstab *it)
{
    if (!it->fuse || !mik_spmv_has_epilogue(it->A)) return false;     // (MIK_KNOB_SOLVER_FORM = 2: no epilogues)
    const int64_t nseg = mik_nseg<T>(it->n), nb = mik_spmv_nwg(it->n);
    const bool lean = nseg <= 1024 && it->ctx->tuning[MIK_KNOB_SOLVER_FORM] == 0;
    return !lean || nb <= 1024;
}

template <typename T> static int bicg_step_impl(mik_bicgstab *it, T *residual)
{
    mik_ctx *ctx = it->ctx;
    const int64_t n = it->n, nseg = mik_nseg<T>(n);
    const int l = it->l;
    BicgDev<T> *d = (BicgDev<T> *)it->dev;
    T *x = (T *)it->x, *rs = (T *)it->rs, *us = (T *)it->us;
    const T *sh = (const T *)it->r_shadow;
    if (nseg == 0) { *residual = T(0); return MIK_OK; }
    const int np = (l + 1) * (l + 2) / 2;
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)nseg * (size_t)np));
    auto col = [](T *base, int64_t ld, int j) { return base + (int64_t)j * ld; };
    auto dot_partials = [&](const T *a, const T *b) {
        OpDot<T> op{a, b};
        return launch_map<T>(ctx, n, op, mik_aligned16(a) && mik_aligned16(b), (T *)ctx->partials, nullptr);
    };
    auto ldiv = [&](T *v) { return it->pl_diag ? mik_divide(ctx, it->dtype, n, v, it->pl_diag, v) : MIK_OK; };
    const bool blockvec = mik_aligned16(us) && mik_aligned16(rs) && mik_aligned16(x) && (it->ldu % VT<T>::W == 0) && (it->ldr % VT<T>::W == 0);
    const bool lean = nseg <= 1024 && ctx->tuning[MIK_KNOB_SOLVER_FORM] == 0;   // the sweeps finalise their producers' reductions themselves (k_map_with; development knob 25 = 1: separate finaliser launches)
    const int bnt = 7;   // the block sweeps stream everything -- also the column the SpMV behind them reads: with 3 to 9 columns in flight the Infinity Cache keeps too little of it to matter (mask 3, that store cached: 1,174-1,177 us; 7: 1,160-1,169; loads only: 1,217; development knob 7 < 0: all cached; bits 4-6: explicit mask)
    // sigma = dot(r_shadow, A u) (:100) and, from the second column on, rho = dot(r_shadow, A r) (:89) leave the SpMV launch that forms the
    // vector (epilogue dot(z, y), one partial per 256-row block: mik_bicgstab_dot_shape) -- a sweep over two vectors and a launch less each
    const int64_t nb = mik_spmv_nwg(n);
    const bool fuse = bicg_fuses<T>(it);
    if (fuse) MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(nb, nseg * np)));
    auto spmv_dot = [&](const T *in, T *out) {                                               // out = A in; partials of dot(r_shadow, out)
        ctx->spmv_ep_z = sh;
        const int rc = mik_spmv_launch<T>(ctx, it->A, in, out, true, (T *)ctx->partials, nullptr);
        ctx->spmv_ep_z = nullptr;
        return rc;
    };
    for (int j = 0; j < l; ++j) {                                                            // BiCG part  :88
        const bool rho_here = fuse && j > 0;                                                 // its partials came out of :107 of the column before
        const bool rho_kept = j == 0 && it->rho_ready;                                       // ... out of the MR sweep of the step before (k_bicg_mr)
        const int64_t mr = rho_here ? nb : nseg;
        const T *rpart = rho_kept ? (const T *)it->rho_part : (const T *)ctx->partials;
        if (!rho_here && !rho_kept) MIK_TRY(dot_partials(sh, col(rs, it->ldr, j)));          // :89
        if (lean) {                                                                          // :90, :93 -- us = rs - beta * us, all j + 1 columns
            OpBicgU<T> op{us, it->ldu, rs, it->ldr, j + 1, Coef<T>{nullptr, T(0)}, bnt};
            MIK_TRY((launch_map_with<T>(ctx, n, op, ProBicgRho<T>{d, j == 0 ? 1 : 0}, blockvec, rpart, (int)mr, (T *)nullptr)));
        } else {
            if (mr > 16384) hipLaunchKernelGGL((k_bicg_fin_rho_spread<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, rpart, mr, d, j == 0 ? 1 : 0, (FinScratch<T> *)it->fin);
            else hipLaunchKernelGGL((k_bicg_fin_rho<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, rpart, mr, d, j == 0 ? 1 : 0);
            MIK_LAUNCH_CHECK(ctx);
            OpBicgU<T> op{us, it->ldu, rs, it->ldr, j + 1, coef_ptr<T>(&d->neg_beta), bnt};
            MIK_TRY((launch_map<T>(ctx, n, op, blockvec, (T *)nullptr, nullptr)));
        }
        const int64_t ms = fuse ? nb : nseg;
        if (fuse) MIK_TRY(spmv_dot(col(us, it->ldu, j), col(us, it->ldu, j + 1)));           // :97 + :100
        else {
            MIK_TRY(mik_spmv_launch<T>(ctx, it->A, col(us, it->ldu, j), col(us, it->ldu, j + 1), false, nullptr, nullptr));   // :97
            MIK_TRY(ldiv(col(us, it->ldu, j + 1)));                                          // :98
            MIK_TRY(dot_partials(sh, col(us, it->ldu, j + 1)));                              // :100
        }
        if (lean) {                                                                          // :101, :103, :111 (x does not depend on :107)
            OpBicgR<T> op{us, it->ldu, rs, it->ldr, j + 1, x, Coef<T>{nullptr, T(0)}, Coef<T>{nullptr, T(0)}, bnt};
            MIK_TRY((launch_map_with<T>(ctx, n, op, ProBicgSigma<T>{d}, blockvec, (const T *)ctx->partials, (int)ms, (T *)nullptr)));
        } else {
            if (ms > 16384) hipLaunchKernelGGL((k_bicg_fin_sigma_spread<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)ctx->partials, ms, d, (FinScratch<T> *)it->fin);
            else hipLaunchKernelGGL((k_bicg_fin_sigma<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, ms, d);
            MIK_LAUNCH_CHECK(ctx);
            OpBicgR<T> op{us, it->ldu, rs, it->ldr, j + 1, x, coef_ptr<T>(&d->neg_alpha), coef_ptr<T>(&d->alpha), bnt};
            MIK_TRY((launch_map<T>(ctx, n, op, blockvec, (T *)nullptr, nullptr)));
        }
        if (fuse && j + 1 < l) MIK_TRY(spmv_dot(col(rs, it->ldr, j), col(rs, it->ldr, j + 1)));   // :107 + :89 of the next column
        else {
            MIK_TRY(mik_spmv_launch<T>(ctx, it->A, col(rs, it->ldr, j), col(rs, it->ldr, j + 1), false, nullptr, nullptr));   // :107
            MIK_TRY(ldiv(col(rs, it->ldr, j + 1)));                                          // :108
        }
    }
    // MR part: M = rs' * rs (:120) in one pass, gamma (:123-125), the three updates and the norm (:127-132) in one sweep
    switch (l + 1) {
    case 2: MIK_TRY((gram_partials<T, 2>(ctx, n, rs, it->ldr))); break;
    case 3: MIK_TRY((gram_partials<T, 3>(ctx, n, rs, it->ldr))); break;
    case 4: MIK_TRY((gram_partials<T, 4>(ctx, n, rs, it->ldr))); break;
    default: MIK_TRY((gram_partials<T, 5>(ctx, n, rs, it->ldr))); break;
    }
    if (lean) {
        hipLaunchKernelGGL((k_bicg_gram_gamma<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, nseg, np, l, d, it->mirror);
        MIK_LAUNCH_CHECK(ctx);
    } else {
        T *packed = (T *)ctx->coef;
        hipLaunchKernelGGL((k_finalize_store<T>), dim3(np), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, nseg, nseg, packed, (const int *)nullptr);
        MIK_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL((k_bicg_gamma<T>), dim3(1), dim3(1), 0, ctx->stream, (const T *)packed, l, d, it->mirror);
        MIK_LAUNCH_CHECK(ctx);
    }
    {
        const int grid = (int)std::min<int64_t>(nseg, mik_max_grid(ctx));
        const bool vec = mik_aligned16(us) && mik_aligned16(rs) && mik_aligned16(x) && (it->ldu % VT<T>::W == 0) && (it->ldr % VT<T>::W == 0);
        BicgGamma<T> gm{};
        // (+ the segment sums of dot(r_shadow, new residual): rho of the next step's first column, while the residual is in registers)
        const bool keep = ctx->tuning[MIK_KNOB_SOLVER_FORM] != 2 && (!vec || mik_aligned16(sh));
        const int mrnt = 3;      // hint mask of the MR sweep
        if (vec) hipLaunchKernelGGL((k_bicg_mr<T, true>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, l, us, it->ldu, rs, it->ldr, x, gm, (T *)ctx->partials, (const T *)d->gamma,
                                    keep ? sh : (const T *)nullptr, (T *)it->rho_part, mrnt);
        else hipLaunchKernelGGL((k_bicg_mr<T, false>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, l, us, it->ldu, rs, it->ldr, x, gm, (T *)ctx->partials, (const T *)d->gamma,
                                keep ? sh : (const T *)nullptr, (T *)it->rho_part);
        MIK_LAUNCH_CHECK(ctx);
        it->rho_ready = keep;
    }
    it->seq += 1;
    hipLaunchKernelGGL((k_bicg_fin_norm<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, nseg, it->mirror, it->seq);
    MIK_LAUNCH_CHECK(ctx);
    // the one host wait of the outer iteration
    for (unsigned long long spins = 0;; ++spins) {
        if (__atomic_load_n(&it->mirror->seq, __ATOMIC_ACQUIRE) == it->seq) break;
        if ((spins & 0xFFFFF) == 0xFFFFF) {
            const hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) {
                if (__atomic_load_n(&it->mirror->seq, __ATOMIC_ACQUIRE) == it->seq) break;
                return mik_fail(ctx, MIK_ERR_HIP, "bicgstabl: stream idle but step %llu was never published", it->seq);
            }
            if (e != hipErrorNotReady) return mik_fail(ctx, MIK_ERR_HIP, "bicgstabl: %s while waiting for a step", hipGetErrorString(e));
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (it->mirror->singular) {
        // the device state is not steppable any more (gamma = 0, omega stale): the handle stays failed, like the reference's iterable
        // after the SingularException of lu! (src/bicgstabl.jl:124)
        it->failed = true;
        return mik_fail(ctx, MIK_ERR_SINGULAR, "bicgstabl: lu! of the %d x %d MR system met an exactly singular pivot", l, l);
    }
    if (it->mirror->range) return mik_safe_norm_slow<T>(ctx, n, rs, residual);               // norm(rs[:, 1]) of a badly scaled residual
    *residual = (T)it->mirror->residual;
    return MIK_OK;
}

extern "C" int mik_bicgstab_dot_shape(const mik_bicgstab *it, int *W, int *L)
{
    if (!it) return MIK_ERR_INVALID;
    if (it->dtype == MIK_F64 ? bicg_fuses<double>(it) : bicg_fuses<float>(it)) return mik_spmv_dot_shape(W, L);
    return mik_reduce_shape(it->dtype, W, L);
}

extern "C" int mik_bicgstab_step(mik_bicgstab *it, void *residual)
{
    if (!it || !residual) return MIK_ERR_INVALID;
    if (it->failed) return mik_fail(it->ctx, MIK_ERR_SINGULAR, "bicgstabl: this handle met a singular MR system in an earlier step and cannot be stepped");
    (void)hipSetDevice(it->ctx->device);
    return it->dtype == MIK_F64 ? bicg_step_impl<double>(it, (double *)residual) : bicg_step_impl<float>(it, (float *)residual);
}


// ---- MINRES -----------------------------------------------------------------------------------------------------------------
// iterate(::MINRESIterable) at src/minres.jl:95-159 with the sweeps of mik_axpy_dot (twice) and mik_minres_update: the Lanczos
// coefficients, the two Givens rotations, the right-hand side of the least-squares problem and the coefficients of the tail
// sweep stay on the device (k_minres_fin_proj, k_minres_fin_norm), so the host waits once per iteration -- for |rhs[2]|, the
// residual norm it returns -- instead of twice in the middle of it.  Same sweeps, same tree, same scalar arithmetic in the
// element type as the statement-by-statement path of the Python mirror (api.py, MINRESIterable.iterate).
namespace {

template <typename T> struct MinresDev {
    T H[4], rhs[2], c_prev, s_prev, c_curr, s_curr;
    T neg_h1_lanczos;                          // -H[2] as the previous iteration left it (:151): the coefficient of :104
    T neg_proj;                                // :109
    T inv_h3, neg_h1, neg_h0, inv_h2, rhs0;    // the tail sweep, :113 and :136-142
    T safmn2, safmx2;                          // constants of givensAlgorithm (computed once on the host: givens_constants)
    int range;                                 // 1: |v_next|^2 left the range of a safe norm -- the tail sweep is held back
};

struct MinresMirror { double resnorm; int range; unsigned long long seq; };

// LinearAlgebra.givensAlgorithm(f, g) -- givens_algorithm of mik_krylov.hip with the two constants passed in
template <typename T> __device__ void givens_dev(T f, T g, T safmn2, T safmx2, T &cs, T &sn, T &r)
{
    if (g == T(0)) { cs = T(1); sn = T(0); r = f; return; }
    if (f == T(0)) { cs = T(0); sn = T(1); r = g; return; }
    T f1 = f, g1 = g;
    T scale = fmax(fabs(f1), fabs(g1));
    int count = 0;
    if (scale >= safmx2) {
        do { ++count; f1 = f1 * safmn2; g1 = g1 * safmn2; scale = fmax(fabs(f1), fabs(g1)); } while (scale >= safmx2);
        { const T a = f1 * f1, b = g1 * g1; r = mik_sqrt(a + b); }
        cs = f1 / r; sn = g1 / r;
        for (int i = 0; i < count; ++i) r = r * safmx2;
    } else if (scale <= safmn2) {
        do { ++count; f1 = f1 * safmx2; g1 = g1 * safmx2; scale = fmax(fabs(f1), fabs(g1)); } while (scale <= safmn2);
        { const T a = f1 * f1, b = g1 * g1; r = mik_sqrt(a + b); }
        cs = f1 / r; sn = g1 / r;
        for (int i = 0; i < count; ++i) r = r * safmn2;
    } else {
        const T a = f1 * f1, b = g1 * g1;
        r = mik_sqrt(a + b); cs = f1 / r; sn = g1 / r;
    }
    if (fabs(f) > fabs(g) && cs < T(0)) { cs = -cs; sn = -sn; r = -r; }
}

// src/minres.jl:113-133 and :145-154 for the Lanczos norm h3 = H[4]: the rotations, rhs, the coefficients of the tail sweep,
// the state of the next iteration and the residual norm
template <typename T>
__device__ void minres_scalars(MinresDev<T> *d, T h3, long long iteration, int skew, MinresMirror *mirror, unsigned long long seq)
{
    T H0 = d->H[0], H1 = d->H[1], H2 = d->H[2];
    const T inv_h3 = T(1) / h3;                                              // :113
    if (iteration > 2) { H0 = d->s_prev * H1; H1 = d->c_prev * H1; }         // :116-119
    if (iteration > 1) {                                                     // :122-126
        const T a = -d->s_curr * H1, b = d->c_curr * H2;
        const T tmp = a + b;
        const T p = d->c_curr * H1, q = d->s_curr * H2;
        H1 = p + q;
        H2 = tmp;
    }
    T c, s, r;
    givens_dev<T>(H2, h3, d->safmn2, d->safmx2, c, s, r);                    // :129
    H2 = r;
    const T rhs1 = -s * d->rhs[0];                                           // :132
    const T rhs0 = c * d->rhs[0];                                            // :133
    d->inv_h3 = inv_h3; d->neg_h1 = -H1; d->neg_h0 = -H0; d->inv_h2 = T(1) / H2; d->rhs0 = rhs0;
    d->c_prev = d->c_curr; d->s_prev = d->s_curr; d->c_curr = c; d->s_curr = s;        // :147
    d->rhs[0] = rhs1; d->rhs[1] = rhs1;                                      // :148
    const T h1n = skew ? -h3 : h3;                                           // :151
    d->H[0] = H0; d->H[1] = h1n; d->H[2] = H2; d->H[3] = h3;
    d->neg_h1_lanczos = -h1n;
    d->range = 0;
    mirror->resnorm = (double)fabs(rhs1);                                    // :154
    mirror->range = 0;
    __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// :107 in front of :109 -- k_minres_fin_proj per workgroup of the orthogonalisation sweep (k_map_with)
template <typename T> struct ProMinresProj {
    MinresDev<T> *d;
    __device__ __forceinline__ void operator()(T tot, OpAxpyDot<T> &op, bool publish) const
    {
        op.alpha = Coef<T>{nullptr, -tot};
        if (publish) { d->H[2] = tot; d->neg_proj = -tot; }
    }
};

// proj = dot(v_curr, v_next) (:107): H[3] = proj, the coefficient of :109
template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_minres_fin_proj(const T *__restrict__ S, int64_t m, MinresDev<T> *d)
{
    __shared__ T lds16[16];
    const T tot = level2_sum(S, m, lds16);
    if (threadIdx.x == 0) { d->H[2] = tot; d->neg_proj = -tot; }
}

// the same over many partials (one per 256-row block when the SpMV forms proj): 16 single-wave workgroups and a ticket, as k_cg_fin_alpha --
// a single 1024-thread workgroup pulls 65,536 partials through one CU (11 us at 256^3 instead of 5)
template <typename T>
__global__ __launch_bounds__(64) void k_minres_fin_proj_spread(const T *__restrict__ S, int64_t m, MinresDev<T> *d, FinScratch<T> *fs)
{
    T tot;
    if (level2_sum_spread(S, m, fs, tot)) { d->H[2] = tot; d->neg_proj = -tot; }
}

// H[4] = norm(v_next) (:112), then the scalar part of the iteration
template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_minres_fin_norm(const T *__restrict__ S, int64_t m, MinresDev<T> *d, long long iteration, int skew,
                                                                     MinresMirror *mirror, unsigned long long seq)
{
    __shared__ T lds16[16];
    const T tot = level2_sum(S, m, lds16);
    if (threadIdx.x == 0) {
        if (mik_nrm_in_range(tot) || tot != tot) minres_scalars<T>(d, mik_sqrt(tot), iteration, skew, mirror, seq);
        else {                     // the host recomputes the norm with a scale (mik_safe_norm_slow) and calls k_minres_scalars
            d->range = 1; mirror->range = 1;
            __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <typename T>
__global__ void k_minres_scalars(MinresDev<T> *d, T h3, long long iteration, int skew, MinresMirror *mirror, unsigned long long seq)
{
    minres_scalars<T>(d, h3, iteration, skew, mirror, seq);
}

template <typename T> void givens_constants(T &safmn2, T &safmx2)      // as givens_algorithm (mik_krylov.hip) computes them
{
    const T eps = std::numeric_limits<T>::epsilon();
    const T safmin = std::numeric_limits<T>::min();
    safmn2 = std::pow(T(2), T((int)(std::log(safmin / eps) / std::log(T(2)) / T(2))));
    safmx2 = T(1) / safmn2;
}

}  // namespace

struct mik_minres {
    mik_ctx *ctx = nullptr;
    const mik_csr *A = nullptr;
    int dtype = MIK_F64, skew = 0;
    int64_t n = 0;
    void *x = nullptr, *v[3] = {nullptr, nullptr, nullptr}, *w[3] = {nullptr, nullptr, nullptr};   // prev, curr, next
    void *dev = nullptr;            // MinresDev<T>
    MinresMirror *mirror = nullptr;
    unsigned long long seq = 0;
    bool epilogue = false;          // the three Lanczos vectors are 16-byte aligned: the Lanczos step may ride on the SpMV (minres_epilogue)
    void *fin = nullptr;            // FinScratch<T>: wave sums + ticket of the spread level-2 sum of proj
};

extern "C" int mik_minres_destroy(mik_minres *it)
{
    if (!it) return MIK_OK;
    ctx_disown(it->ctx, it);
    (void)hipSetDevice(it->ctx->device);
    (void)hipStreamSynchronize(it->ctx->stream);
    if (it->dev) (void)hipFree(it->dev);
    if (it->fin) (void)hipFree(it->fin);
    if (it->mirror) (void)hipHostFree(it->mirror);
    delete it;
    return MIK_OK;
}

extern "C" int mik_minres_create(mik_ctx *ctx, const mik_csr *A, void *x, void *v_prev, void *v_curr, void *v_next, void *w_prev, void *w_curr,
                                 void *w_next, double resnorm0, int skew_hermitian, mik_minres **out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    if (!A || A->ctx != ctx || A->n_rows != A->n_cols) return mik_fail(ctx, MIK_ERR_MISMATCH, "mik_minres_create: A must be a square operator of this context");
    const int64_t n = A->n_rows;
    if (n && (!x || !v_prev || !v_curr || !v_next || !w_prev || !w_curr || !w_next)) return mik_fail(ctx, MIK_ERR_INVALID, "mik_minres_create: NULL vector");
    mik_minres *it = new (std::nothrow) mik_minres();
    if (!it) return mik_fail(ctx, MIK_ERR_NOMEM, "mik_minres_create: host allocation failed");
    it->ctx = ctx; it->A = A; it->dtype = A->dtype; it->skew = skew_hermitian ? 1 : 0; it->n = n; it->x = x;
    it->epilogue = mik_aligned16(v_prev) && mik_aligned16(v_curr) && mik_aligned16(v_next);   // the vectors allow it; whether the operator's kernel takes an epilogue is asked per step (minres_epilogue)
    it->v[0] = v_prev; it->v[1] = v_curr; it->v[2] = v_next;
    it->w[0] = w_prev; it->w[1] = w_curr; it->w[2] = w_next;
    (void)hipSetDevice(ctx->device);
    hipError_t e;
    const size_t db = A->dtype == MIK_F64 ? sizeof(MinresDev<double>) : sizeof(MinresDev<float>);
    if ((e = hipMalloc(&it->dev, db)) != hipSuccess || (e = hipMalloc(&it->fin, 512)) != hipSuccess || (e = hipMemset(it->fin, 0, 512)) != hipSuccess ||
        (e = hipHostMalloc((void **)&it->mirror, sizeof(MinresMirror), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) {
        const int rc = mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_minres_create: %s", hipGetErrorString(e));
        mik_minres_destroy(it);
        return rc;
    }
    memset(it->mirror, 0, sizeof(MinresMirror));
    // H = 0, rhs = [resnorm; 0], (c, s)_prev = (c, s)_curr = (1, 0)   (src/minres.jl:70-71, :76-77)
    if (A->dtype == MIK_F64) {
        MinresDev<double> h{};
        h.rhs[0] = resnorm0; h.c_prev = 1.0; h.c_curr = 1.0; givens_constants<double>(h.safmn2, h.safmx2);
        e = hipMemcpy(it->dev, &h, sizeof(h), hipMemcpyHostToDevice);
    } else {
        MinresDev<float> h{};
        h.rhs[0] = (float)resnorm0; h.c_prev = 1.0f; h.c_curr = 1.0f; givens_constants<float>(h.safmn2, h.safmx2);
        e = hipMemcpy(it->dev, &h, sizeof(h), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        const int rc = mik_fail(ctx, MIK_ERR_HIP, "mik_minres_create: %s", hipGetErrorString(e));
        mik_minres_destroy(it);
        return rc;
    }
    ctx->owned.push_back({it, [](void *h) { return mik_minres_destroy((mik_minres *)h); }});
    *out = it;
    return MIK_OK;
}

template <typename T> static int minres_wait(mik_minres *it)
{
    for (unsigned long long spins = 0;; ++spins) {
        if (__atomic_load_n(&it->mirror->seq, __ATOMIC_ACQUIRE) == it->seq) return MIK_OK;
        if ((spins & 0xFFFFF) == 0xFFFFF) {
            const hipError_t e = hipStreamQuery(it->ctx->stream);
            if (e == hipSuccess) {
                if (__atomic_load_n(&it->mirror->seq, __ATOMIC_ACQUIRE) == it->seq) return MIK_OK;
                return mik_fail(it->ctx, MIK_ERR_HIP, "minres: stream idle but iteration %llu was never published", it->seq);
            }
            if (e != hipErrorNotReady) return mik_fail(it->ctx, MIK_ERR_HIP, "minres: %s while waiting for an iteration", hipGetErrorString(e));
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

// Does this step's Lanczos update ride on the SpMV launch?  Asked per step, like bicg_fuses: the operator's layout and the development knobs
// may change between steps (mik_csr_set_layout, mik_set_tuning) -- the step then falls back to the separate sweep instead of failing, and
// mik_minres_proj_shape reports the shape of the NEXT step's projection.
static bool minres_epilogue(const mik_minres *it) { return it->epilogue && mik_spmv_has_epilogue(it->A); }

template <typename T> static int minres_step_impl(mik_minres *it, int64_t iteration, T *resnorm)
{
    mik_ctx *ctx = it->ctx;
    const int64_t n = it->n, nseg = mik_nseg<T>(n);
    MinresDev<T> *d = (MinresDev<T> *)it->dev;
    T *x = (T *)it->x, *v_prev = (T *)it->v[0], *v_curr = (T *)it->v[1], *v_next = (T *)it->v[2];
    T *w_prev = (T *)it->w[0], *w_curr = (T *)it->w[1], *w_next = (T *)it->w[2];
    // The Lanczos step (:102-107) as the EPILOGUE of the SpMV where the operator's kernel takes one (mik_spmv_has_epilogue, round 4): v_next =
    // A v_curr - H[2] v_prev is stored once and proj = dot(v_curr, v_next) leaves the launch as one partial per 256-row block (the shape
    // of the dot fused into the CG SpMV: mik_minres_proj_shape) -- the sweep that re-read v_prev, v_next and v_curr is gone.
    const bool ep = minres_epilogue(it);
    const int64_t na = ep ? mik_spmv_nwg(n) : nseg;           // partials of the projection
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(na + nseg, 1)));
    const bool lean = na <= 1024 && nseg <= 1024 && ctx->tuning[MIK_KNOB_SOLVER_FORM] == 0;   // the orthogonalisation sweep finalises the projection itself (k_map_with)
    T *part_a = (T *)ctx->partials, *part_b = lean ? part_a + na : part_a;
    if (ep) {
        ctx->spmv_ep_w = iteration > 1 ? v_prev : nullptr;   // (no v_prev in the first iteration: the plain fused dot)
        ctx->spmv_ep_c = &d->neg_h1_lanczos;
        const int rc_ep = mik_spmv_launch<T>(ctx, it->A, v_curr, v_next, true, part_a, nullptr);               // :102, :104, :107
        ctx->spmv_ep_w = nullptr; ctx->spmv_ep_c = nullptr;
        MIK_TRY(rc_ep);
    } else {
        MIK_TRY(mik_spmv_launch<T>(ctx, it->A, v_curr, v_next, false, nullptr, nullptr));                    // :102
        // v_next -= H[2] v_prev (iteration > 1) and proj = dot(v_curr, v_next)                               :104, :107; v_prev is dead afterwards
        const T *xp = iteration > 1 ? v_prev : nullptr;
        OpAxpyDot<T> op{xp, v_next, v_curr, coef_ptr<T>(&d->neg_h1_lanczos), 1};
        const bool vec = mik_aligned16(v_next) && (!xp || mik_aligned16(xp)) && mik_aligned16(v_curr);
        MIK_TRY((launch_map<T>(ctx, n, op, vec, part_a, nullptr)));
    }
    if (!lean) {
        if (na > 16384) hipLaunchKernelGGL((k_minres_fin_proj_spread<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)part_a, na, d, (FinScratch<T> *)it->fin);
        else hipLaunchKernelGGL((k_minres_fin_proj<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)part_a, na, d);
        MIK_LAUNCH_CHECK(ctx);
    }
    it->seq += 1;
    {   // v_next -= proj v_curr; H[4] = norm(v_next)                                                          :109, :112
        OpAxpyDot<T> op{v_curr, v_next, nullptr, coef_ptr<T>(&d->neg_proj), 0};
        const bool vec = mik_aligned16(v_next) && mik_aligned16(v_curr);
        if (lean) MIK_TRY((launch_map_with<T>(ctx, n, op, ProMinresProj<T>{d}, vec, (const T *)part_a, (int)na, part_b)));
        else MIK_TRY((launch_map<T>(ctx, n, op, vec, part_b, nullptr)));
        hipLaunchKernelGGL((k_minres_fin_norm<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)part_b, nseg, d, (long long)iteration,
                           it->skew, it->mirror, it->seq);
        MIK_LAUNCH_CHECK(ctx);
    }
    auto tail = [&]() {   // v_next /= H[4]; w_next = (v_curr - H[2] w_curr - H[1] w_prev) / H[3]; x += rhs[1] w_next   :113, :136-142
        const T *wc = iteration > 1 ? w_curr : nullptr, *wp = iteration > 2 ? w_prev : nullptr;
        OpMinresUpdate<T> op{v_next, v_curr, wc, wp, w_next, x, coef_ptr<T>(&d->inv_h3), coef_ptr<T>(&d->neg_h1), coef_ptr<T>(&d->neg_h0),
                             coef_ptr<T>(&d->inv_h2), coef_ptr<T>(&d->rhs0), 15};
        const bool vec = mik_aligned16(v_next) && mik_aligned16(v_curr) && mik_aligned16(w_next) && mik_aligned16(x) && (!wc || mik_aligned16(wc)) &&
                         (!wp || mik_aligned16(wp));
        return launch_map<T>(ctx, n, op, vec, (T *)nullptr, (const int *)&d->range);     // held back while the norm is being rescaled
    };
    MIK_TRY(tail());
    MIK_TRY(minres_wait<T>(it));
    if (it->mirror->range) {       // |v_next|^2 outside the range of a safe sum of squares: the scaled norm, then the scalars and the tail
        T h3;
        MIK_TRY(mik_safe_norm_slow<T>(ctx, n, v_next, &h3));
        it->seq += 1;
        hipLaunchKernelGGL((k_minres_scalars<T>), dim3(1), dim3(1), 0, ctx->stream, d, h3, (long long)iteration, it->skew, it->mirror, it->seq);
        MIK_LAUNCH_CHECK(ctx);
        MIK_TRY(tail());
        MIK_TRY(minres_wait<T>(it));
    }
    *resnorm = (T)it->mirror->resnorm;
    void *t = it->v[0]; it->v[0] = it->v[1]; it->v[1] = it->v[2]; it->v[2] = t;              // :145
    t = it->w[0]; it->w[0] = it->w[1]; it->w[1] = it->w[2]; it->w[2] = t;                    // :146
    return MIK_OK;
}

extern "C" int mik_minres_proj_shape(const mik_minres *it, int *W, int *L)
{
    if (!it) return MIK_ERR_INVALID;
    if (minres_epilogue(it)) return mik_spmv_dot_shape(W, L);
    return mik_reduce_shape(it->dtype, W, L);
}

extern "C" int mik_minres_step(mik_minres *it, int64_t iteration, void *resnorm)
{
    if (!it || !resnorm || iteration < 1) return MIK_ERR_INVALID;
    (void)hipSetDevice(it->ctx->device);
    return it->dtype == MIK_F64 ? minres_step_impl<double>(it, iteration, (double *)resnorm) : minres_step_impl<float>(it, iteration, (float *)resnorm);
}
