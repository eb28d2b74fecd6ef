// mik_krylov.hip -- L2 Krylov helpers (orthogonalize_and_normalize!, gemv) and the L3 iterables
// (CGIterable / PCGIterable, GMRESIterable) of include/mik.h.
//
// Host-side control flow restates the reference's iterate() methods (src/cg.jl:43-100,
// src/gmres.jl:57-106); all vector arithmetic runs in the kernels of mik_kernels.h.
#include <algorithm>
#include <cfloat>
#include <cstddef>
#include <cmath>
#include <map>
#include <new>
#include <tuple>

#include "mik_kernels.h"
#include "mik_mgs_res.h"
#include "mik_iter.h"
#include "mik_mail.h"

template <typename T>
int mik_spmv_launch(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done);
template <typename T>
int mik_spmv_launch_range(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done, int rb_begin,
                          int rb_count);
template <typename T>
int mik_spmv_launch_outside(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done, int skip_begin, int skip_end);
bool mik_spmv_can_split(const mik_csr *A);
bool mik_spmv_is_light(const mik_csr *A);



// =============================================================================================
// Hessenberg least squares (host) -- src/hessenberg.jl:15-46
// =============================================================================================

// LinearAlgebra.givensAlgorithm(f, g): LAPACK xLARTG-style plane rotation with power-of-two
// rescaling; [c s; -s c] * [f; g] = [r; 0].
template <typename T> static void givens_algorithm(T f, T g, T &cs, T &sn, T &r)
{
    // safmn2 = LinearAlgebra.floatmin2(T) = 2^trunc(log2(floatmin(T) / eps(T)) / 2) with eps(T) = the spacing at 1
    // (2^-52 / 2^-23, NOT LAPACK's unit roundoff): 2^-485 for Float64 (0x21a0000000000000), 2^-51 for Float32.
    const T eps = std::numeric_limits<T>::epsilon();
    const T safmin = std::numeric_limits<T>::min();
    const T safmn2 = std::pow(T(2), T((int)(std::log(safmin / eps) / std::log(T(2)) / T(2))));
    const T safmx2 = T(1) / safmn2;
    if (g == T(0)) { cs = T(1); sn = T(0); r = f; return; }
    if (f == T(0)) { cs = T(0); sn = T(1); r = g; return; }
    T f1 = f, g1 = g;
    T scale = std::max(std::fabs(f1), std::fabs(g1));
    int count = 0;
    if (scale >= safmx2) {
        do { ++count; f1 *= safmn2; g1 *= safmn2; scale = std::max(std::fabs(f1), std::fabs(g1)); } while (scale >= safmx2);
        r = std::sqrt(f1 * f1 + g1 * g1); cs = f1 / r; sn = g1 / r;
        for (int i = 0; i < count; ++i) r *= safmx2;
    } else if (scale <= safmn2) {
        do { ++count; f1 *= safmx2; g1 *= safmx2; scale = std::max(std::fabs(f1), std::fabs(g1)); } while (scale <= safmn2);
        r = std::sqrt(f1 * f1 + g1 * g1); cs = f1 / r; sn = g1 / r;
        for (int i = 0; i < count; ++i) r *= safmn2;
    } else {
        r = std::sqrt(f1 * f1 + g1 * g1); cs = f1 / r; sn = g1 / r;
    }
    if (std::fabs(f) > std::fabs(g) && cs < T(0)) { cs = -cs; sn = -sn; r = -r; }
}

template <typename T> static void hessenberg_ldiv(T *H, int64_t ldh, int width, T *rhs)
{
    auto at = [&](int i, int j) -> T & { return H[(size_t)j * ldh + i]; };
    for (int i = 0; i < width; ++i) {
        T c, s, rr;
        givens_algorithm(at(i, i), at(i + 1, i), c, s, rr);
        at(i, i) = c * at(i, i) + s * at(i + 1, i);
        for (int j = i + 1; j < width; ++j) {
            const T tmp = -s * at(i, j) + c * at(i + 1, j);
            at(i, j) = c * at(i, j) + s * at(i + 1, j);
            at(i + 1, j) = tmp;
        }
        const T tmp = -s * rhs[i] + c * rhs[i + 1];
        rhs[i] = c * rhs[i] + s * rhs[i + 1];
        rhs[i + 1] = tmp;
    }
    // UpperTriangular back substitution, column sweep from the last column
    for (int j = width - 1; j >= 0; --j) {
        rhs[j] = rhs[j] / at(j, j);
        const T t = rhs[j];
        for (int i = 0; i < j; ++i) rhs[i] = rhs[i] - t * at(i, j);
    }
}

extern "C" int mik_hessenberg_ldiv(int dtype, void *H, int64_t ldh, int width, void *rhs)
{
    if (!H || !rhs || width < 0 || ldh < width + 1) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) hessenberg_ldiv<double>((double *)H, ldh, width, (double *)rhs);
    else if (dtype == MIK_F32) hessenberg_ldiv<float>((float *)H, ldh, width, (float *)rhs);
    else return MIK_ERR_INVALID;
    return MIK_OK;
}

// =============================================================================================
// gemv-N and orthogonalisation
// =============================================================================================
// Stream the Krylov basis past the caches when it cannot stay resident anyway (k columns exceed the 256 MB
// Infinity Cache), so that the vector being orthogonalised does.  
template <typename T> static inline int mik_basis_nt(const mik_ctx *ctx, int64_t n, int k, int bit)
{
    return (double)n * (double)k * sizeof(T) > 192.0e6 ? 1 : 0;
}
// Cache hints of a pass of the multi-launch Modified Gram-Schmidt chain (OpMgsPass::nt: 1 = v, 2 = z, 4 = w load, 8 = w store non-temporal).
// The column that is subtracted in this pass and not needed again is streamed; when three vectors cannot share the 256 MB Infinity Cache
// anyway, w is streamed too (load and store), so that the column projected on in this pass -- the one the NEXT pass subtracts -- is what
// the cache keeps.  Measured at 256^3 fp64 (gmres!(30), 60 inner iterations, one box, round 5): mask 0: 1,879 us per inner iteration,
// 1 (v): 1,651, 3 (v, z): 1,684, 9: 1,648, 5: 1,660, 11: 1,686, 15 (all): 1,835, 13 (v, w): 1,622.
static inline int mik_mgs_pass_hints(const mik_ctx *ctx, int64_t n, size_t es)
{
    (void)ctx;
    return (double)n * (double)es > 96.0e6 ? 13 : 1;
}

template <typename T>
static int gemv_n_dev(mik_ctx *ctx, int64_t n, int k, const T *V, int64_t ldv, const T *cf_dev, T alpha, T *y)
{
    const int64_t nseg = mik_nseg<T>(n);
    if (nseg == 0 || k == 0) return MIK_OK;
    const int grid = (int)std::min<int64_t>(nseg, mik_max_grid(ctx));
    const bool vec = mik_aligned16(V) && mik_aligned16(y) && (ldv % VT<T>::W == 0);
    if (vec) hipLaunchKernelGGL((k_gemv_n<T, true>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, k, V, ldv, cf_dev, alpha, y, mik_basis_nt<T>(ctx, n, k, 2));
    else hipLaunchKernelGGL((k_gemv_n<T, false>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, k, V, ldv, cf_dev, alpha, y, 0);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// Copy k host scalars into the coefficient area of the context (device), at element `slot`.
template <typename T> static int coef_upload(mik_ctx *ctx, int slot, const T *host, int k)
{
    if ((size_t)(slot + k) * sizeof(T) > mik_ctx::COEF_SAFE_SLOT) return mik_fail(ctx, MIK_ERR_NOTIMPL, "coefficient block too large (k = %d)", k);
    MIK_HIP(ctx, mik_wait(ctx));                        // staging buffer must be idle
    memcpy((T *)ctx->coef_host + slot, host, sizeof(T) * k);
    MIK_HIP(ctx, hipMemcpyAsync((T *)ctx->coef + slot, (T *)ctx->coef_host + slot, sizeof(T) * k, hipMemcpyHostToDevice, ctx->stream));
    return MIK_OK;
}

template <typename T> static int coef_download(mik_ctx *ctx, int slot, T *host, int k)
{
    return mik_read_scalars<T>(ctx, (const T *)ctx->coef + slot, k, host);
}

template <typename T>
static int gemv_n_impl(mik_ctx *ctx, int64_t n, int k, const void *V, int64_t ldv, const void *c, const void *alpha, void *y)
{
    if (k == 0 || n == 0) return MIK_OK;
    MIK_TRY(coef_upload<T>(ctx, 0, (const T *)c, k));
    return gemv_n_dev<T>(ctx, n, k, (const T *)V, ldv, (const T *)ctx->coef, *(const T *)alpha, (T *)y);
}

extern "C" int mik_gemv_n(mik_ctx *ctx, int dtype, int64_t n, int k, const void *V, int64_t ldv, const void *c,
                          const void *alpha, void *y)
{
    if (!ctx || n < 0 || k < 0 || !alpha || (k && !c) || (n && k && (!V || !y)) || ldv < n) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return gemv_n_impl<double>(ctx, n, k, V, ldv, c, alpha, y);
    if (dtype == MIK_F32) return gemv_n_impl<float>(ctx, n, k, V, ldv, c, alpha, y);
    return MIK_ERR_INVALID;
}

template <typename T> static int finalize_store(mik_ctx *ctx, int64_t nseg, int cols, T *out_dev)
{
    hipLaunchKernelGGL((k_finalize_store<T>), dim3(cols), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, nseg,
                       nseg, out_dev, (const int *)nullptr);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

template <typename T> static int finalize_nrm_inv(mik_ctx *ctx, int64_t nseg, T *out_dev)
{
    hipLaunchKernelGGL((k_finalize_nrm_inv<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, nseg, out_dev);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

template <typename T>
static int multidot(mik_ctx *ctx, int64_t n, int k, const T *V, int64_t ldv, const T *w, T *out_dev)
{
    const int64_t nseg = mik_nseg<T>(n);
    if (k <= 0) return MIK_OK;
    if (nseg == 0) {   // empty vectors: every dot is +0
        MIK_HIP(ctx, hipMemsetAsync(out_dev, 0, sizeof(T) * k, ctx->stream));
        return MIK_OK;
    }
    const int grid = (int)std::min<int64_t>(nseg, mik_max_grid(ctx));
    const bool vec = mik_aligned16(V) && mik_aligned16(w) && (ldv % VT<T>::W == 0);
    if (vec) hipLaunchKernelGGL((k_multidot<T, true>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, k, V, ldv, w, (T *)ctx->partials, mik_basis_nt<T>(ctx, n, k, 4));
    else hipLaunchKernelGGL((k_multidot<T, false>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, k, V, ldv, w, (T *)ctx->partials, 0);
    MIK_LAUNCH_CHECK(ctx);
    return finalize_store<T>(ctx, nseg, k, out_dev);
}

// orthogonalize_and_normalize!(V[:, 1:k], w, h, method) -> nrm   -- src/orthogonalize.jl:13-79
// Coefficient area layout (elements of T): [0, k) = h, [k] = nrm, [k+1] = 1/nrm, [k+2, 2k+2) = DGKS correction.

// Workspace the chains below need in ctx->partials (allocate BEFORE a graph capture).
template <typename T> static size_t orthogonalize_workspace(int64_t n, int k)
{
    const int64_t nseg = mik_nseg<T>(n);
    return sizeof(T) * std::max<size_t>((size_t)std::max<int64_t>(nseg, 1) * (size_t)std::max(k, 1), 2 * 1024);
}

// Kernels only (no host synchronisation; capturable into a hipGraph): ModifiedGramSchmidt / ClassicalGramSchmidt
// up to and including w .*= inv(nrm); leaves h in coef[0, k), nrm in coef[k].  For DGKS: the first CGS sweep and
// the norm (the re-orthogonalisation loop needs the host, see orthogonalize_impl).
template <typename T>
static int orthogonalize_enqueue(mik_ctx *ctx, int64_t n, int k, const T *V, int64_t ldv, T *w, int method, const T *const *cols = nullptr)
{
    const int64_t nseg = mik_nseg<T>(n);
    T *hd = (T *)ctx->coef;
    T *part = (T *)ctx->partials;
    const bool vecw = mik_aligned16(w);
    bool vec = vecw && (cols || (mik_aligned16(V) && (ldv % VT<T>::W == 0)));
    // `cols` (ModifiedGramSchmidt only): V is a vector of k separate device vectors -- the method of src/orthogonalize.jl:53-65
    auto col = [&](int i) -> const T * { return cols ? cols[i] : V + (int64_t)i * ldv; };
    if (cols) for (int i = 0; i < k; ++i) vec = vec && mik_aligned16(cols[i]);
    OpDot<T> dn{w, w};

    if (method == MIK_MGS && nseg <= 1024 && ctx->tuning[MIK_KNOB_GS] != 1) {
        // src/orthogonalize.jl:69-76, launch-lean form for n up to ~1M: every pass finalises the
        // previous pass's reduction itself (k_map_pro), so the chain is k + 2 launches instead of
        // 2k + 3.  Segment sums ping-pong between two buffers (a pass reads one while writing the other).
        T *P[2] = {part, part + 1024};
        const int m = (int)nseg;
        if (k > 0) {
            OpDot<T> d0{col(0), w};
            MIK_TRY((launch_map<T>(ctx, n, d0, vec, P[0], nullptr)));
            for (int i = 0; i + 1 < k; ++i) {
                OpMgsPass<T, false> op{w, col(i), col(i + 1), coef_val<T>(T(0))};
                MIK_TRY((launch_map_pro<T, 1>(ctx, n, op, vec, P[(i + 1) & 1], P[i & 1], m, hd + i)));
            }
            OpMgsPass<T, true> last{w, col(k - 1), nullptr, coef_val<T>(T(0))};
            MIK_TRY((launch_map_pro<T, 1>(ctx, n, last, vec, P[k & 1], P[(k - 1) & 1], m, hd + k - 1)));
        } else {
            MIK_TRY((launch_map<T>(ctx, n, dn, vecw, P[0], nullptr)));
        }
        OpScal<T> sc{w, coef_val<T>(T(0))};                                // w .*= inv(norm(w))  :75-76
        return launch_map_pro<T, 2>(ctx, n, sc, vecw, (T *)nullptr, P[k & 1], m, hd + k);
    }
    if (method == MIK_MGS) {
        // src/orthogonalize.jl:69-76.  Pass i subtracts h[i] * V[:, i] from w and, in the same sweep,
        // accumulates the next projection dot(V[:, i+1], w) -- or norm(w)^2 on the last pass.
        if (k > 0) {
            OpDot<T> d0{col(0), w};
            MIK_TRY((launch_map<T>(ctx, n, d0, vec, part, nullptr)));
            MIK_TRY(finalize_store<T>(ctx, nseg, 1, hd));
            for (int i = 0; i + 1 < k; ++i) {
                OpMgsPass<T, false> op{w, col(i), col(i + 1), coef_ptr<T>(hd + i), mik_mgs_pass_hints(ctx, n, sizeof(T))};
                MIK_TRY((launch_map<T>(ctx, n, op, vec, part, nullptr)));
                MIK_TRY(finalize_store<T>(ctx, nseg, 1, hd + i + 1));
            }
            OpMgsPass<T, true> last{w, col(k - 1), nullptr, coef_ptr<T>(hd + k - 1), mik_mgs_pass_hints(ctx, n, sizeof(T))};
            MIK_TRY((launch_map<T>(ctx, n, last, vec, part, nullptr)));
        } else {
            MIK_TRY((launch_map<T>(ctx, n, dn, vecw, part, nullptr)));
        }
        MIK_TRY(finalize_nrm_inv<T>(ctx, nseg, hd + k));
    } else {
        // src/orthogonalize.jl:15-17 / :43-45: h = V' w (batched dot), w -= V h (axpy sweep), norm
        MIK_TRY(multidot<T>(ctx, n, k, V, ldv, w, hd));
        MIK_TRY(gemv_n_dev<T>(ctx, n, k, V, ldv, hd, T(-1), w));
        MIK_TRY((launch_map<T>(ctx, n, dn, vecw, part, nullptr)));
        MIK_TRY(finalize_nrm_inv<T>(ctx, nseg, hd + k));
        if (method == MIK_DGKS) return MIK_OK;                           // normalisation after the host loop
    }
    OpScal<T> sc{w, coef_ptr<T>(hd + k + 1)};                           // w .*= inv(nrm)  :76 / :48
    return launch_map<T>(ctx, n, sc, vecw, (T *)nullptr, nullptr);
}

// The closing norm of a Gram-Schmidt chain came back as NaN: its sum of squares was outside the safe range and the
// kernels left w unscaled (k_finalize_nrm_inv / k_map_pro<2> multiply by 1).  Scaled norm, then w .*= inv(nrm).
template <typename T> static int orth_rescale(mik_ctx *ctx, int64_t n, T *w, T *nrm_host)
{
    T nrm;
    MIK_TRY(mik_safe_norm_slow<T>(ctx, n, w, &nrm));
    OpScal<T> sc{w, coef_val<T>(T(1) / nrm)};
    MIK_TRY((launch_map<T>(ctx, n, sc, mik_aligned16(w), (T *)nullptr, nullptr)));
    *nrm_host = nrm;
    return MIK_OK;
}

// The DGKS loop of src/orthogonalize.jl:26-36 from a given state {w (unscaled), hh, nrm, projection size}, then the
// normalisation; multi-launch chain + host control.  Shared by orthogonalize_impl and the single-launch kernel's hand-back.
template <typename T> static T dgks_small_norm(const T *v, int len)
{
    T s = T(0);
    for (int j = 0; j < len; ++j) { T p = v[j] * v[j]; s = s + p; }
    return (T)std::sqrt(s);
}

template <typename T>
static int dgks_host_loop(mik_ctx *ctx, int64_t n, int k, const T *V, int64_t ldv, T *w, T *hh, T *nrm_io, T projection_size)
{
    const int64_t nseg = mik_nseg<T>(n);
    MIK_TRY(mik_ensure_partials(ctx, orthogonalize_workspace<T>(n, k)));
    T *hd = (T *)ctx->coef;
    const bool vecw = mik_aligned16(w);
    OpDot<T> dn{w, w};
    std::vector<T> corr(std::max(k, 1));
    T nrm = *nrm_io;
    const T eta = T(1) / std::sqrt(T(2));                               // :20
    while (nrm < eta * projection_size) {                                // :26
        T *cd = hd + k + 2;
        MIK_TRY(multidot<T>(ctx, n, k, V, ldv, w, cd));                  // :27
        MIK_TRY(gemv_n_dev<T>(ctx, n, k, V, ldv, cd, T(-1), w));         // :30
        MIK_TRY((launch_map<T>(ctx, n, dn, vecw, (T *)ctx->partials, nullptr)));
        MIK_TRY(finalize_nrm_inv<T>(ctx, nseg, hd + k));                // :32
        MIK_TRY(coef_download<T>(ctx, k + 2, corr.data(), k));
        T nn[2];
        MIK_TRY(coef_download<T>(ctx, k, nn, 2));
        projection_size = dgks_small_norm<T>(corr.data(), k);           // :28
        for (int j = 0; j < k; ++j) hh[j] = hh[j] + corr[j];            // :31
        nrm = nn[0];
        if (nrm != nrm) MIK_TRY(mik_safe_norm_slow<T>(ctx, n, w, &nrm));
    }
    OpScal<T> sc{w, coef_val<T>(T(1) / nrm)};                           // :36 (same IEEE quotient the device forms)
    MIK_TRY((launch_map<T>(ctx, n, sc, vecw, (T *)nullptr, nullptr)));
    *nrm_io = nrm;
    return MIK_OK;
}

template <typename T>
static int orthogonalize_impl(mik_ctx *ctx, int64_t n, int k, const T *V, int64_t ldv, T *w, T *h_host, T *nrm_host, int method,
                              const T *const *cols = nullptr)
{
    if ((size_t)(2 * k + 4) * sizeof(T) > mik_ctx::COEF_SAFE_SLOT) return mik_fail(ctx, MIK_ERR_NOTIMPL, "orthogonalize: k = %d too large", k);
    MIK_TRY(mik_ensure_partials(ctx, orthogonalize_workspace<T>(n, k)));
    MIK_TRY(orthogonalize_enqueue<T>(ctx, n, k, V, ldv, w, method, cols));
    if (method == MIK_DGKS) {                                            // src/orthogonalize.jl:20-36
        std::vector<T> hh(k + 2);
        MIK_TRY(coef_download<T>(ctx, 0, hh.data(), k + 2));
        T nrm = hh[k];
        if (nrm != nrm) MIK_TRY(mik_safe_norm_slow<T>(ctx, n, w, &nrm));   // sum of squares outside the safe range (k_finalize_nrm_inv)
        T projection_size = dgks_small_norm<T>(hh.data(), k);          // :22
        MIK_TRY(dgks_host_loop<T>(ctx, n, k, V, ldv, w, hh.data(), &nrm, projection_size));
        for (int j = 0; j < k; ++j) h_host[j] = hh[j];
        *nrm_host = nrm;
        return MIK_OK;
    }
    std::vector<T> out(k + 1);
    MIK_TRY(coef_download<T>(ctx, 0, out.data(), k + 1));
    for (int j = 0; j < k; ++j) h_host[j] = out[j];
    *nrm_host = out[k];
    if (out[k] != out[k]) MIK_TRY(orth_rescale<T>(ctx, n, w, nrm_host));
    return MIK_OK;
}

extern "C" int mik_orthogonalize(mik_ctx *ctx, int dtype, int64_t n, int k, const void *V, int64_t ldv, void *w, void *h,
                                 void *nrm, int method)
{
    if (!ctx || n < 0 || k < 0 || !nrm || (k && (!h || !V)) || (n && !w) || (k && ldv < n)) return MIK_ERR_INVALID;
    if (method != MIK_MGS && method != MIK_CGS && method != MIK_DGKS) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return orthogonalize_impl<double>(ctx, n, k, (const double *)V, ldv, (double *)w, (double *)h, (double *)nrm, method);
    if (dtype == MIK_F32) return orthogonalize_impl<float>(ctx, n, k, (const float *)V, ldv, (float *)w, (float *)h, (float *)nrm, method);
    return MIK_ERR_INVALID;
}

extern "C" int mik_orthogonalize_vectors(mik_ctx *ctx, int dtype, int64_t n, int k, const void *const *V, void *w, void *h, void *nrm)
{
    if (!ctx || n < 0 || k < 0 || !nrm || (k && (!h || !V)) || (n && !w)) return MIK_ERR_INVALID;
    for (int i = 0; i < k; ++i) if (!V[i] && n) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return orthogonalize_impl<double>(ctx, n, k, nullptr, 0, (double *)w, (double *)h, (double *)nrm, MIK_MGS, (const double *const *)V);
    if (dtype == MIK_F32) return orthogonalize_impl<float>(ctx, n, k, nullptr, 0, (float *)w, (float *)h, (float *)nrm, MIK_MGS, (const float *const *)V);
    return MIK_ERR_INVALID;
}

// =============================================================================================
// CGIterable / PCGIterable
// =============================================================================================
// (level2_sum_spread / FinScratch: csrc/mik_iter.h -- shared with the mailbox finalisers of csrc/mik_comm.hip)

// the same for TWO reductions finalised by one launch (the fused PCG tail): one ticket, the last arrival adds both sets
template <typename T> struct FinScratch2 { T ws[MIK_FIN_WGS]; T ws2[MIK_FIN_WGS]; unsigned ticket; };

template <typename T>
__device__ __forceinline__ bool level2_sum_spread2(const T *__restrict__ S1, const T *__restrict__ S2, int64_t m, FinScratch2<T> *fs, T &tot1, T &tot2)
{
    const int w = blockIdx.x, lane = threadIdx.x;              // blockDim.x == 64, gridDim.x == MIK_FIN_WGS
    T a1 = T(0), a2 = T(0);
    int64_t j = 64 * (int64_t)w + lane;
    for (; j + 15 * (int64_t)MIK_FIN_THREADS < m; j += 16 * (int64_t)MIK_FIN_THREADS) {
        T v[16], z[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { v[q] = S1[j + q * (int64_t)MIK_FIN_THREADS]; z[q] = S2[j + q * (int64_t)MIK_FIN_THREADS]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) { a1 = a1 + v[q]; a2 = a2 + z[q]; }
    }
    for (; j < m; j += MIK_FIN_THREADS) { a1 = a1 + S1[j]; a2 = a2 + S2[j]; }
    a1 = wave_tree(a1);
    a2 = wave_tree(a2);
    bool last = false;
    if (lane == 0) {
        __hip_atomic_store(&fs->ws[w], a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&fs->ws2[w], a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tk = __hip_atomic_fetch_add(&fs->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk == (unsigned)gridDim.x - 1u) {
            T t = __hip_atomic_load(&fs->ws[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            T z = __hip_atomic_load(&fs->ws2[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 1; q < MIK_FIN_WGS; ++q) {
                t = t + __hip_atomic_load(&fs->ws[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                z = z + __hip_atomic_load(&fs->ws2[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(&fs->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tot1 = t; tot2 = z;
            last = true;
        }
    }
    return last;
}

// after norm(r) of cg_iterator! (src/cg.jl:140-152)
template <typename T> __device__ __forceinline__ void cg_init_scalars(CgDev<T> *d, T tot, T res, T reltol, T abstol, long long maxiter)
{
    const T a = reltol * res;
    d->rr = tot;
    d->res = res;
    d->prev_res = T(1);                       // one(residual)      :146
    d->rho = T(1);                            // one(eltype(x))     :151
    d->tol = a > abstol ? a : abstol;         // :141
    d->beta = (res * res) / (T(1) * T(1));    // what the first iterate() will use (:50)
    d->alpha = T(0);
    d->dot_uc = T(0);
    d->done = (0 >= maxiter || res <= d->tol) ? 1 : 0;
    d->nhist = 0;
    d->x_pending = 0;
}

template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_cg_fin_init(const T *__restrict__ S, int64_t m, CgDev<T> *d, T reltol, T abstol,
                                                                 long long maxiter)
{
    __shared__ T lds16[16];
    T tot = level2_sum(S, m, lds16);
    if (threadIdx.x == 0) cg_init_scalars(d, tot, mik_sqrt(tot), reltol, abstol, maxiter);   // the host re-does this if tot is out of range
}

// the same with a residual norm the host obtained through the scaled pass (mik_safe_norm_slow)
template <typename T> __global__ void k_cg_set_init(CgDev<T> *d, T res, T reltol, T abstol, long long maxiter)
{
    cg_init_scalars(d, res * res, res, reltol, abstol, maxiter);
}

// alpha = residual^2 / dot(u, c) (src/cg.jl:55) or rho / dot(u, c) (:90)
template <typename T>
__global__ __launch_bounds__(64) void k_cg_fin_alpha(const T *__restrict__ S, int64_t m, CgDev<T> *d, int pcg, FinScratch<T> *fs)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) d->x_pending = 0;      // the sweep over u before this launch applied it (OpXpbyX)
    if (d->done) return;
    T tot;
    if (level2_sum_spread(S, m, fs, tot)) {
        d->dot_uc = tot;
        const T num = pcg ? d->rho : d->res * d->res;
        d->alpha = num / tot;
    }
}

// rho = dot(c, r); beta = rho / rho_prev (src/cg.jl:81-85)
template <typename T>
__global__ __launch_bounds__(64) void k_cg_fin_rho(const T *__restrict__ S, int64_t m, CgDev<T> *d, FinScratch<T> *fs)
{
    if (d->done) return;
    T tot;
    if (level2_sum_spread(S, m, fs, tot)) {
        const T rho_prev = d->rho;
        d->rho = tot;
        d->beta = tot / rho_prev;
    }
}

// residual = norm(r) (src/cg.jl:61-62 / :96), history, and the stopping test of :36 for the NEXT
// iterate() call (iteration index `it_next`)
template <typename T>
__device__ __forceinline__ void cg_res_scalars(CgDev<T> *d, T tot, T res, T *__restrict__ hist, long long it_next, long long maxiter,
                                               CgMirror *mirror, unsigned long long seq, int hist_index, int pcg_fused = 0)
{
    const T prev = d->res;
    d->rr = tot;
    d->prev_res = prev;
    d->res = res;
    d->beta = pcg_fused ? d->beta_rho : (res * res) / (prev * prev);    // :50 of the next step (PCG: rho / rho_prev, :85)
    hist[hist_index] = res;                   // step `hist_index` of this host call (the host zeroes mirror->nhist)
    const int dn = (it_next >= maxiter || res <= d->tol) ? 1 : 0;
    if (dn) d->done = 1;
    mirror->res = (double)res;
    mirror->prev_res = (double)prev;
    mirror->done = dn;
    mirror->nhist = hist_index + 1;
    __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
__global__ __launch_bounds__(64) void k_cg_fin_res(const T *__restrict__ S, int64_t m, CgDev<T> *d, T *__restrict__ hist,
                                                    long long it_next, long long maxiter, CgMirror *mirror,
                                                    unsigned long long seq, int hist_index, FinScratch<T> *fs, int fuse_x)
{
    if (d->done) {
        // a no-op step (the stopping test fired earlier in this batch): still publish, state unchanged
        if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    T tot;
    if (level2_sum_spread(S, m, fs, tot)) {
        if (fuse_x) d->x_pending = 1;          // r of this step is final; its x .+= alpha .* u rides on the next sweep over u
        if (!mik_nrm_in_range(tot)) {
            // |r|^2 underflowed / overflowed (or r is exactly zero): x and r of this step are final, its norm is not.
            // Freeze the batch (later steps become no-ops) and let the host finish the step with the scaled norm.
            d->done = 1;
            mirror->done = 0;
            mirror->nhist = hist_index;
            mirror->range = 1;
            __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        cg_res_scalars(d, tot, mik_sqrt(tot), hist, it_next, maxiter, mirror, seq, hist_index);
    }
}

// closes a step whose residual norm the host computed through the scaled pass
template <typename T>
__global__ void k_cg_fix_res(CgDev<T> *d, T res, T *__restrict__ hist, long long it_next, long long maxiter, CgMirror *mirror,
                             unsigned long long seq, int hist_index, int pcg_fused = 0)
{
    d->done = 0;
    mirror->range = 0;
    cg_res_scalars(d, res * res, res, hist, it_next, maxiter, mirror, seq, hist_index, pcg_fused);
}

// the tail finaliser of the fused PCG step: residual = norm(r) as k_cg_fin_res, and rho = dot(Pl \\ r, r), beta = rho / rho_prev of
// the NEXT step (src/cg.jl:81-85) from the second reduction of the same sweep
template <typename T>
__global__ __launch_bounds__(64) void k_cg_fin_res2(const T *__restrict__ S1, const T *__restrict__ S2, int64_t m, CgDev<T> *d, T *__restrict__ hist,
                                                     long long it_next, long long maxiter, CgMirror *mirror, unsigned long long seq, int hist_index,
                                                     FinScratch2<T> *fs)
{
    if (d->done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    T tot, tot2;
    if (level2_sum_spread2(S1, S2, m, fs, tot, tot2)) {
        d->x_pending = 1;                      // r of this step is final; its x .+= alpha .* u rides on the next sweep over u
        const T rho_prev = d->rho;
        d->rho = tot2;
        d->beta_rho = tot2 / rho_prev;
        if (!mik_nrm_in_range(tot)) {          // as k_cg_fin_res: freeze the batch, the host finishes the step with the scaled norm
            d->done = 1;
            mirror->done = 0;
            mirror->nhist = hist_index;
            mirror->range = 1;
            __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        cg_res_scalars(d, tot, mik_sqrt(tot), hist, it_next, maxiter, mirror, seq, hist_index, 1);
    }
}

static int cg_profile_collect(mik_cg *it)
{
    // called after the host has seen the last tail of a call: every bracketed launch has finished except those of a head
    // enqueued ahead, whose events stay in the list until the next call (or a flush) finds them complete
    size_t i = 0;
    for (; i + 1 < it->ev_used; i += 2) {
        if (hipEventQuery(it->ev[i + 1]) != hipSuccess) { (void)hipGetLastError(); break; }
        float ms = 0.f;
        const int kind = it->ev_kind[i / 2];
        if (hipEventElapsedTime(&ms, it->ev[i], it->ev[i + 1]) == hipSuccess) { it->kern_ms[kind] += ms; it->kern_launches[kind] += 1; }
    }
    size_t keep = 0;
    for (size_t j = i; j + 1 < it->ev_used; j += 2, keep += 2) {
        std::swap(it->ev[keep], it->ev[j]);
        std::swap(it->ev[keep + 1], it->ev[j + 1]);
        it->ev_kind[keep / 2] = it->ev_kind[j / 2];
    }
    it->ev_used = keep;
    return MIK_OK;
}

static void cg_profile_flush(mik_cg *it)
{
    if (!it->ev_used) return;
    (void)hipStreamSynchronize(it->ctx->stream);
    (void)cg_profile_collect(it);
}

// HIP events on the ctx stream around one launch of the step (kind: 0 = SpMV, 1 = xpby, 2 = update)
struct CgProfileScope {
    mik_cg *it; bool on;
    CgProfileScope(mik_cg *it_, int kind) : it(it_), on(it_->profile == 2 || (it_->profile == 1 && kind == 0))
    {
        if (!on) return;
        if (it->ev_used + 2 > it->ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            it->ev.push_back(a); it->ev.push_back(b);
            it->ev_kind.push_back(0);
        }
        it->ev_kind[it->ev_used / 2] = kind;
        (void)hipEventRecord(it->ev[it->ev_used], it->ctx->stream);
    }
    ~CgProfileScope()
    {
        if (!on) return;
        (void)hipEventRecord(it->ev[it->ev_used + 1], it->ctx->stream);
        it->ev_used += 2;
    }
};

// Cache hints of the two vector kernels of a CG step (results unchanged).  x is touched once per iteration, c
// and u are dead / about to be overwritten after the update, and r has just been read for the last time in
// u .= r .+ beta .* u: streaming them past L2 (non-temporal) leaves the cache to the operator's gather and
// took the in-loop SpMV from 322 to 307 us and the step from 546 to 507 us at 256^3.
// bits 0-2: xpby {r load, u load, u store}; bits 3-7: update {x, c load, u load, r load, r store}.
// Masks (found by sweeps of single-bit flips inside the CG loop at 256^3 in round 2): 57 for the classic step; 248 when x .+= alpha .* u rides on the next sweep over u (OpXpbyX /
// OpCgUpdateR: bit 3 = x of that sweep): x, c and both directions of r in the update streamed, r read with the default policy by
// the sweep that follows.  With k_spmv_sdiab2 the r STORE is the bit that matters: streamed, it leaves the Infinity Cache to
// what the SpMV reads (in-loop SpMV 62 -> 48 us = its back-to-back time; 121 -> 249: 4,330 -> 4,520 it/s), and r then read
// cached rather than streamed takes 8 us off the update that wrote it (249 -> 248: 4,720 it/s).
// (With an operator whose SpMV itself streams gigabytes -- CSR, per-row values -- the earlier mask 121 stays: 248 cost the CSR
// loop 10 us per step.)
static inline int cg_stream_hints(const mik_ctx *ctx, bool fused_x = false, const mik_csr *A = nullptr)
{
    (void)ctx;
    return fused_x ? (mik_spmv_is_light(A) ? 248 : 121) : 57;
}

// One iterate() = HEAD (u = r + beta u [after c = Pl \ r, rho]; c = A u; alpha) + TAIL (x, r update; residual, stopping test).
// The head only writes the iterable's internal vectors u and c and scalars, and all of its inputs are final once the previous
// tail has run -- so the head of step k + 1 may be put on the stream BEFORE the host waits for the residual of step k
// (cg_iterate_many_impl): the device never idles while the host reacts.  If step k met the stopping test the head kernels
// early-exit on the device's `done` flag like every later step of a batch.
template <typename T> static int cg_enqueue_head(mik_cg *it)
{
    mik_ctx *ctx = it->ctx;
    CgDev<T> *d = (CgDev<T> *)it->dev;
    const int *done = &d->done;
    const int64_t n = it->n;
    const int64_t nseg = mik_nseg<T>(n);
    const int64_t nb = mik_spmv_nwg(n);
    T *x = (T *)it->x, *u = (T *)it->u, *r = (T *)it->r, *c = (T *)it->c;
    const bool vec = mik_aligned16(x) && mik_aligned16(u) && mik_aligned16(r) && mik_aligned16(c) && (!it->diag || mik_aligned16(it->diag));
    const int pcg = (it->diag || it->pl_fn) ? 1 : 0;
    if (it->pcg_fused) {
        // c = Pl \ r and rho = dot(c, r) (src/cg.jl:79-85) came with the previous tail (cg init for the first step): nothing to do here
    } else if (it->diag) {
        // c = Pl \ r; rho = dot(c, r)                                   src/cg.jl:79-82
        OpJacobiDot<T> pj{r, (const T *)it->diag, c, cg_stream_hints(ctx) != 0};
        MIK_TRY((launch_map<T>(ctx, n, pj, vec, (T *)it->seg_vec, done)));
    } else if (it->pl_fn) {
        // any Pl: ldiv!(c, Pl, r) through the callback, then the dot as its own sweep
        if (it->pl_fn(it->pl_user, c, r) != 0) return mik_fail(ctx, MIK_ERR_CALLBACK, "cg: the preconditioner callback failed");
        OpDot<T> dcr{c, r};
        MIK_TRY((launch_map<T>(ctx, n, dcr, vec, (T *)it->seg_vec, done)));
    }
    if (pcg) {
        if (!it->pcg_fused) {
            hipLaunchKernelGGL((k_cg_fin_rho<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)it->seg_vec, nseg, d, (FinScratch<T> *)it->fin);
            MIK_LAUNCH_CHECK(ctx);
        }
        // u .= c .+ beta .* u                                           src/cg.jl:86
        CgProfileScope ps(it, 1);
        if (it->fuse_x) {
            OpXpbyX<T> op{c, u, x, coef_ptr<T>(&d->beta), coef_ptr<T>(&d->alpha), done, &d->x_pending, (cg_stream_hints(ctx, true, it->A) & 8) | 1};   // c = Pl \\ r is dead after this sweep: streamed
            MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)nullptr, (const int *)nullptr)));
        } else {
            OpXpby<T> op{c, u, coef_ptr<T>(&d->beta), cg_stream_hints(ctx) & 1};   // c = Pl \\ r is dead after this sweep
            MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)nullptr, done)));
        }
    } else {
        // u .= r .+ beta .* u                                           src/cg.jl:50-51
        CgProfileScope ps(it, 1);
        if (it->fuse_x) {   // ... and x .+= alpha .* u of the previous step, on the u this sweep reads anyway (OpXpbyX)
            OpXpbyX<T> op{r, u, x, coef_ptr<T>(&d->beta), coef_ptr<T>(&d->alpha), done, &d->x_pending, cg_stream_hints(ctx, true, it->A) & 15};
            MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)nullptr, (const int *)nullptr)));
        } else {
            OpXpby<T> op{r, u, coef_ptr<T>(&d->beta), cg_stream_hints(ctx) & 7};
            MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)nullptr, done)));
        }
    }
    // c = A * u with the dot(u, c) epilogue                             src/cg.jl:54-55
    if (it->A) {
        {
            CgProfileScope ps(it, 0);
            const int rc_spmv = mik_spmv_launch<T>(ctx, it->A, u, c, true, (T *)it->seg_spmv, done);
            MIK_TRY(rc_spmv);
        }
        hipLaunchKernelGGL((k_cg_fin_alpha<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)it->seg_spmv, nb, d, pcg, (FinScratch<T> *)it->fin);
    } else {
        // any operator: mul!(c, A, u) through the callback; dot(u, c) as its own sweep with the vector tree shape
        if (it->op_mul(it->op_user, u, c) != 0) return mik_fail(ctx, MIK_ERR_CALLBACK, "cg: the operator callback failed");
        OpDot<T> duc{u, c};
        MIK_TRY((launch_map<T>(ctx, n, duc, vec, (T *)it->seg_vec, done)));
        hipLaunchKernelGGL((k_cg_fin_alpha<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)it->seg_vec, nseg, d, pcg, (FinScratch<T> *)it->fin);
    }
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

template <typename T> __global__ void k_cg_clear_pending(CgDev<T> *d) { d->x_pending = 0; }

template <typename T> static int cg_enqueue_xflush(mik_cg *it)
{
    mik_ctx *ctx = it->ctx;
    CgDev<T> *d = (CgDev<T> *)it->dev;
    T *x = (T *)it->x, *u = (T *)it->u;
    OpXFlush<T> op{u, x, coef_ptr<T>(&d->alpha), &d->x_pending};
    MIK_TRY((launch_map<T>(ctx, it->n, op, mik_aligned16(x) && mik_aligned16(u), (T *)nullptr, (const int *)nullptr)));
    hipLaunchKernelGGL((k_cg_clear_pending<T>), dim3(1), dim3(1), 0, ctx->stream, d);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

template <typename T> static int cg_enqueue_tail(mik_cg *it, long long it_next, int hist_index)
{
    mik_ctx *ctx = it->ctx;
    CgDev<T> *d = (CgDev<T> *)it->dev;
    const int *done = &d->done;
    const int64_t n = it->n;
    const int64_t nseg = mik_nseg<T>(n);
    T *x = (T *)it->x, *u = (T *)it->u, *r = (T *)it->r, *c = (T *)it->c;
    const bool vec = mik_aligned16(x) && mik_aligned16(u) && mik_aligned16(r) && mik_aligned16(c) && (!it->diag || mik_aligned16(it->diag));
    // x .+= alpha .* u; r .-= alpha .* c; norm(r)                       src/cg.jl:58-62
    if (it->pcg_fused) {
        {
            CgProfileScope ps(it, 2);
            // r is not read again before the next tail (the head reads c = Pl \\ r, u, x): both directions streamedherwise.
            // c too (bits 5, 6; round 4): c = Pl \\ r stored with the default policy shares the 256 MB Infinity Cache with the u the head writes for the SpMV --
            // streamed, the in-loop SpMV runs at its back-to-back time (73 -> 52 us) and this sweep pays most of it back (99 -> 116 us): 280.7 -> 275.6 us
            // per step, ten words per row + the operator at the copy ceiling (profiles/r04_pcg_kernel_stats.txt).
            OpPcgUpdateR<T> up{r, c, (const T *)it->diag, coef_ptr<T>(&d->alpha), 120};
            MIK_TRY((launch_map2<T>(ctx, n, up, vec, (T *)it->seg_vec, (T *)it->seg_vec2, done)));
        }
        it->seq += 1;
        hipLaunchKernelGGL((k_cg_fin_res2<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)it->seg_vec, (const T *)it->seg_vec2, nseg, d, (T *)it->hist,
                           it_next, (long long)it->maxiter, it->mirror, it->seq, hist_index, (FinScratch2<T> *)((unsigned char *)it->fin + 512));
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    {
        CgProfileScope ps(it, 2);
        if (it->fuse_x) {
            OpCgUpdateR<T> up{r, c, coef_ptr<T>(&d->alpha), cg_stream_hints(ctx, true, it->A) >> 3};
            MIK_TRY((launch_map<T>(ctx, n, up, vec, (T *)it->seg_vec, done)));
        } else {
            OpCgUpdate<T> up{x, r, u, c, coef_ptr<T>(&d->alpha), cg_stream_hints(ctx) >> 3};
            MIK_TRY((launch_map<T>(ctx, n, up, vec, (T *)it->seg_vec, done)));
        }
    }
    it->seq += 1;
    hipLaunchKernelGGL((k_cg_fin_res<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)it->seg_vec, nseg, d, (T *)it->hist,
                       it_next, (long long)it->maxiter, it->mirror, it->seq, hist_index, (FinScratch<T> *)it->fin, it->fuse_x ? 1 : 0);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// Wait until the device has published step `seq` in the host-mapped mirror (bounded spin).
int cg_wait_mirror(mik_cg *it)
{
    volatile unsigned long long *p = &it->mirror->seq;
    const unsigned long long want = it->seq;
    for (unsigned long long spins = 0;; ++spins) {
        if (__atomic_load_n((const unsigned long long *)p, __ATOMIC_ACQUIRE) == want) return MIK_OK;
        if ((spins & 0xFFFFF) == 0xFFFFF) {   // every ~1M polls: has the stream died or finished without publishing?
            hipError_t e = hipStreamQuery(it->ctx->stream);
            if (e == hipSuccess) {
                if (__atomic_load_n((const unsigned long long *)p, __ATOMIC_ACQUIRE) == want) return MIK_OK;
                return mik_fail(it->ctx, MIK_ERR_HIP, "cg: stream idle but step %llu was never published", want);
            }
            if (e != hipErrorNotReady) return mik_fail(it->ctx, MIK_ERR_HIP, "cg: %s while waiting for a step", hipGetErrorString(e));
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

template <typename T> static int cg_fetch_state(mik_cg *it, CgDev<T> *host)
{
    mik_ctx *ctx = it->ctx;
    MIK_HIP(ctx, hipMemcpyAsync(ctx->coef_host, it->dev, sizeof(CgDev<T>), hipMemcpyDeviceToHost, ctx->stream));
    MIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host, ctx->coef_host, sizeof(CgDev<T>));
    return MIK_OK;
}

template <typename T>
static int cg_init_impl(mik_cg *it, double abstol, double reltol, int initially_zero)
{
    mik_ctx *ctx = it->ctx;
    const int64_t n = it->n;
    const int64_t nseg = mik_nseg<T>(n);
    T *x = (T *)it->x, *u = (T *)it->u, *r = (T *)it->r, *c = (T *)it->c;
    const T *b = (const T *)it->b;
    OpFill<T> z{u, T(0)};                                                 // u .= 0          :129
    MIK_TRY((launch_map<T>(ctx, n, z, mik_aligned16(u), (T *)nullptr, nullptr)));
    const bool vec = mik_aligned16(r) && mik_aligned16(b) && mik_aligned16(c);
    if (initially_zero) {
        it->mv_products = 0;                                              // :134
        OpSubNrm<T> op{b, nullptr, r};                                    // copyto!(r, b)   :130
        MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)it->seg_vec, nullptr)));
    } else {
        it->mv_products = 1;                                              // :136
        if (it->A) MIK_TRY(mik_spmv_launch<T>(ctx, it->A, x, c, false, nullptr, nullptr));   // :137
        else if (it->op_mul(it->op_user, x, c) != 0) return mik_fail(ctx, MIK_ERR_CALLBACK, "cg: the operator callback failed");
        OpSubNrm<T> op{b, c, r};                                          // r = b - c       :130,138
        MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)it->seg_vec, nullptr)));
    }
    hipLaunchKernelGGL((k_cg_fin_init<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)it->seg_vec, nseg,
                       (CgDev<T> *)it->dev, (T)reltol, (T)abstol, (long long)it->maxiter);
    MIK_LAUNCH_CHECK(ctx);
    CgDev<T> h;
    MIK_TRY(cg_fetch_state<T>(it, &h));
    if (!mik_nrm_in_range(h.rr)) {
        // badly scaled system (or r = 0): the over-/underflow-safe norm, then the same scalar set-up
        T res;
        MIK_TRY(mik_safe_norm_slow<T>(ctx, n, r, &res));
        hipLaunchKernelGGL((k_cg_set_init<T>), dim3(1), dim3(1), 0, ctx->stream, (CgDev<T> *)it->dev, res, (T)reltol, (T)abstol, (long long)it->maxiter);
        MIK_LAUNCH_CHECK(ctx);
        MIK_TRY(cg_fetch_state<T>(it, &h));
    }
    it->residual = (double)h.res;
    it->prev_residual = (double)h.prev_res;
    it->tol = (double)h.tol;
    it->dev_done = h.done != 0;
    if (it->pcg_fused) {
        // rho of the FIRST step (src/cg.jl:79-85 with rho_prev = one): later steps get theirs from the tail of the step before
        OpJacobiDot<T> pj{r, (const T *)it->diag, c, 0};
        MIK_TRY((launch_map<T>(ctx, n, pj, vec && mik_aligned16(it->diag), (T *)it->seg_vec, (const int *)nullptr)));
        hipLaunchKernelGGL((k_cg_fin_rho<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)it->seg_vec, nseg, (CgDev<T> *)it->dev, (FinScratch<T> *)it->fin);
        MIK_LAUNCH_CHECK(ctx);
    }
    return MIK_OK;
}

static int cg_create_common(mik_ctx *ctx, const mik_csr *A, int dtype, int64_t n, mik_mul_fn op_mul, void *op_user, const void *jacobi_diag,
                            mik_ldiv_fn pl_fn, void *pl_user, void *x, const void *b, void *u, void *r, void *c, double abstol, double reltol,
                            int64_t maxiter, int initially_zero, mik_cg **out)
{
    if (n && (!x || !b || !u || !r || !c)) return mik_fail(ctx, MIK_ERR_INVALID, "mik_cg_create: NULL vector");
    mik_cg *it = new (std::nothrow) mik_cg();
    if (!it) return mik_fail(ctx, MIK_ERR_NOMEM, "mik_cg_create: host allocation failed");
    it->ctx = ctx; it->A = A; it->dtype = dtype; it->n = n;
    it->op_mul = op_mul; it->op_user = op_user; it->pl_fn = pl_fn; it->pl_user = pl_user;
    it->x = x; it->b = b; it->u = u; it->r = r; it->c = c; it->diag = jacobi_diag;
    it->maxiter = maxiter;
    it->fuse_x = A != nullptr && !pl_fn && (ctx->tuning[MIK_KNOB_CG_STEP] & 1) == 0;       // development knob MIK_KNOB_CG_STEP bit 0: x updated by the step's own sweep
    it->pcg_fused = it->fuse_x && jacobi_diag != nullptr && (ctx->tuning[MIK_KNOB_CG_STEP] & 2) == 0;   // bit 1: the three-sweep PCG step
    const size_t es = mik_dtype_size(dtype);
    const int64_t nseg = dtype == MIK_F64 ? mik_nseg<double>(n) : mik_nseg<float>(n);
    const int64_t nb = mik_spmv_nwg(n);
    hipError_t e;
    (void)hipSetDevice(ctx->device);
    if ((e = hipMalloc(&it->dev, 256)) != hipSuccess || (e = hipMalloc(&it->fin, 1024)) != hipSuccess ||       // [0, 512): FinScratch; [512, 1024): FinScratch2
        (e = hipMemsetAsync(it->fin, 0, 1024, ctx->stream)) != hipSuccess ||
        (it->pcg_fused && (e = hipMalloc(&it->seg_vec2, es * (size_t)std::max<int64_t>(nseg, 1))) != hipSuccess) || (e = hipMalloc(&it->seg_spmv, es * (size_t)std::max<int64_t>(nb, 1))) != hipSuccess ||
        (e = hipMalloc(&it->seg_vec, es * (size_t)std::max<int64_t>(nseg, 1))) != hipSuccess ||
        (e = hipMalloc(&it->hist, es * 64)) != hipSuccess) {
        mik_cg_destroy(it);
        return mik_fail(ctx, MIK_ERR_NOMEM, "mik_cg_create: hipMalloc: %s", hipGetErrorString(e));
    }
    it->hist_cap = 64;
    if ((e = hipHostMalloc((void **)&it->mirror, sizeof(CgMirror), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) {
        mik_cg_destroy(it);
        return mik_fail(ctx, MIK_ERR_NOMEM, "mik_cg_create: hipHostMalloc: %s", hipGetErrorString(e));
    }
    memset(it->mirror, 0, sizeof(CgMirror));
    int rc = dtype == MIK_F64 ? cg_init_impl<double>(it, abstol, reltol, initially_zero)
                              : cg_init_impl<float>(it, abstol, reltol, initially_zero);
    if (rc) { mik_cg_destroy(it); return rc; }
    *out = it;
    return MIK_OK;
}

extern "C" int mik_cg_create(mik_ctx *ctx, const mik_csr *A, void *x, const void *b, void *u, void *r, void *c,
                             const void *jacobi_diag, double abstol, double reltol, int64_t maxiter, int initially_zero,
                             mik_cg **out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    if (!A || A->n_rows != A->n_cols) return mik_fail(ctx, MIK_ERR_MISMATCH, "mik_cg_create: A must be square");
    return cg_create_common(ctx, A, A->dtype, A->n_rows, nullptr, nullptr, jacobi_diag, nullptr, nullptr, x, b, u, r, c, abstol, reltol, maxiter,
                            initially_zero, out);
}

// the operator of a *_create_op call: a CSR handle or a callback
static int resolve_operator(mik_ctx *ctx, const mik_operator *A, const char *who, const mik_csr **csr, int *dtype, int64_t *n)
{
    if (!A || (!A->csr && !A->mul)) return mik_fail(ctx, MIK_ERR_INVALID, "%s: the operator needs a csr handle or a mul callback", who);
    if (A->csr) {
        if (A->csr->n_rows != A->csr->n_cols) return mik_fail(ctx, MIK_ERR_MISMATCH, "%s: A must be square", who);
        *csr = A->csr; *dtype = A->csr->dtype; *n = A->csr->n_rows;
        return MIK_OK;
    }
    if ((A->dtype != MIK_F64 && A->dtype != MIK_F32) || A->n < 0) return mik_fail(ctx, MIK_ERR_INVALID, "%s: bad dtype / size of the callback operator", who);
    *csr = nullptr; *dtype = A->dtype; *n = A->n;
    return MIK_OK;
}

extern "C" int mik_cg_create_op(mik_ctx *ctx, const mik_operator *A, const mik_precond *Pl, void *x, const void *b, void *u, void *r, void *c,
                                double abstol, double reltol, int64_t maxiter, int initially_zero, mik_cg **out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    const mik_csr *csr = nullptr;
    int dtype = MIK_F64;
    int64_t n = 0;
    MIK_TRY(resolve_operator(ctx, A, "mik_cg_create_op", &csr, &dtype, &n));
    if (Pl && Pl->diag && Pl->ldiv) return mik_fail(ctx, MIK_ERR_INVALID, "mik_cg_create_op: Pl has both a diagonal and a callback");
    return cg_create_common(ctx, csr, dtype, n, csr ? nullptr : A->mul, A->user, Pl ? Pl->diag : nullptr, Pl ? Pl->ldiv : nullptr,
                            Pl ? Pl->user : nullptr, x, b, u, r, c, abstol, reltol, maxiter, initially_zero, out);
}

extern "C" int mik_cg_destroy(mik_cg *it)
{
    if (!it) return MIK_OK;
    if (it->ctx) (void)hipStreamSynchronize(it->ctx->stream);
    if (it->dev) (void)hipFree(it->dev);
    if (it->fin) (void)hipFree(it->fin);
    if (it->hist) (void)hipFree(it->hist);
    if (it->seg_spmv) (void)hipFree(it->seg_spmv);
    if (it->seg_vec) (void)hipFree(it->seg_vec);
    if (it->seg_vec2) (void)hipFree(it->seg_vec2);
    for (hipEvent_t e : it->ev) (void)hipEventDestroy(e);
    if (it->mirror) (void)hipHostFree(it->mirror);
    delete it;
    return MIK_OK;
}

template <typename T>
static int cg_iterate_many_impl(mik_cg *it, int64_t iteration, int64_t max_steps, double *residuals, int64_t *steps_done)
{
    mik_ctx *ctx = it->ctx;
    *steps_done = 0;
    // done(it, iteration)                                               src/cg.jl:36
    if (max_steps <= 0 || iteration >= it->maxiter || it->residual <= it->tol) return MIK_OK;
    max_steps = std::min(max_steps, it->maxiter - iteration);
    if ((it->op_mul || it->pl_fn) && max_steps > 1) {
        // Host callbacks run when a step is ENQUEUED: a batch would call mul! / ldiv! for steps that the device-side stopping
        // test later turns into no-ops, and the caller may count those calls.  One host wait per step, like the reference's loop.
        int64_t total = 0;
        for (int64_t j = 0; j < max_steps; ++j) {
            int64_t nd1 = 0;
            double res1 = 0;
            MIK_TRY(cg_iterate_many_impl<T>(it, iteration + j, 1, &res1, &nd1));
            if (nd1 == 0) break;
            if (residuals) residuals[total] = res1;
            total += 1;
        }
        *steps_done = total;
        return MIK_OK;
    }
    if (max_steps > it->hist_cap) {
        MIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MIK_HIP(ctx, hipFree(it->hist));
        it->hist = nullptr;
        MIK_HIP(ctx, hipMalloc(&it->hist, sizeof(T) * (size_t)max_steps));
        it->hist_cap = max_steps;
    }
    CgDev<T> *d = (CgDev<T> *)it->dev;
    // The device is idle here (the previous call waited for its last step), so the host may reset the
    // step counter in the host-mapped mirror directly.  The stopping flag only needs clearing if a previous
    // call left it set while the host test above said "not done" (e.g. the caller restarted the count).
    it->mirror->nhist = 0;
    it->mirror->range = 0;
    if (it->dev_done) MIK_HIP(ctx, hipMemsetAsync(&d->done, 0, sizeof(int), ctx->stream));
    CgMirror m;
    // the head of the step AFTER this call goes on the stream before the host waits (never with host callbacks, whose call
    // count the caller may observe; MIK_KNOB_NO_LOOKAHEAD: off)
    const bool ahead_ok = ctx->tuning[MIK_KNOB_NO_LOOKAHEAD] == 0 && !it->op_mul && !it->pl_fn && iteration + max_steps < it->maxiter;
    for (int64_t j0 = 0;;) {
        for (int64_t j = j0; j < max_steps; ++j) {
            if (!it->head_ahead) MIK_TRY(cg_enqueue_head<T>(it));
            it->head_ahead = false;
            MIK_TRY(cg_enqueue_tail<T>(it, (long long)(iteration + j + 1), (int)j));
        }
        if (ahead_ok) MIK_TRY(cg_enqueue_head<T>(it));
        else if (it->fuse_x) MIK_TRY(cg_enqueue_xflush<T>(it));         // no sweep over u follows: apply the last x .+= alpha .* u now
        MIK_TRY(cg_wait_mirror(it));
        m = *it->mirror;
        if (!m.range) { it->head_ahead = ahead_ok && !m.done; break; }       // stopped: the head ahead was a no-op
        it->head_ahead = false;                                              // frozen batch: so was everything behind the frozen step
        // Step m.nhist of this call updated x and r, but |r|^2 left the range in which sqrt(sum of squares) is safe
        // (include/mik.h "Norms"): the device froze the batch; finish that step with the scaled norm and go on.
        T res;
        MIK_TRY(mik_safe_norm_slow<T>(ctx, it->n, (const T *)it->r, &res));
        it->seq += 1;
        hipLaunchKernelGGL((k_cg_fix_res<T>), dim3(1), dim3(1), 0, ctx->stream, d, res, (T *)it->hist, (long long)(iteration + m.nhist + 1),
                           (long long)it->maxiter, it->mirror, it->seq, (int)m.nhist, it->pcg_fused ? 1 : 0);
        MIK_LAUNCH_CHECK(ctx);
        MIK_TRY(cg_wait_mirror(it));
        m = *it->mirror;
        j0 = m.nhist;
        if (m.done || j0 >= max_steps) break;
    }
    const int64_t nd = m.nhist;
    if (nd == 1) {
        if (residuals) residuals[0] = m.res;
    } else if (nd > 1) {
        std::vector<T> tmp((size_t)nd);
        MIK_HIP(ctx, hipMemcpyAsync(tmp.data(), it->hist, sizeof(T) * (size_t)nd, hipMemcpyDeviceToHost, ctx->stream));
        MIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (residuals) for (int64_t j = 0; j < nd; ++j) residuals[j] = (double)tmp[j];
    }
    if (nd > 0) {
        it->residual = m.res;
        it->prev_residual = m.prev_res;
    }
    it->dev_done = m.done != 0;
    it->mv_products += nd;
    *steps_done = nd;
    if (it->profile) cg_profile_collect(it);
    return MIK_OK;
}

extern "C" int mik_cg_iterate_many(mik_cg *it, int64_t iteration, int64_t max_steps, double *residuals, int64_t *steps_done)
{
    if (!it || !steps_done || iteration < 0) return MIK_ERR_INVALID;
    return it->dtype == MIK_F64 ? cg_iterate_many_impl<double>(it, iteration, max_steps, residuals, steps_done)
                                : cg_iterate_many_impl<float>(it, iteration, max_steps, residuals, steps_done);
}

extern "C" int mik_cg_iterate(mik_cg *it, int64_t iteration, double *residual, int *done)
{
    if (!it || !done) return MIK_ERR_INVALID;
    int64_t nd = 0;
    double res = 0;
    int rc = mik_cg_iterate_many(it, iteration, 1, &res, &nd);
    if (rc) return rc;
    *done = nd == 0 ? 1 : 0;
    if (residual) *residual = it->residual;
    return MIK_OK;
}

extern "C" int mik_cg_fused_x(const mik_cg *it, int *fused)
{
    if (!it || !fused) return MIK_ERR_INVALID;
    *fused = it->fuse_x ? 1 : 0;
    return MIK_OK;
}

extern "C" int mik_cg_profile(mik_cg *it, int enable, double *spmv_ms_total, int64_t *spmv_launches)
{
    if (!it) return MIK_ERR_INVALID;
    cg_profile_flush(it);
    if (spmv_ms_total) *spmv_ms_total = it->kern_ms[0];
    if (spmv_launches) *spmv_launches = it->kern_launches[0];
    if (enable >= 0) {
        it->profile = enable;
        if (enable) { for (int q = 0; q < 3; ++q) { it->kern_ms[q] = 0; it->kern_launches[q] = 0; } it->ev_used = 0; }
    }
    return MIK_OK;
}

// The same for the row-partitioned iterable: HIP events on the compute stream around every SpMV launch of its steps (one per step with
// flag-ordered halos; the interior and the boundary launch when the halo is ordered by events and an interior range is set).
extern "C" int mik_cgd_profile(mik_cgd *it, int enable, double *spmv_ms_total, int64_t *spmv_launches)
{
    if (!it) return MIK_ERR_INVALID;
    return mik_cg_profile(&it->base, enable > 1 ? 1 : enable, spmv_ms_total, spmv_launches);
}

extern "C" int mik_cg_profile_kernels(const mik_cg *it, double *ms_total, int64_t *launches)
{
    if (!it) return MIK_ERR_INVALID;
    cg_profile_flush(const_cast<mik_cg *>(it));
    for (int q = 0; q < 3; ++q) {
        if (ms_total) ms_total[q] = it->kern_ms[q];
        if (launches) launches[q] = it->kern_launches[q];
    }
    return MIK_OK;
}

extern "C" int mik_cg_state(const mik_cg *it, double *residual, double *prev_residual, double *tol, int64_t *maxiter,
                            int64_t *mv_products, int *converged)
{
    if (!it) return MIK_ERR_INVALID;
    if (residual) *residual = it->residual;
    if (prev_residual) *prev_residual = it->prev_residual;
    if (tol) *tol = it->tol;
    if (maxiter) *maxiter = it->maxiter;
    if (mv_products) *mv_products = it->mv_products;
    if (converged) *converged = it->residual <= it->tol ? 1 : 0;   // src/cg.jl:32
    return MIK_OK;
}

// =============================================================================================
// GMRESIterable
// =============================================================================================
struct mik_gmres {
    mik_ctx *ctx = nullptr;
    const mik_csr *A = nullptr;
    int dtype = MIK_F64;
    int64_t n = 0, ldv = 0;
    void *x = nullptr;
    const void *b = nullptr;
    const void *pl = nullptr, *pr = nullptr;   // diagonal left / right preconditioners (device n-vectors) or NULL = Identity()
    mik_mul_fn op_mul = nullptr;                // A as a callback (A == nullptr): mik_gmres_create_op
    void *op_user = nullptr;
    mik_ldiv_fn pl_fn = nullptr, pr_fn = nullptr;   // Pl / Pr as callbacks
    void *pl_user = nullptr, *pr_user = nullptr;
    void *V = nullptr;        // device n x (restart + 1), column-major   src/gmres.jl:13
    void *Ax = nullptr;       // device work vector                        src/gmres.jl:125
    std::vector<double> H64; std::vector<float> H32;             // (restart+1) x restart   :14
    std::vector<double> nv64; std::vector<float> nv32;           // Residual.nullvec        :27
    double current = 1, accumulator = 1, res_beta = 1;           // Residual                :24-29
    double g_beta = 1, tol = 0;
    int k = 1, restart = 0, method = MIK_MGS;
    int64_t maxiter = 0, mv_products = 0;
    bool dist = false;        // row-partitioned: SpMV input goes through part.x_ext + halo(), sums through reduce()
    mik_partition part{};
    // single-launch Modified Gram-Schmidt (k_mgs_fused): slot buffers [2][restart + 1][256] and the host-mapped mirror of (h, nrm)
    void *mgs_P = nullptr;
    int mgs_G = 1, mgs_stride = 256;  // segments per workgroup of the single-launch kernels; slots per row of mgs_P
    int mgs_res_S = 0;               // > 0: Modified Gram-Schmidt runs in the resident-w form (k_mgs_resident) with this many segments per workgroup
    int gs_timeouts = 0;             // how often a single-launch column came back timed out (mik_dev_gmres_form)
    bool fused_off = false;          // latched when the single-launch kernel's bounded spin expired once: multi-launch chains from then on
    unsigned *xl_chk = nullptr;      // device, 2 words: XCC id + 1 of the participants of the XCD-local form (k_mgs_fused XL), per parity
    bool xl_off = false;             // the XCD-local form failed once (its workgroups were not on one XCD, or a wait expired): all-XCD form from then on
    bool xl_last = false;            // the column last enqueued used the XCD-local form
    int mgs_rounds = 1;              // DGKS rounds the single-launch kernel runs before it hands back to the host loop
    MgsMirror *mgs_mirror = nullptr;          // two mirrors (bytes apart: mgs_mirror_stride), used alternately
    size_t mgs_mirror_stride = 0;
    unsigned long long mgs_seq = 0, mgs_slot_seq[2] = {0, 0};
    int mgs_parity = 0;
    int pre_k = 0, pre_slot = 0;              // Arnoldi column already enqueued ahead of the host (0 = none) and its mirror
};

template <typename T> static int gather_launch(mik_ctx *ctx, int64_t m, const int *idx, const T *x, T *out, const int *done);

// mul!(dst, A, src) on this rank's rows; with a partition, src is staged in x_ext and the halo callback
// fills its ghost tail before the local block is applied.                src/gmres.jl:245,287,293,301
template <typename T> static int gm_spmv(mik_gmres *g, const T *src, T *dst)
{
    mik_ctx *ctx = g->ctx;
    if (g->op_mul) {                                    // any operator: mul!(dst, A, src) through the callback
        if (g->op_mul(g->op_user, src, dst) != 0) return mik_fail(ctx, MIK_ERR_CALLBACK, "gmres: the operator callback failed");
        return MIK_OK;
    }
    if (!g->dist) return mik_spmv_launch<T>(ctx, g->A, src, dst, false, nullptr, nullptr);
    T *ext = (T *)g->part.x_ext;
    if (src != ext && g->n > 0) MIK_HIP(ctx, hipMemcpyAsync(ext, src, sizeof(T) * (size_t)g->n, hipMemcpyDeviceToDevice, ctx->stream));
    if (g->part.n_send > 0) MIK_TRY(gather_launch<T>(ctx, g->part.n_send, g->part.send_idx, ext, (T *)g->part.send_buf, nullptr));
    if (g->part.link) MIK_TRY(plink_halo(g->part.link, g->part.send_buf, ext + g->n));     // device-driven: push, land -- no host in between
    else if (g->part.halo && g->part.halo(g->part.user) != 0) return mik_fail(ctx, MIK_ERR_CALLBACK, "gmres: halo callback failed");
    return mik_spmv_launch<T>(ctx, g->A, ext, dst, false, nullptr, nullptr);
}

// values[0..count): this rank's partial sums -> sums over ranks in rank order (identity without a partition)
template <typename T> static int gm_reduce(mik_gmres *g, T *values, int count)
{
    if (!g->dist || !g->part.reduce || count <= 0) return MIK_OK;
    if (g->part.reduce(g->part.user, g->dtype, count, values) != 0) return mik_fail(g->ctx, MIK_ERR_CALLBACK, "gmres: reduce callback failed");
    return MIK_OK;
}

// Row-partitioned norm(x) from ss = the rank-ordered sum of the local sums of squares (identical on every rank, so every rank
// takes the same branch): sqrt(ss) inside the safe range; outside (0 included: every square may have underflowed), the over-/underflow-safe
// recomputation of include/mik.h "Norms" across the ranks -- amax = max over the ranks of the local max |x_i|, obtained through
// the SUM callback by letting every rank contribute its value in its own slot of a P-vector of zeros (0 + ... + a_q + 0 is
// exact); s = 2^-exponent(amax), the same power of two everywhere; t' = the rank-ordered sum of the local tree sums of
// (x_i s)^2; norm = sqrt(t') / s.  oracle/orc_impl.inc safe_nrm_ with a partition is this arithmetic.
template <typename T> static int gm_part_norm(mik_gmres *g, const T *x, T ss, T *nrm)
{
    if (mik_nrm_in_range(ss)) { *nrm = std::sqrt(ss); return MIK_OK; }       // (an exact 0 may be an underflow: it takes the scaled pass too)
    mik_ctx *ctx = g->ctx;
    const int64_t n = g->n, nseg = mik_nseg<T>(n);
    const int P = g->part.nranks, rank = g->part.rank;
    if (P < 1 || P > 256 || rank < 0 || rank >= P) return mik_fail(ctx, MIK_ERR_RANGE, "gmres (row-partitioned): scaled norm needs 1 <= nranks <= 256");
    T *scr = (T *)((unsigned char *)ctx->coef + mik_ctx::COEF_SAFE_SLOT);
    const int grid = (int)std::min<int64_t>((std::max<int64_t>(n, 1) + MIK_BLOCK - 1) / MIK_BLOCK, 1024);
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(std::max<int64_t>(nseg, grid), 1)));
    hipLaunchKernelGGL((k_amax<T>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, x, (T *)ctx->partials);
    MIK_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL((k_amax<T>), dim3(1), dim3(MIK_BLOCK), 0, ctx->stream, (int64_t)grid, (const T *)ctx->partials, scr);
    MIK_LAUNCH_CHECK(ctx);
    std::vector<T> v((size_t)P, T(0));
    MIK_HIP(ctx, hipMemcpyAsync(&v[(size_t)rank], scr, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    MIK_HIP(ctx, mik_wait(ctx));
    MIK_TRY(gm_reduce<T>(g, v.data(), P));
    T amax = T(0);
    for (T a : v) amax = (a > amax || a != a) ? a : amax;
    if (amax == T(0) || amax != amax || amax > std::numeric_limits<T>::max()) { *nrm = amax; return MIK_OK; }
    int e;
    (void)std::frexp((double)amax, &e);
    e = std::max(-NrmRange<T>::EC, std::min(NrmRange<T>::EC, e));
    const T sc = (T)std::ldexp(1.0, -e), sinv = (T)std::ldexp(1.0, e);
    OpScaledSq<T> op{x, sc};
    MIK_TRY((launch_map<T>(ctx, n, op, mik_aligned16(x), (T *)ctx->partials, nullptr)));
    hipLaunchKernelGGL((k_finalize_store<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, nseg, (int64_t)0, scr, (const int *)nullptr);
    MIK_LAUNCH_CHECK(ctx);
    T t2;
    MIK_HIP(ctx, hipMemcpyAsync(&t2, scr, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    MIK_HIP(ctx, mik_wait(ctx));
    MIK_TRY(gm_reduce<T>(g, &t2, 1));
    *nrm = (T)std::sqrt(t2) * sinv;
    return MIK_OK;
}

// orthogonalize_and_normalize! over a row partition (src/orthogonalize.jl:13-79): the same sweeps as
// orthogonalize_impl on the local rows; every projection / norm^2 is finalised to a host scalar, summed
// over the ranks by the caller's reduce() and fed back as a kernel argument.
template <typename T>
static int orthogonalize_part(mik_gmres *g, int k, const T *V, int64_t ldv, T *w, T *h, T *nrm_out, int method)
{
    mik_ctx *ctx = g->ctx;
    const int64_t n = g->n;
    if ((size_t)(2 * k + 4) * sizeof(T) > mik_ctx::COEF_SAFE_SLOT) return mik_fail(ctx, MIK_ERR_NOTIMPL, "orthogonalize: k = %d too large", k);
    const int64_t nseg = mik_nseg<T>(n);
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(nseg, 1) * (size_t)std::max(k, 1)));
    T *hd = (T *)ctx->coef;
    T *part = (T *)ctx->partials;
    const bool vecw = mik_aligned16(w);
    const bool vec = vecw && mik_aligned16(V) && (ldv % VT<T>::W == 0);
    auto fetch = [&](int slot, T *dst) -> int {     // level 2 of the local tree -> host -> sum over ranks
        MIK_TRY(finalize_store<T>(ctx, nseg, 1, hd + slot));
        MIK_TRY(coef_download<T>(ctx, slot, dst, 1));
        return gm_reduce<T>(g, dst, 1);
    };
    OpDot<T> dn{w, w};
    T ss = T(0);
    if (method == MIK_MGS) {                                             // :69-76
        if (k > 0) {
            OpDot<T> d0{V, w};
            MIK_TRY((launch_map<T>(ctx, n, d0, vec, part, nullptr)));
            MIK_TRY(fetch(0, &h[0]));
            for (int i = 0; i + 1 < k; ++i) {
                OpMgsPass<T, false> op{w, V + (int64_t)i * ldv, V + (int64_t)(i + 1) * ldv, coef_val<T>(h[i])};
                MIK_TRY((launch_map<T>(ctx, n, op, vec, part, nullptr)));
                MIK_TRY(fetch(0, &h[i + 1]));
            }
            OpMgsPass<T, true> last{w, V + (int64_t)(k - 1) * ldv, nullptr, coef_val<T>(h[k - 1])};
            MIK_TRY((launch_map<T>(ctx, n, last, vec, part, nullptr)));
        } else {
            MIK_TRY((launch_map<T>(ctx, n, dn, vecw, part, nullptr)));
        }
        MIK_TRY(fetch(0, &ss));
    } else {                                                             // :15-17 / :43-45
        auto project = [&](int slot, T *coef) -> int {   // coef = V' w over all ranks; w -= V coef
            if (k == 0) return MIK_OK;
            MIK_TRY(multidot<T>(ctx, n, k, V, ldv, w, hd + slot));
            MIK_TRY(coef_download<T>(ctx, slot, coef, k));
            MIK_TRY(gm_reduce<T>(g, coef, k));
            MIK_TRY(coef_upload<T>(ctx, slot, coef, k));
            return gemv_n_dev<T>(ctx, n, k, V, ldv, hd + slot, T(-1), w);
        };
        MIK_TRY(project(0, h));
        MIK_TRY((launch_map<T>(ctx, n, dn, vecw, part, nullptr)));
        MIK_TRY(fetch(k, &ss));
        if (method == MIK_DGKS) {
            std::vector<T> corr((size_t)std::max(k, 1));
            auto small_norm = [](const T *v, int len) { T s = T(0); for (int j = 0; j < len; ++j) { T p = v[j] * v[j]; s = s + p; } return (T)std::sqrt(s); };
            const T eta = T(1) / std::sqrt(T(2));                        // :20
            T nrm;
            MIK_TRY(gm_part_norm<T>(g, w, ss, &nrm));
            T projection_size = small_norm(h, k);                        // :22
            while (nrm < eta * projection_size) {                        // :26
                MIK_TRY(project(k + 2, corr.data()));                    // :27, :30
                MIK_TRY((launch_map<T>(ctx, n, dn, vecw, part, nullptr)));
                MIK_TRY(fetch(k, &ss));                                  // :32
                MIK_TRY(gm_part_norm<T>(g, w, ss, &nrm));
                projection_size = small_norm(corr.data(), k);            // :28
                for (int j = 0; j < k; ++j) h[j] = h[j] + corr[j];       // :31
            }
        }
    }
    T nrm;
    MIK_TRY(gm_part_norm<T>(g, w, ss, &nrm));
    const T inv = T(1) / nrm;
    OpScal<T> sc{w, coef_val<T>(inv)};                                   // w .*= inv(nrm)  :76 / :48 / :36
    MIK_TRY((launch_map<T>(ctx, n, sc, vecw, (T *)nullptr, nullptr)));
    *nrm_out = nrm;
    return MIK_OK;
}


// ---- the same over a DEVICE-DRIVEN link (mik_partition.link; csrc/mik_comm.hip "mik_plink") -------------------------------------------
// Every reduction is finalised AND summed over the ranks by one kernel (plink_fin_sum: level 2 of the local tree, the rank totals through the
// peer-mapped mailboxes, added in rank order -- the additions of gm_reduce's callback, the same bits); the sweeps take their coefficients from
// device memory, so the k + 1 dependent reductions of an Arnoldi column run back to back on the stream and the host waits ONCE per inner
// iteration, for the column of H (src/orthogonalize.jl:69-76).

// norm(x) over the partition when the plain sum of squares left the safe range (the kernels flagged it with NaN): gm_part_norm's scaled
// recomputation with the exchanges on the device
template <typename T> static int gm_link_norm_slow(mik_gmres *g, const T *x, T *nrm)
{
    mik_ctx *ctx = g->ctx;
    mik_plink *pl = g->part.link;
    const int64_t n = g->n, nseg = mik_nseg<T>(n);
    const int P = plink_nranks(pl), rank = plink_rank(pl);
    T *scr = (T *)((unsigned char *)ctx->coef + mik_ctx::COEF_SAFE_SLOT);          // P scalars (P <= 64: the mailbox's limit)
    if ((size_t)(P + 2) * sizeof(T) > mik_ctx::COEF_SAFE_SLOT) return mik_fail(ctx, MIK_ERR_NOTIMPL, "gmres (row-partitioned): scaled norm with %d ranks", P);
    const int grid = (int)std::min<int64_t>((std::max<int64_t>(n, 1) + MIK_BLOCK - 1) / MIK_BLOCK, 1024);
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(std::max<int64_t>(nseg, grid), 1)));
    hipLaunchKernelGGL((k_amax<T>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, x, (T *)ctx->partials);
    MIK_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL((k_amax<T>), dim3(1), dim3(MIK_BLOCK), 0, ctx->stream, (int64_t)grid, (const T *)ctx->partials, scr + rank);
    MIK_LAUNCH_CHECK(ctx);
    MIK_TRY(plink_gather(pl, scr));
    std::vector<T> v((size_t)P, T(0));
    MIK_HIP(ctx, hipMemcpyAsync(v.data(), scr, sizeof(T) * (size_t)P, hipMemcpyDeviceToHost, ctx->stream));
    MIK_HIP(ctx, mik_wait(ctx));
    MIK_TRY(plink_check(pl, "gmres (row-partitioned)"));
    T amax = T(0);
    for (T a : v) amax = (a > amax || a != a) ? a : amax;
    if (amax == T(0) || amax != amax || amax > std::numeric_limits<T>::max()) { *nrm = amax; return MIK_OK; }
    int e;
    (void)std::frexp((double)amax, &e);
    e = std::max(-NrmRange<T>::EC, std::min(NrmRange<T>::EC, e));
    const T sc = (T)std::ldexp(1.0, -e), sinv = (T)std::ldexp(1.0, e);
    OpScaledSq<T> op{x, sc};
    MIK_TRY((launch_map<T>(ctx, n, op, mik_aligned16(x), (T *)ctx->partials, nullptr)));
    MIK_TRY(plink_fin_sum(pl, ctx->partials, nseg, scr, 0));
    T t2;
    MIK_HIP(ctx, hipMemcpyAsync(&t2, scr, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    MIK_HIP(ctx, mik_wait(ctx));
    MIK_TRY(plink_check(pl, "gmres (row-partitioned)"));
    *nrm = (T)std::sqrt(t2) * sinv;
    return MIK_OK;
}

// The DGKS loop (src/orthogonalize.jl:26-33) over a link, from a first projection on: w unscaled on entry, scaled on exit; every round = batched dot,
// one exchange of the k sums, the update, the norm over the ranks, ONE host wait.  Entered by the chain below and by the single launch's hand-back.
template <typename T>
static int dgks_link_loop(mik_gmres *g, int k, const T *V, int64_t ldv, T *w, T *h, T *nrm_io, T projection_size)
{
    mik_ctx *ctx = g->ctx;
    mik_plink *pl = g->part.link;
    const int64_t n = g->n, nseg = mik_nseg<T>(n);
    if ((size_t)(2 * k + 4) * sizeof(T) > mik_ctx::COEF_SAFE_SLOT) return mik_fail(ctx, MIK_ERR_NOTIMPL, "orthogonalize: k = %d too large", k);
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(nseg, 1) * (size_t)std::max(k, 1)));
    T *hd = (T *)ctx->coef, *part = (T *)ctx->partials;          // [k] nrm, [k + 1] 1 / nrm, [k + 2, 2k + 2) the round's correction
    const bool vecw = mik_aligned16(w);
    OpDot<T> dn{w, w};
    std::vector<T> corr((size_t)std::max(k, 1));
    const T eta = T(1) / std::sqrt(T(2));                                // :20
    T nrm = *nrm_io;
    while (nrm < eta * projection_size) {                                // :26
        if (k > 0) {                                                     // :27, :30
            MIK_TRY(multidot<T>(ctx, n, k, V, ldv, w, hd + k + 2));
            MIK_TRY(plink_sum_vec(pl, hd + k + 2, k));
            MIK_TRY(gemv_n_dev<T>(ctx, n, k, V, ldv, hd + k + 2, T(-1), w));
        }
        MIK_TRY((launch_map<T>(ctx, n, dn, vecw, part, nullptr)));       // :32
        MIK_TRY(plink_fin_sum(pl, part, nseg, hd + k, 1));
        MIK_TRY(coef_download<T>(ctx, k + 2, corr.data(), k));
        MIK_TRY(coef_download<T>(ctx, k, &nrm, 1));
        MIK_TRY(plink_check(pl, "gmres (row-partitioned)"));
        if (nrm != nrm) MIK_TRY(gm_link_norm_slow<T>(g, w, &nrm));
        projection_size = dgks_small_norm<T>(corr.data(), k);            // :28
        for (int j = 0; j < k; ++j) h[j] = h[j] + corr[(size_t)j];       // :31
    }
    OpScal<T> sc{w, coef_val<T>(T(1) / nrm)};                            // :36
    MIK_TRY((launch_map<T>(ctx, n, sc, vecw, (T *)nullptr, nullptr)));
    *nrm_io = nrm;
    return MIK_OK;
}

template <typename T>
static int orthogonalize_link(mik_gmres *g, int k, const T *V, int64_t ldv, T *w, T *h, T *nrm_out, int method)
{
    mik_ctx *ctx = g->ctx;
    mik_plink *pl = g->part.link;
    const int64_t n = g->n;
    if ((size_t)(2 * k + 4) * sizeof(T) > mik_ctx::COEF_SAFE_SLOT) return mik_fail(ctx, MIK_ERR_NOTIMPL, "orthogonalize: k = %d too large", k);
    const int64_t nseg = mik_nseg<T>(n);
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(nseg, 1) * (size_t)std::max(k, 1)));
    T *hd = (T *)ctx->coef;                    // [0, k) h, [k] nrm, [k + 1] 1 / nrm, [k + 2, 2k + 2) DGKS correction
    T *part = (T *)ctx->partials;
    const bool vecw = mik_aligned16(w);
    const bool vec = vecw && mik_aligned16(V) && (ldv % VT<T>::W == 0);
    OpDot<T> dn{w, w};
    std::vector<T> out((size_t)k + 2);
    auto download = [&](int slot, T *dst, int cnt) -> int {      // the one host wait of the column (DGKS: of the round)
        MIK_TRY(coef_download<T>(ctx, slot, dst, cnt));
        return plink_check(pl, "gmres (row-partitioned)");
    };
    T nrm;
    if (method == MIK_MGS && nseg <= mik_resident_cap(ctx) && ctx->tuning[MIK_KNOB_GS] != 1) {
        // the launch-lean chain: every pass finalises AND exchanges the previous reduction itself (k + 2 launches; csrc/mik_comm.hip plink_mgs_lean).
        // At most one segment per compute unit (256 on an unpartitioned MI355X): every workgroup of a pass spins on the mailbox, and ranks that share a GPU must all be resident on it.
        MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * 2048));
        MIK_TRY(plink_mgs_lean(pl, n, k, V, ldv, w, hd, ctx->partials, vec, vecw, mik_mgs_pass_hints(ctx, n, sizeof(T))));
        MIK_TRY(download(0, out.data(), k + 1));
        for (int j = 0; j < k; ++j) h[j] = out[(size_t)j];
        nrm = out[(size_t)k];
        if (nrm != nrm) {                                                // w was left unscaled
            MIK_TRY(gm_link_norm_slow<T>(g, w, &nrm));
            OpScal<T> sc2{w, coef_val<T>(T(1) / nrm)};
            MIK_TRY((launch_map<T>(ctx, n, sc2, vecw, (T *)nullptr, nullptr)));
        }
        *nrm_out = nrm;
        return MIK_OK;
    }
    if (method == MIK_MGS) {                                             // :69-76
        if (k > 0) {
            OpDot<T> d0{V, w};
            MIK_TRY((launch_map<T>(ctx, n, d0, vec, part, nullptr)));
            MIK_TRY(plink_fin_sum(pl, part, nseg, hd, 0));
            for (int i = 0; i + 1 < k; ++i) {
                OpMgsPass<T, false> op{w, V + (int64_t)i * ldv, V + (int64_t)(i + 1) * ldv, coef_ptr<T>(hd + i), mik_mgs_pass_hints(ctx, n, sizeof(T))};
                MIK_TRY((launch_map<T>(ctx, n, op, vec, part, nullptr)));
                MIK_TRY(plink_fin_sum(pl, part, nseg, hd + i + 1, 0));
            }
            OpMgsPass<T, true> last{w, V + (int64_t)(k - 1) * ldv, nullptr, coef_ptr<T>(hd + k - 1), mik_mgs_pass_hints(ctx, n, sizeof(T))};
            MIK_TRY((launch_map<T>(ctx, n, last, vec, part, nullptr)));
        } else {
            MIK_TRY((launch_map<T>(ctx, n, dn, vecw, part, nullptr)));
        }
        MIK_TRY(plink_fin_sum(pl, part, nseg, hd + k, 1));               // nrm, 1 / nrm (NaN, 1 outside the safe range)
        OpScal<T> sc{w, coef_ptr<T>(hd + k + 1)};                        // w .*= inv(nrm)  :76
        MIK_TRY((launch_map<T>(ctx, n, sc, vecw, (T *)nullptr, nullptr)));
        MIK_TRY(download(0, out.data(), k + 1));
        for (int j = 0; j < k; ++j) h[j] = out[(size_t)j];
        nrm = out[(size_t)k];
        if (nrm != nrm) {                                                // w was left unscaled
            MIK_TRY(gm_link_norm_slow<T>(g, w, &nrm));
            OpScal<T> sc2{w, coef_val<T>(T(1) / nrm)};
            MIK_TRY((launch_map<T>(ctx, n, sc2, vecw, (T *)nullptr, nullptr)));
        }
        *nrm_out = nrm;
        return MIK_OK;
    }
    // :15-17 / :43-45: h = V' w (batched dot, summed over the ranks in one exchange), w -= V h, norm
    auto project = [&](int slot) -> int {
        if (k == 0) return MIK_OK;
        MIK_TRY(multidot<T>(ctx, n, k, V, ldv, w, hd + slot));
        MIK_TRY(plink_sum_vec(pl, hd + slot, k));
        return gemv_n_dev<T>(ctx, n, k, V, ldv, hd + slot, T(-1), w);
    };
    auto norm_w = [&](T *dst) -> int {           // norm(w) over the partition into hd[k] (and the host), scaled pass if flagged
        MIK_TRY((launch_map<T>(ctx, n, dn, vecw, part, nullptr)));
        return plink_fin_sum(pl, part, nseg, hd + k, 1);
    };
    MIK_TRY(project(0));
    MIK_TRY(norm_w(&nrm));
    if (method == MIK_CGS) {
        OpScal<T> sc{w, coef_ptr<T>(hd + k + 1)};                        // :48
        MIK_TRY((launch_map<T>(ctx, n, sc, vecw, (T *)nullptr, nullptr)));
        MIK_TRY(download(0, out.data(), k + 1));
        for (int j = 0; j < k; ++j) h[j] = out[(size_t)j];
        nrm = out[(size_t)k];
        if (nrm != nrm) {
            MIK_TRY(gm_link_norm_slow<T>(g, w, &nrm));
            OpScal<T> sc2{w, coef_val<T>(T(1) / nrm)};
            MIK_TRY((launch_map<T>(ctx, n, sc2, vecw, (T *)nullptr, nullptr)));
        }
        *nrm_out = nrm;
        return MIK_OK;
    }
    // DGKS (:19-36): the loop condition lives on the host, one wait per round
    MIK_TRY(download(0, out.data(), k + 1));
    for (int j = 0; j < k; ++j) h[j] = out[(size_t)j];
    nrm = out[(size_t)k];
    if (nrm != nrm) MIK_TRY(gm_link_norm_slow<T>(g, w, &nrm));
    MIK_TRY(dgks_link_loop<T>(g, k, V, ldv, w, h, &nrm, dgks_small_norm<T>(h, k)));   // :22
    *nrm_out = nrm;
    return MIK_OK;
}

template <typename T> static std::vector<T> &gm_H(mik_gmres *g);
template <> std::vector<double> &gm_H<double>(mik_gmres *g) { return g->H64; }
template <> std::vector<float> &gm_H<float>(mik_gmres *g) { return g->H32; }
template <typename T> static std::vector<T> &gm_nv(mik_gmres *g);
template <> std::vector<double> &gm_nv<double>(mik_gmres *g) { return g->nv64; }
template <> std::vector<float> &gm_nv<float>(mik_gmres *g) { return g->nv32; }

// ldiv!(y, P, x) for the left (side 0) or right (side 1) preconditioner: fused diagonal sweep or the callback
template <typename T> static int gm_ldiv(mik_gmres *g, int side, T *y, const T *x)
{
    const void *diag = side ? g->pr : g->pl;
    mik_ldiv_fn fn = side ? g->pr_fn : g->pl_fn;
    if (diag) {
        OpDivide<T> dv{x, (const T *)diag, y};
        return launch_map<T>(g->ctx, g->n, dv, mik_aligned16(x) && mik_aligned16(y) && mik_aligned16(diag), (T *)nullptr, nullptr);
    }
    if (fn && fn(side ? g->pr_user : g->pl_user, y, x) != 0) return mik_fail(g->ctx, MIK_ERR_CALLBACK, "gmres: the %s preconditioner callback failed", side ? "right" : "left");
    return MIK_OK;
}
static inline bool gm_has_pl(const mik_gmres *g) { return g->pl || g->pl_fn; }
static inline bool gm_has_pr(const mik_gmres *g) { return g->pr || g->pr_fn; }

// init!(arnoldi, x, b, Pl = Identity, Ax; initially_zero) -> beta      src/gmres.jl:235-255
template <typename T> static int gmres_init_residual(mik_gmres *g, int initially_zero, T *beta_out)
{
    mik_ctx *ctx = g->ctx;
    const int64_t n = g->n;
    const int64_t nseg = mik_nseg<T>(n);
    T *V0 = (T *)g->V;
    const T *b = (const T *)g->b;
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(nseg, 1) * (size_t)(g->restart + 1)));
    const bool vec = mik_aligned16(b) && mik_aligned16(V0) && mik_aligned16(g->Ax);
    if (initially_zero) {
        OpSubNrm<T> op{b, nullptr, V0};                                   // :241
        MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)ctx->partials, nullptr)));
    } else {
        MIK_TRY(gm_spmv<T>(g, (const T *)g->x, (T *)g->Ax));              // :245
        OpSubNrm<T> op{b, (const T *)g->Ax, V0};                          // :241,246
        MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)ctx->partials, nullptr)));
    }
    if (gm_has_pl(g)) {                                                   // ldiv!(Pl, first_col)  :249
        MIK_TRY(gm_ldiv<T>(g, 0, V0, V0));
        OpDot<T> dn{V0, V0};
        MIK_TRY((launch_map<T>(ctx, n, dn, mik_aligned16(V0), (T *)ctx->partials, nullptr)));
    }
    T *hd = (T *)ctx->coef;
    if (g->dist && g->part.link) {                                        // device-driven: norm over the ranks and 1 / norm inside the finaliser
        MIK_TRY(plink_fin_sum(g->part.link, ctx->partials, nseg, hd, 1)); // :252
        OpScal<T> scd{V0, coef_ptr<T>(hd + 1)};                           // :253
        MIK_TRY((launch_map<T>(ctx, n, scd, mik_aligned16(V0), (T *)nullptr, nullptr)));
        T out[2];
        MIK_TRY(coef_download<T>(ctx, 0, out, 2));
        MIK_TRY(plink_check(g->part.link, "gmres (row-partitioned)"));
        *beta_out = out[0];
        if (out[0] != out[0]) {                                           // badly scaled residual: scaled norm across the ranks, then :253
            MIK_TRY(gm_link_norm_slow<T>(g, V0, beta_out));
            OpScal<T> sc2{V0, coef_val<T>(T(1) / *beta_out)};
            MIK_TRY((launch_map<T>(ctx, n, sc2, mik_aligned16(V0), (T *)nullptr, nullptr)));
        }
        return MIK_OK;
    }
    if (g->dist) {
        T ss;
        MIK_TRY(finalize_store<T>(ctx, nseg, 1, hd));
        MIK_TRY(coef_download<T>(ctx, 0, &ss, 1));
        MIK_TRY(gm_reduce<T>(g, &ss, 1));
        T beta;
        MIK_TRY(gm_part_norm<T>(g, V0, ss, &beta));                       // :252
        const T inv = T(1) / beta;
        OpScal<T> scd{V0, coef_val<T>(inv)};                              // :253
        MIK_TRY((launch_map<T>(ctx, n, scd, mik_aligned16(V0), (T *)nullptr, nullptr)));
        *beta_out = beta;
        return MIK_OK;
    }
    MIK_TRY(finalize_nrm_inv<T>(ctx, nseg, hd));                          // :252
    OpScal<T> sc{V0, coef_ptr<T>(hd + 1)};                                // :253
    MIK_TRY((launch_map<T>(ctx, n, sc, mik_aligned16(V0), (T *)nullptr, nullptr)));
    T out[2];
    MIK_TRY(coef_download<T>(ctx, 0, out, 2));
    *beta_out = out[0];
    if (out[0] != out[0]) MIK_TRY(orth_rescale<T>(ctx, n, V0, beta_out));   // badly scaled residual: scaled norm, then :253
    return MIK_OK;
}

template <typename T> static int gmres_create_impl(mik_gmres *g, double abstol, double reltol, int initially_zero)
{
    const int m = g->restart;
    gm_H<T>(g).assign((size_t)(m + 1) * (size_t)std::max(m, 1), T(0));   // :14
    gm_nv<T>(g).assign((size_t)m + 1, T(1));                             // :27
    g->mv_products = initially_zero ? 1 : 0;                             // :122 (sic)
    T beta;
    MIK_TRY(gmres_init_residual<T>(g, initially_zero, &beta));           // :126
    g->current = (double)beta;
    g->accumulator = 1.0;                                                // :258
    g->res_beta = (double)beta;                                          // :259
    const T a = (T)reltol * beta;
    const T tol = a > (T)abstol ? a : (T)abstol;                         // :129
    g->tol = (double)tol;
    g->g_beta = (double)beta;                                            // :133
    g->k = 1;
    return MIK_OK;
}

struct GmresOpArgs {          // the callback forms of A / Pl / Pr (mik_gmres_create_op); all NULL for the CSR / diagonal entry points
    int dtype = MIK_F64;
    int64_t n = 0;
    mik_mul_fn mul = nullptr; void *mul_user = nullptr;
    mik_ldiv_fn pl = nullptr; void *pl_user = nullptr;
    mik_ldiv_fn pr = nullptr; void *pr_user = nullptr;
};

static int gmres_create_common(mik_ctx *ctx, const mik_csr *A, void *x, const void *b, const void *pl_diag, const void *pr_diag,
                               double abstol, double reltol, int restart, int64_t maxiter, int initially_zero, int orth_method,
                               const mik_partition *part, mik_gmres **out, const GmresOpArgs *op = nullptr)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    if (!A && !(op && op->mul)) return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create: NULL operator");
    if (A && !part && A->n_rows != A->n_cols) return mik_fail(ctx, MIK_ERR_MISMATCH, "mik_gmres_create: A must be square");
    if (part) {
        if (part->nranks < 1 || part->rank < 0 || part->rank >= part->nranks) return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create_partitioned: bad rank %d of %d", part->rank, part->nranks);
        if (part->n_ext != A->n_cols || part->n_ext < A->n_rows) return mik_fail(ctx, MIK_ERR_MISMATCH, "mik_gmres_create_partitioned: A_loc must be n_loc x n_ext");
        if ((part->n_ext && !part->x_ext) || part->n_send < 0 || (part->n_send && (!part->send_idx || !part->send_buf)))
            return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create_partitioned: NULL halo buffers");
        if (part->link) {
            if (!plink_ready(part->link)) return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create_partitioned: the link is not connected (mik_plink_connect, mik_comm_mailbox_connect)");
            if (plink_ctx(part->link) != ctx) return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create_partitioned: the link's communicator lives on another context (stream)");
            if (plink_rank(part->link) != part->rank || plink_nranks(part->link) != part->nranks)
                return mik_fail(ctx, MIK_ERR_MISMATCH, "mik_gmres_create_partitioned: the link belongs to rank %d of %d", plink_rank(part->link), plink_nranks(part->link));
        } else if (part->nranks > 1 && (!part->reduce || !part->halo))
            return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create_partitioned: nranks > 1 needs the callbacks or a connected link");
    }
    if (restart < 1) return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create: restart must be >= 1");
    if (orth_method != MIK_MGS && orth_method != MIK_CGS && orth_method != MIK_DGKS) return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create: bad orth_method");
    const int64_t n = A ? A->n_rows : op->n;
    const int dtype = A ? A->dtype : op->dtype;
    if (n && (!x || !b)) return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create: NULL vector");
    if ((size_t)(2 * restart + 6) * 8 > mik_ctx::COEF_SAFE_SLOT) return mik_fail(ctx, MIK_ERR_NOTIMPL, "mik_gmres_create: restart %d too large", restart);
    mik_gmres *g = new (std::nothrow) mik_gmres();
    if (!g) return mik_fail(ctx, MIK_ERR_NOMEM, "mik_gmres_create: host allocation failed");
    g->ctx = ctx; g->A = A; g->dtype = dtype; g->n = n; g->x = x; g->b = b; g->pl = pl_diag; g->pr = pr_diag;
    if (op) { g->op_mul = A ? nullptr : op->mul; g->op_user = op->mul_user; g->pl_fn = op->pl; g->pl_user = op->pl_user; g->pr_fn = op->pr; g->pr_user = op->pr_user; }
    g->restart = restart; g->maxiter = maxiter; g->method = orth_method;
    if (part) { g->dist = true; g->part = *part; }
    g->ldv = (n + 63) / 64 * 64;
    if (g->ldv == 0) g->ldv = 64;
    const size_t es = mik_dtype_size(dtype);
    hipError_t e;
    (void)hipSetDevice(ctx->device);
    if ((e = hipMalloc(&g->V, es * (size_t)g->ldv * (size_t)(restart + 1))) != hipSuccess ||
        (e = hipMalloc(&g->Ax, es * (size_t)g->ldv)) != hipSuccess) {
        mik_gmres_destroy(g);
        return mik_fail(ctx, MIK_ERR_NOMEM, "mik_gmres_create: hipMalloc of the Krylov basis (%lld x %d): %s", (long long)n, restart + 1, hipGetErrorString(e));
    }
    {
        const int64_t nseg = dtype == MIK_F64 ? mik_nseg<double>(n) : mik_nseg<float>(n);
        // single-launch Gram-Schmidt: G = 1 / 2 / 4 / 8 reduction segments per workgroup, one workgroup per compute unit at most;
        // G > 1 needs the library's own 16-byte aligned V (always the case here)
        // (a row partition: only with a device-driven link, Modified Gram-Schmidt and restart <= 62 -- the totals of a launch's passes travel
        // between the ranks through one vector slot per pass, csrc/mik_mail.h MailSumPass)
        const bool part_ok = !part || (part->link && restart <= MIK_MAIL_VEC - 2);   // (one vector mail slot per column total and the norm; DGKS: per round)
        // every workgroup of the launch waits for all the others: at most one per compute unit of THIS device (mik_resident_cap: 256 on an
        // unpartitioned MI355X, so up to 2048 segments; a 32-CU partition: up to 256), G = the smallest of 1 / 2 / 4 / 8 segments per workgroup that fits
        const int64_t cap = std::min(mik_resident_cap(ctx), 256);        // (the slot layout of k_mgs_fused / k_cgs_fused holds 256 workgroups)
        const int G = nseg <= cap ? 1 : nseg <= 2 * cap ? 2 : nseg <= 4 * cap ? 4 : 8;
        // Beyond that: Modified Gram-Schmidt in its "resident w" form (csrc/mik_mgs_res.h) -- one workgroup per compute unit keeps its part of w in
        // registers and LDS between the passes; S consecutive segments per workgroup, as long as at least 0.6 of w fits (86 segments per workgroup:
        // 22.5 M fp64 / 45 M fp32 elements on 256 CUs).  Single GPU only; MIK_KNOB_GS = 6 keeps the chains.
        const int64_t res_S = (nseg + mik_resident_cap(ctx) - 1) / mik_resident_cap(ctx);
        const bool resident = !part && orth_method == MIK_MGS && nseg > 8 * cap && res_S <= MIK_MGS_RES_MAX_S(MIK_MGS_RES_RR, MIK_MGS_RES_RL) && restart <= 254 && g->ldv % 4 == 0 && A != nullptr &&
                              ctx->tuning[MIK_KNOB_GS] != 6 && ctx->lds_per_cu >= 160 * 1024;
        if (resident) g->mgs_res_S = (int)res_S;
        if (resident || (part_ok && nseg >= 1 && nseg <= 8 * cap && restart <= 254 && (G == 1 || g->ldv % 4 == 0))) {
            g->mgs_G = resident ? 1 : G;
            g->mgs_stride = resident ? (int)((nseg + 63) / 64 * 64) : std::max<int>(256, (int)((nseg + g->mgs_G - 1) / g->mgs_G) * g->mgs_G);
            // k_cgs_fused: one more row per round (the final h values); DGKS: up to 3 rounds in the kernel
            g->mgs_rounds = orth_method == MIK_DGKS ? (ctx->tuning[MIK_KNOB_GS] == 3 ? 1 : 3) : 1;   // DGKS rounds the kernel runs before it hands back to the host loop (MIK_KNOB_GS = 3: one)
            if (part && orth_method == MIK_DGKS) g->mgs_rounds = std::max(1, std::min(g->mgs_rounds, MIK_MAIL_VEC / (restart + 1)));     // (the rounds of a launch share the 64 vector slots)
            const size_t pbytes = es * 2 * (size_t)g->mgs_rounds * (size_t)(restart + 2) * (size_t)g->mgs_stride;
            if ((e = hipMalloc(&g->mgs_P, pbytes)) != hipSuccess || (e = hipMemsetAsync(g->mgs_P, 0xFF, pbytes, ctx->stream)) != hipSuccess ||
                (e = hipMalloc((void **)&g->xl_chk, 2 * sizeof(unsigned))) != hipSuccess || (e = hipMemsetAsync(g->xl_chk, 0, 2 * sizeof(unsigned), ctx->stream)) != hipSuccess ||
                (g->mgs_mirror_stride = (sizeof(MgsMirror) + es * (size_t)(restart + 2) + 255) / 256 * 256, false) ||
                (e = hipHostMalloc((void **)&g->mgs_mirror, 2 * g->mgs_mirror_stride, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) {
                mik_gmres_destroy(g);
                return mik_fail(ctx, MIK_ERR_NOMEM, "mik_gmres_create: single-launch Gram-Schmidt buffers: %s", hipGetErrorString(e));
            }
            memset(g->mgs_mirror, 0, 2 * g->mgs_mirror_stride);
        }
    }
    if ((e = hipMemsetAsync(g->V, 0, es * (size_t)g->ldv * (size_t)(restart + 1), ctx->stream)) != hipSuccess) {   // zeros(T, n, m+1) :13
        mik_gmres_destroy(g);
        return mik_fail(ctx, MIK_ERR_HIP, "mik_gmres_create: memset: %s", hipGetErrorString(e));
    }
    int rc = dtype == MIK_F64 ? gmres_create_impl<double>(g, abstol, reltol, initially_zero)
                              : gmres_create_impl<float>(g, abstol, reltol, initially_zero);
    if (rc) { mik_gmres_destroy(g); return rc; }
    *out = g;
    return MIK_OK;
}

extern "C" int mik_gmres_create(mik_ctx *ctx, const mik_csr *A, void *x, const void *b, const void *pl_diag, const void *pr_diag,
                                double abstol, double reltol, int restart, int64_t maxiter, int initially_zero, int orth_method,
                                mik_gmres **out)
{
    return gmres_create_common(ctx, A, x, b, pl_diag, pr_diag, abstol, reltol, restart, maxiter, initially_zero, orth_method, nullptr, out);
}

extern "C" int mik_gmres_create_op(mik_ctx *ctx, const mik_operator *A, const mik_precond *Pl, const mik_precond *Pr, void *x, const void *b,
                                   double abstol, double reltol, int restart, int64_t maxiter, int initially_zero, int orth_method,
                                   mik_gmres **out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    const mik_csr *csr = nullptr;
    GmresOpArgs op;
    MIK_TRY(resolve_operator(ctx, A, "mik_gmres_create_op", &csr, &op.dtype, &op.n));
    if ((Pl && Pl->diag && Pl->ldiv) || (Pr && Pr->diag && Pr->ldiv)) return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create_op: a preconditioner has both a diagonal and a callback");
    if (!csr) { op.mul = A->mul; op.mul_user = A->user; }
    if (Pl) { op.pl = Pl->ldiv; op.pl_user = Pl->user; }
    if (Pr) { op.pr = Pr->ldiv; op.pr_user = Pr->user; }
    return gmres_create_common(ctx, csr, x, b, Pl ? Pl->diag : nullptr, Pr ? Pr->diag : nullptr, abstol, reltol, restart, maxiter, initially_zero,
                               orth_method, nullptr, out, &op);
}

extern "C" int mik_gmres_create_partitioned(mik_ctx *ctx, const mik_csr *A_loc, void *x, const void *b, const void *pl_diag,
                                            const void *pr_diag, double abstol, double reltol, int restart, int64_t maxiter,
                                            int initially_zero, int orth_method, const mik_partition *part, mik_gmres **out)
{
    if (!part) return mik_fail(ctx, MIK_ERR_INVALID, "mik_gmres_create_partitioned: NULL partition");
    return gmres_create_common(ctx, A_loc, x, b, pl_diag, pr_diag, abstol, reltol, restart, maxiter, initially_zero, orth_method, part, out);
}

extern "C" int mik_gmres_destroy(mik_gmres *g)
{
    if (!g) return MIK_OK;
    if (g->ctx) (void)hipStreamSynchronize(g->ctx->stream);
    if (g->mgs_P) (void)hipFree(g->mgs_P);
    if (g->xl_chk) (void)hipFree(g->xl_chk);
    if (g->mgs_mirror) (void)hipHostFree(g->mgs_mirror);
    if (g->V) (void)hipFree(g->V);
    if (g->Ax) (void)hipFree(g->Ax);
    delete g;
    return MIK_OK;
}

// expand!(arnoldi, Pl, Pr, k, Ax): V[:, k+1] = Pl \ (A * (Pr \ V[:, k]))   src/gmres.jl:285-304 (kernels only)
template <typename T> static int gm_expand(mik_gmres *g, T *vk, T *vk1)
{
    mik_ctx *ctx = g->ctx;
    if (gm_has_pr(g)) {
        // Pl \ (A * (Pr \ v)) through the work vector Ax                  :297-304
        MIK_TRY(gm_ldiv<T>(g, 1, vk1, vk));                                // ldiv!(nextV, Pr, V[:, k])  :300
        MIK_TRY(gm_spmv<T>(g, vk1, (T *)g->Ax));
        MIK_HIP(ctx, hipMemcpyAsync(vk1, g->Ax, sizeof(T) * (size_t)g->n, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        MIK_TRY(gm_spmv<T>(g, vk, vk1));                                  // V[:, k+1] = A * V[:, k]   :287
    }
    if (gm_has_pl(g)) MIK_TRY(gm_ldiv<T>(g, 0, vk1, vk1));              // ldiv!(Pl, nextV)  :294 / :303
    return MIK_OK;
}

// orthogonalize_and_normalize! with ModifiedGramSchmidt as ONE launch (k_mgs_fused) + one poll of a host-mapped mirror:
// n up to 256 reduction segments, where the k + 2 launches of the chain are pure dependent-launch latency.
// Split into "enqueue" (expand! + the kernel for Arnoldi column k) and "wait" (poll + read h, nrm), so that the column
// AFTER the current one can be put on the stream before the host has seen the current result: its inputs are all on the
// device, only "did the solve just converge?" is not known yet -- then the extra column is simply never read.
static inline MgsMirror *gm_mirror(mik_gmres *g, int slot) { return (MgsMirror *)((unsigned char *)g->mgs_mirror + (size_t)slot * g->mgs_mirror_stride); }

template <typename T> static int gm_expand(mik_gmres *g, T *vk, T *vk1);

template <typename T> static int gm_fused_enqueue(mik_gmres *g, int k, int slot)
{
    mik_ctx *ctx = g->ctx;
    const int64_t n = g->n;
    const int nseg = (int)mik_nseg<T>(n), G = g->mgs_G, m = (nseg + G - 1) / G, stride = g->mgs_stride;
    T *V = (T *)g->V;
    T *vk = V + (int64_t)(k - 1) * g->ldv, *w = V + (int64_t)k * g->ldv;
    MIK_TRY(gm_expand<T>(g, vk, w));
    const bool vec = mik_aligned16(V) && mik_aligned16(w) && (g->ldv % VT<T>::W == 0);
    g->mgs_seq += 1;
    g->mgs_slot_seq[slot] = g->mgs_seq;
#define MIK_CGS_GO(VECV, DG, GG)                                                                                                               \
    do {                                                                                                                                        \
        if (g->dist) {        /* a row partition with a link: column totals and the norm are summed over the ranks inside the launch */        \
            const PlinkMail pm = plink_mail(g->part.link);                                                                                      \
            const MailSumPass xch{pm.peers, pm.P, pm.rank, plink_next_vec_tag(g->part.link), pm.ticks, pm.err};                                 \
            hipLaunchKernelGGL((k_cgs_fused<T, VECV, DG, GG, MailSumPass>), dim3(m), dim3(MIK_BLOCK), 0, ctx->stream, n, k, (const T *)V, g->ldv, w, (T *)g->mgs_P, \
                               g->restart, stride, nseg, g->mgs_rounds, g->mgs_parity, gm_mirror(g, slot), g->mgs_seq, xch);                    \
        } else                                                                                                                                  \
            hipLaunchKernelGGL((k_cgs_fused<T, VECV, DG, GG>), dim3(m), dim3(MIK_BLOCK), 0, ctx->stream, n, k, (const T *)V, g->ldv, w, (T *)g->mgs_P, g->restart, \
                               stride, nseg, g->mgs_rounds, g->mgs_parity, gm_mirror(g, slot), g->mgs_seq, NoExchange{});                       \
    } while (0)
#define MIK_MGS_GO(VECV, GG)                                                                                                                   \
    do {                                                                                                                                        \
        if (g->dist) {        /* a row partition with a link: the totals of every pass are summed over the ranks inside the launch */          \
            const PlinkMail pm = plink_mail(g->part.link);                                                                                      \
            const MailSumPass xch{pm.peers, pm.P, pm.rank, plink_next_vec_tag(g->part.link), pm.ticks, pm.err};                                 \
            hipLaunchKernelGGL((k_mgs_fused<T, VECV, GG, false, MailSumPass>), dim3(m), dim3(MIK_BLOCK), 0, ctx->stream, n, k, (const T *)V, g->ldv, w, (T *)g->mgs_P, \
                               g->restart, stride, nseg, g->mgs_parity, gm_mirror(g, slot), g->mgs_seq, (unsigned *)nullptr, xch);              \
        } else                                                                                                                                  \
            hipLaunchKernelGGL((k_mgs_fused<T, VECV, GG>), dim3(m), dim3(MIK_BLOCK), 0, ctx->stream, n, k, (const T *)V, g->ldv, w, (T *)g->mgs_P, g->restart, \
                               stride, nseg, g->mgs_parity, gm_mirror(g, slot), g->mgs_seq, (unsigned *)nullptr, NoExchange{});                 \
    } while (0)
#define MIK_GS_GO(VECV, GG)                                                                          \
    do {                                                                                             \
        if (g->method == MIK_DGKS) MIK_CGS_GO(VECV, true, GG);                                       \
        else if (g->method == MIK_CGS) MIK_CGS_GO(VECV, false, GG);                                  \
        else MIK_MGS_GO(VECV, GG);                                                                   \
    } while (0)
    if (g->mgs_res_S > 0) {
        // the resident-w form: ceil(nseg / S) workgroups of 512 threads, one per compute unit (128 KB of LDS each)
        const int S = g->mgs_res_S, wgs = (nseg + S - 1) / S;
        hipLaunchKernelGGL((k_mgs_resident<T, 512, MIK_MGS_RES_RR, MIK_MGS_RES_RL>), dim3(wgs), dim3(512), 0, ctx->stream, n, k, (const T *)V, g->ldv, w, (T *)g->mgs_P, g->restart,
                           stride, nseg, S, g->mgs_parity, gm_mirror(g, slot), g->mgs_seq);
        g->xl_last = false;
        MIK_LAUNCH_CHECK(ctx);
        g->mgs_parity ^= 1;
        return MIK_OK;
    }
    // Modified Gram-Schmidt on small systems: the XCD-local form (k_mgs_fused XL) -- at most 128 workgroups, all on the first XCD, a column
    // of at most 512 KB per pass: every column then comes through ONE XCD's share of the fabric (~1 MB per us).  fe_shell (363 KB columns):
    // GMRES(50) 64.2 -> 50.9 us per inner iteration; configs[2] (1 MB columns) would lose (36.5 -> 47.8 us) and keeps the device-wide
    // form.  MIK_KNOB_GS = 4: the device-wide form at every size.
    // (the form is compiled for the 8-XCD dispatch: blockIdx % 8 == 0 -> first XCD; up to 4 workgroups on each of that XCD's compute units)
    const bool xl = g->method == MIK_MGS && G == 1 && vec && mik_xcd_maps(ctx) && m <= 4 * (mik_cus(ctx) / mik_xcds(ctx)) && ((size_t)n * sizeof(T) <= (1u << 19) || ctx->tuning[MIK_KNOB_GS] == 5) && !g->xl_off && ctx->tuning[MIK_KNOB_GS] != 4 && g->xl_chk && !g->dist;   // (MIK_KNOB_GS = 5: whatever the column size)
    g->xl_last = xl;
    if (xl) {
        hipLaunchKernelGGL((k_mgs_fused<T, true, 1, true>), dim3(8 * m), dim3(MIK_BLOCK), 0, ctx->stream, n, k, (const T *)V, g->ldv, w, (T *)g->mgs_P, g->restart,
                           stride, nseg, g->mgs_parity, gm_mirror(g, slot), g->mgs_seq, g->xl_chk, NoExchange{});
    } else
    // G > 1 (more than 256 segments) is instantiated for 16-byte aligned bases only: gmres_create allocates V that way
    if (G == 1) { if (vec) MIK_GS_GO(true, 1); else MIK_GS_GO(false, 1); }
    else if (G == 2) MIK_GS_GO(true, 2);
    else if (G == 4) MIK_GS_GO(true, 4);
    else MIK_GS_GO(true, 8);
#undef MIK_GS_GO
#undef MIK_MGS_GO
#undef MIK_CGS_GO
    MIK_LAUNCH_CHECK(ctx);
    g->mgs_parity ^= 1;
    return MIK_OK;
}

constexpr int MIK_GS_TIMEOUT = -100;      // internal: the bounded spin of the single-launch Gram-Schmidt expired

template <typename T> static int gm_fused_wait(mik_gmres *g, int k, int slot, T *h_out, T *nrm_out, bool *rescaled)
{
    mik_ctx *ctx = g->ctx;
    MgsMirror *mir = gm_mirror(g, slot);
    const unsigned long long want = g->mgs_slot_seq[slot];
    volatile unsigned long long *p = &mir->seq;
    for (unsigned long long spins = 0;; ++spins) {
        if (__atomic_load_n((const unsigned long long *)p, __ATOMIC_ACQUIRE) == want) break;
        if ((spins & 0xFFFFF) == 0xFFFFF) {
            hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) {
                if (__atomic_load_n((const unsigned long long *)p, __ATOMIC_ACQUIRE) == want) break;
                return mik_fail(ctx, MIK_ERR_HIP, "gmres: stream idle but the Gram-Schmidt step %llu was never published", want);
            }
            if (e != hipErrorNotReady) return mik_fail(ctx, MIK_ERR_HIP, "gmres: %s while waiting for the Gram-Schmidt step", hipGetErrorString(e));
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (mir->err || ctx->tuning[MIK_KNOB_GS_TIMEOUT] == 1) return MIK_GS_TIMEOUT;        // (development knob MIK_KNOB_GS_TIMEOUT: pretend it happened)
    // a workgroup gave up waiting for a slot (GPU shared with other work?): the caller redoes the column
    const T *out = reinterpret_cast<const T *>(mir + 1);
    for (int j = 0; j < k; ++j) h_out[j] = out[j];
    *nrm_out = out[k];
    *rescaled = false;
    if (g->method == MIK_DGKS && mir->pad) {    // the kernel stopped before the DGKS loop ended (round limit / unsafe norm): w is unscaled
        T *w = (T *)g->V + (int64_t)k * g->ldv;
        T nrm = out[k];
        if (g->dist) {                          // (every rank sees the same totals: all of them come here together)
            if (nrm != nrm) MIK_TRY(gm_link_norm_slow<T>(g, w, &nrm));
            MIK_TRY(dgks_link_loop<T>(g, k, (const T *)g->V, g->ldv, w, h_out, &nrm, out[k + 1]));
        } else {
            if (nrm != nrm) MIK_TRY(mik_safe_norm_slow<T>(ctx, g->n, w, &nrm));
            MIK_TRY(dgks_host_loop<T>(ctx, g->n, k, (const T *)g->V, g->ldv, w, h_out, &nrm, out[k + 1]));
        }
        *nrm_out = nrm;
        *rescaled = true;
        return MIK_OK;
    }
    if (out[k] != out[k]) {                     // sum of squares outside the safe range: the kernel left w unscaled
        T *w = (T *)g->V + (int64_t)k * g->ldv;
        if (g->dist) {                          // ... on every rank alike (identical totals): the scaled norm across the ranks
            MIK_TRY(gm_link_norm_slow<T>(g, w, nrm_out));
            OpScal<T> sc{w, coef_val<T>(T(1) / *nrm_out)};
            MIK_TRY((launch_map<T>(ctx, g->n, sc, mik_aligned16(w), (T *)nullptr, nullptr)));
        } else MIK_TRY(orth_rescale<T>(ctx, g->n, w, nrm_out));
        *rescaled = true;
    }
    return MIK_OK;
}

// iterate(g::GMRESIterable, iteration)                                   src/gmres.jl:57-106
template <typename T> static int gmres_iterate_impl(mik_gmres *g, int64_t iteration, double *residual, int *done)
{
    mik_ctx *ctx = g->ctx;
    const int m = g->restart;
    const int64_t ldh = m + 1;
    std::vector<T> &H = gm_H<T>(g);
    std::vector<T> &nullvec = gm_nv<T>(g);
    auto Hat = [&](int i, int j) -> T & { return H[(size_t)j * ldh + i]; };
    auto is_done = [&](int64_t it) { return it >= g->maxiter || (T)g->current <= (T)g->tol; };   // :55, :51

    if (is_done(iteration)) { *done = 1; if (residual) *residual = g->current; return MIK_OK; }   // :59
    *done = 0;
    int k = g->k;                                                         // 1-based, as in the reference
    T *V = (T *)g->V;
    T *vk = V + (int64_t)(k - 1) * g->ldv, *vk1 = V + (int64_t)k * g->ldv;

    // expand! (:64, :285-304), then H[k+1, k] = orthogonalize_and_normalize!(V[:, 1:k], V[:, k+1], H[1:k, k], orth_meth)  :68-73
    T nrm;
    {
        if (g->mgs_P && (!g->dist || g->part.link) && !g->fused_off && ctx->tuning[MIK_KNOB_GS] != 1 && ctx->tuning[MIK_KNOB_GS] != 2) {    // MIK_KNOB_GS: 1 / 2 = the multi-launch chains, 4 = single launch on all XCDs
            // single-launch Gram-Schmidt, one column ahead of the host: column k is on the stream already if the previous
            // call put it there; column k + 1 goes on the stream BEFORE this call waits for column k (never across a restart,
            // never with host callbacks in expand!, whose call count the caller may observe)
            const int slot = g->pre_k == k ? g->pre_slot : 0;
            if (g->pre_k != k) MIK_TRY(gm_fused_enqueue<T>(g, k, slot));
            g->pre_k = 0;
            const bool ahead = k < m && iteration + 1 < g->maxiter && !g->op_mul && !g->pl_fn && !g->pr_fn && ctx->tuning[MIK_KNOB_NO_LOOKAHEAD] == 0;
            if (ahead) MIK_TRY(gm_fused_enqueue<T>(g, k + 1, slot ^ 1));
            bool rescaled = false;
            const int rcw = gm_fused_wait<T>(g, k, slot, &Hat(0, k - 1), &nrm, &rescaled);
            if (rcw == MIK_GS_TIMEOUT && g->dist)      // (no local fall-back over a partition: the ranks' exchange sequences would part ways)
                return mik_fail(ctx, MIK_ERR_HIP, "gmres (row-partitioned): a hand-off of the single-launch Gram-Schmidt timed out (a peer rank stopped, or the GPU is shared with other work)");
            if (rcw == MIK_GS_TIMEOUT) {
                g->gs_timeouts += 1;
                // The hand-off needs every workgroup resident; a GPU shared with other streams or processes can break that.
                // Not an error of the solve: drain the stream (a column enqueued ahead drains with it), switch this handle to the
                // multi-launch chains for good, and redo column k from V[:, k - 1], which the kernel never writes.
                MIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                gm_mirror(g, 0)->err = 0;
                gm_mirror(g, 1)->err = 0;
                if (g->xl_last && !g->xl_off) g->xl_off = true;       // the XCD-local form: back to the all-XCD single launch, not to the chains
                else g->fused_off = true;
                g->pre_k = 0;
                MIK_TRY(gm_expand<T>(g, vk, vk1));
                MIK_TRY(orthogonalize_impl<T>(ctx, g->n, k, V, g->ldv, vk1, &Hat(0, k - 1), &nrm, g->method));
            } else {
                MIK_TRY(rcw);
                if (ahead && !rescaled) { g->pre_k = k + 1; g->pre_slot = slot ^ 1; }    // rescaled: column k + 1 was built on an unscaled w
            }
        } else {
            MIK_TRY(gm_expand<T>(g, vk, vk1));
            if (g->dist && g->part.link) MIK_TRY(orthogonalize_link<T>(g, k, V, g->ldv, vk1, &Hat(0, k - 1), &nrm, g->method));
            else if (g->dist) MIK_TRY(orthogonalize_part<T>(g, k, V, g->ldv, vk1, &Hat(0, k - 1), &nrm, g->method));
            else MIK_TRY(orthogonalize_impl<T>(ctx, g->n, k, V, g->ldv, vk1, &Hat(0, k - 1), &nrm, g->method));
        }
    }
    g->mv_products += 1;                                                  // :65
    Hat(k, k - 1) = nrm;

    // update_residual!                                                   :76, :224-233
    T current = (T)g->current, accumulator = (T)g->accumulator;
    if (Hat(k, k - 1) == T(0)) {
        current = T(0);
    } else {
        T dsum = T(0);
        for (int i = 0; i < k; ++i) { T p = nullvec[i] * Hat(i, k - 1); dsum = dsum + p; }
        nullvec[k] = -(dsum / Hat(k, k - 1));
        T sq = nullvec[k] * nullvec[k];
        accumulator = accumulator + sq;
        current = (T)g->res_beta / std::sqrt(accumulator);
    }
    g->current = (double)current;
    g->accumulator = (double)accumulator;

    k += 1;                                                               // :78

    if (k == m + 1 || is_done(iteration + 1)) {                           // :82
        // solve_least_squares!: rhs = [beta, 0, ...] of length k; H view = H[1:k, 1:k-1]   :85, :262-271
        std::vector<T> rhs((size_t)k, T(0));
        rhs[0] = (T)g->g_beta;
        hessenberg_ldiv<T>(H.data(), ldh, k - 1, rhs.data());
        // update_solution!: x += V[:, 1:k-1] * y                         :88, :273-276
        MIK_TRY(coef_upload<T>(ctx, 0, rhs.data(), k - 1));
        if (gm_has_pr(g)) {
            // x += Pr \ (V * y) with Ax as work space                     :278-283
            T *Ax = (T *)g->Ax;
            OpFill<T> z{Ax, T(0)};
            MIK_TRY((launch_map<T>(ctx, g->n, z, mik_aligned16(Ax), (T *)nullptr, nullptr)));
            MIK_TRY(gemv_n_dev<T>(ctx, g->n, k - 1, V, g->ldv, (const T *)ctx->coef, T(1), Ax));
            MIK_TRY(gm_ldiv<T>(g, 1, Ax, Ax));                               // ldiv!(Pr, Ax)  :281
            OpAxpy<T> ax{Ax, (T *)g->x, coef_val<T>(T(1))};
            MIK_TRY((launch_map<T>(ctx, g->n, ax, mik_aligned16(Ax) && mik_aligned16(g->x), (T *)nullptr, nullptr)));
        } else {
            MIK_TRY(gemv_n_dev<T>(ctx, g->n, k - 1, V, g->ldv, (const T *)ctx->coef, T(1), (T *)g->x));
        }
        k = 1;                                                            // :90
        g->pre_k = 0;                                                     // a column enqueued ahead belongs to the cycle that just ended
        if (!is_done(iteration)) {                                        // :93
            T beta;
            MIK_TRY(gmres_init_residual<T>(g, 0, &beta));                 // :96
            g->g_beta = (double)beta;
            g->accumulator = 1.0;                                         // :99, :258
            g->res_beta = (double)beta;                                   // :259
            g->mv_products += 1;                                          // :101
        }
    }
    g->k = k;
    if (residual) *residual = g->current;                                 // :105
    return MIK_OK;
}

extern "C" int mik_gmres_iterate(mik_gmres *g, int64_t iteration, double *residual, int *done)
{
    if (!g || !done || iteration < 0) return MIK_ERR_INVALID;
    return g->dtype == MIK_F64 ? gmres_iterate_impl<double>(g, iteration, residual, done)
                               : gmres_iterate_impl<float>(g, iteration, residual, done);
}

// Up to max_steps consecutive iterate() calls inside the library: the loop of gmres! (src/gmres.jl:207-214) without a
// trip through the host language per inner iteration (each step still waits once for its Hessenberg column).
extern "C" int mik_gmres_iterate_many(mik_gmres *g, int64_t iteration, int64_t max_steps, double *residuals, int64_t *steps_done)
{
    if (!g || !steps_done || iteration < 0) return MIK_ERR_INVALID;
    *steps_done = 0;
    for (int64_t j = 0; j < max_steps; ++j) {
        double res = 0;
        int done = 0;
        int rc = g->dtype == MIK_F64 ? gmres_iterate_impl<double>(g, iteration + j, &res, &done)
                                     : gmres_iterate_impl<float>(g, iteration + j, &res, &done);
        if (rc) return rc;
        if (done) break;
        if (residuals) residuals[j] = res;
        *steps_done = j + 1;
    }
    return MIK_OK;
}

// development introspection (include/mik_dev.h): which form the next Arnoldi column's orthogonalize_and_normalize! runs in
extern "C" int mik_dev_gmres_form(const mik_gmres *g, int *single_launch, int *segments_per_workgroup, int *xcd_local_last, int *timeouts)
{
    if (!g) return MIK_ERR_INVALID;
    const mik_ctx *ctx = g->ctx;
    const bool single = g->mgs_P && (!g->dist || g->part.link) && !g->fused_off && ctx->tuning[MIK_KNOB_GS] != 1 && ctx->tuning[MIK_KNOB_GS] != 2;
    if (single_launch) *single_launch = single ? 1 : 0;
    if (segments_per_workgroup) *segments_per_workgroup = g->mgs_P ? (g->mgs_res_S > 0 ? g->mgs_res_S : g->mgs_G) : 0;
    if (xcd_local_last) *xcd_local_last = g->xl_last ? 1 : 0;
    if (timeouts) *timeouts = g->gs_timeouts;
    return MIK_OK;
}

int mik_mgs_resident_max_s() { return MIK_MGS_RES_MAX_S(MIK_MGS_RES_RR, MIK_MGS_RES_RL); }

extern "C" int mik_dev_mgs_resident_shape(int *threads, int *register_rounds, int *lds_rounds)
{
    if (threads) *threads = 512;
    if (register_rounds) *register_rounds = MIK_MGS_RES_RR;
    if (lds_rounds) *lds_rounds = MIK_MGS_RES_RL;
    return MIK_OK;
}

extern "C" int mik_gmres_state(const mik_gmres *g, double *residual, double *tol, double *beta, int *k, int64_t *mv_products,
                               int *converged)
{
    if (!g) return MIK_ERR_INVALID;
    if (residual) *residual = g->current;
    if (tol) *tol = g->tol;
    if (beta) *beta = g->g_beta;
    if (k) *k = g->k;
    if (mv_products) *mv_products = g->mv_products;
    if (converged) *converged = g->current <= g->tol ? 1 : 0;            // src/gmres.jl:51
    return MIK_OK;
}

// =============================================================================================
// Row-partitioned CGIterable (one process per GPU)
// =============================================================================================
//
// New design (the reference is single-process): rank p owns a contiguous block of rows; the
// operator's local block is a CSR matrix whose columns are renumbered to [0, n_loc) for owned
// entries of x and [n_loc, n_loc + n_ghost) for halo entries received from the neighbours.  The
// host side (dist.py) runs the exchanges with torch.distributed (RCCL over xGMI on the GPU box,
// gloo in the CPU tests) between the phases below; every phase only enqueues kernels.
//
//   phase 10  init A: (x given) copy x into u_ext, pack the halo send buffer
//   phase 11  init B: r = b - A*u_ext (or r = b), local sum of r.^2 -> rr_all[rank]; u_ext = 0
//   phase 12  init C: residual = sqrt(sum_p rr_all[p]), tol, beta; publish
//   phase  0  step A: u = r + beta*u (src/cg.jl:51); pack the halo send buffer
//   phase  1  step B: c = A_loc * u_ext with the local dot(u, c) -> dot_all[rank]   (:54-55)
//   phase  2  step C: alpha = residual^2 / sum_p dot_all[p]; x += alpha*u; r -= alpha*c;
//                     local sum of r.^2 -> rr_all[rank]                               (:55-59)
//   phase  3  step D: residual = sqrt(sum_p rr_all[p]) (:62); beta; stopping test; publish
//
// Cross-rank sums run in rank order 0..P-1 on every rank, so all ranks hold bit-identical
// scalars and the history is reproducible (and equals the single-GPU path for P = 1).

template <typename T>
__global__ void k_gather(int64_t m, const int *__restrict__ idx, const T *__restrict__ x, T *__restrict__ out, const int *__restrict__ done)
{
    if (done && *done) return;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) out[i] = x[idx[i]];
}

// The rows the neighbours need, updated and packed in ONE launch (every send index occurs once): u = r + beta u (and the
// pending x update) exactly as OpXpbyX::apply does it, then the new u straight into the send buffer.
template <typename T>
__global__ __launch_bounds__(MIK_BLOCK) void k_cgd_early(int64_t m, const int *__restrict__ idx, const T *__restrict__ r, T *__restrict__ u, T *__restrict__ x,
                                                         const T *__restrict__ beta, const T *__restrict__ alpha, const int *__restrict__ done,
                                                         const int *__restrict__ pending, int fuse_x, T *__restrict__ out)
{
    const int dn = *done, pd = fuse_x ? *pending : 0;
    for (int64_t j = (int64_t)blockIdx.x * MIK_BLOCK + threadIdx.x; j < m; j += (int64_t)gridDim.x * MIK_BLOCK) {
        const int i = idx[j];
        T uo = u[i];
        if (pd) { T t = *alpha * uo; x[i] = x[i] + t; }
        if (!dn) { T t = *beta * uo; uo = r[i] + t; u[i] = uo; }
        out[j] = uo;
    }
}

template <typename T> static int gather_launch(mik_ctx *ctx, int64_t m, const int *idx, const T *x, T *out, const int *done)
{
    if (m <= 0) return MIK_OK;
    const int grid = (int)std::min<int64_t>((m + MIK_BLOCK - 1) / MIK_BLOCK, mik_max_grid(ctx));
    hipLaunchKernelGGL((k_gather<T>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, m, idx, x, out, done);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

extern "C" int mik_gather(mik_ctx *ctx, int dtype, int64_t m, const int32_t *idx, const void *x, void *out)
{
    if (!ctx || m < 0 || (m && (!idx || !x || !out))) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return gather_launch<double>(ctx, m, idx, (const double *)x, (double *)out, nullptr);
    if (dtype == MIK_F32) return gather_launch<float>(ctx, m, idx, (const float *)x, (float *)out, nullptr);
    return MIK_ERR_INVALID;
}

// sum of the P per-rank partial sums, rank order
template <typename T> __device__ __forceinline__ T rank_sum(const T *__restrict__ all, int nranks)
{
    T s = all[0];
    for (int p = 1; p < nranks; ++p) s = s + all[p];
    return s;
}

// this rank's partial (level-2 sum of its segment sums, spread over MIK_FIN_WGS single-wave workgroups like k_cg_fin_*) into its slot
template <typename T>
__global__ __launch_bounds__(64) void k_cgd_fin_slot(const T *__restrict__ S, int64_t m, T *__restrict__ slot, const int *__restrict__ done, FinScratch<T> *fs)
{
    if (done && *done) return;
    T tot;
    if (level2_sum_spread(S, m, fs, tot)) *slot = tot;
}

// k_cgd_fin_slot for |r|^2 that also does what k_cgd_alpha did for this step (whose sweep formed alpha by itself: CoefAlphaRanks):
// the pending-x flag cleared, dot(u, c) and alpha stored for the head of the next step -- one launch less per step
template <typename T>
__global__ __launch_bounds__(64) void k_cgd_fin_slot_alpha(const T *__restrict__ S, int64_t m, T *__restrict__ slot, const int *__restrict__ done,
                                                           FinScratch<T> *fs, const T *__restrict__ dot_all, int nranks, CgDev<T> *d)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) d->x_pending = 0;     // the sweep over u of this step's head applied it (OpXpbyX)
    if (done && *done) return;
    T tot;
    if (level2_sum_spread(S, m, fs, tot)) {
        *slot = tot;
        const T dt = rank_sum(dot_all, nranks);
        d->dot_uc = dt;
        d->alpha = (d->res * d->res) / dt;
    }
}

template <typename T> __global__ void k_cgd_alpha(const T *__restrict__ dot_all, int nranks, CgDev<T> *d)
{
    d->x_pending = 0;                          // the sweep over u of this step's head applied it (OpXpbyX)
    if (d->done) return;
    const T tot = rank_sum(dot_all, nranks);
    d->dot_uc = tot;
    d->alpha = (d->res * d->res) / tot;
}

template <typename T>
__global__ void k_cgd_fin_init(const T *__restrict__ rr_all, int nranks, CgDev<T> *d, T reltol, T abstol, long long maxiter, CgMirror *mirror,
                               unsigned long long seq)
{
    const T tot = rank_sum(rr_all, nranks);
    if (!mik_nrm_in_range(tot)) {
        // |r|^2 left the range of a safe sqrt(sum of squares) -- every rank sees the same total and takes this branch: the hosts
        // recompute the norm with a common scale (phases 20-24) and finish the initialisation with k_cgd_fix_init
        d->done = 1; mirror->done = 0; mirror->range = 1;
        __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const T res = mik_sqrt(tot);
    const T a = reltol * res;
    d->rr = tot; d->res = res; d->prev_res = T(1); d->rho = T(1);
    d->tol = a > abstol ? a : abstol;
    d->beta = (res * res) / (T(1) * T(1));
    d->alpha = T(0); d->dot_uc = T(0);
    d->done = (0 >= maxiter || res <= d->tol) ? 1 : 0;
    d->nhist = 0;
    mirror->res = (double)res; mirror->prev_res = 1.0; mirror->done = d->done; mirror->nhist = 0;
    mirror->tol = (double)d->tol; mirror->tol_valid = 1;
    __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
__global__ void k_cgd_fin_res(const T *__restrict__ rr_all, int nranks, CgDev<T> *d, T *__restrict__ hist, long long it_next, long long maxiter,
                              CgMirror *mirror, unsigned long long seq, int hist_index, int fuse_x)
{
    if (d->done) { __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); return; }
    cgd_close_step<T>(d, rank_sum(rr_all, nranks), hist, it_next, maxiter, mirror, seq, hist_index, fuse_x);
}

// finishes cg_iterator! with a residual norm obtained through the scaled pass over the partition
template <typename T>
__global__ void k_cgd_fix_init(CgDev<T> *d, T res, T reltol, T abstol, long long maxiter, CgMirror *mirror, unsigned long long seq)
{
    const T a = reltol * res;
    d->rr = res * res; d->res = res; d->prev_res = T(1); d->rho = T(1);
    d->tol = a > abstol ? a : abstol;
    d->beta = (res * res) / (T(1) * T(1));
    d->alpha = T(0); d->dot_uc = T(0);
    d->done = (0 >= maxiter || res <= d->tol) ? 1 : 0;
    d->nhist = 0;
    mirror->range = 0;
    mirror->res = (double)res; mirror->prev_res = 1.0; mirror->done = d->done; mirror->nhist = 0;
    mirror->tol = (double)d->tol; mirror->tol_valid = 1;
    __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int mik_cgd_create(mik_ctx *ctx, const mik_csr *A_loc, void *x, const void *b, void *u_ext, void *r, void *c,
                              const int32_t *send_idx, int64_t n_send, void *send_buf, void *dot_all, void *rr_all, int rank,
                              int nranks, double abstol, double reltol, int64_t maxiter, int initially_zero, mik_cgd **out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    if (!A_loc || A_loc->n_cols < A_loc->n_rows) return mik_fail(ctx, MIK_ERR_MISMATCH, "mik_cgd_create: local block must be n_loc x (n_loc + n_ghost)");
    if (nranks < 1 || rank < 0 || rank >= nranks || !dot_all || !rr_all || (n_send && (!send_idx || !send_buf)))
        return mik_fail(ctx, MIK_ERR_INVALID, "mik_cgd_create: bad rank / comm buffers");
    const int64_t n = A_loc->n_rows;
    if (n && (!x || !b || !u_ext || !r || !c)) return mik_fail(ctx, MIK_ERR_INVALID, "mik_cgd_create: NULL vector");
    mik_cgd *it = new (std::nothrow) mik_cgd();
    if (!it) return mik_fail(ctx, MIK_ERR_NOMEM, "mik_cgd_create: host allocation failed");
    mik_cg &bs = it->base;
    bs.ctx = ctx; bs.A = A_loc; bs.dtype = A_loc->dtype; bs.n = n;
    bs.x = x; bs.b = b; bs.u = u_ext; bs.r = r; bs.c = c; bs.maxiter = maxiter;
    it->rank = rank; it->nranks = nranks; it->n_send = n_send; it->send_idx = send_idx; it->send_buf = send_buf;
    it->u_ext = u_ext; it->n_ext = A_loc->n_cols; it->dot_all = dot_all; it->rr_all = rr_all;
    it->abstol = abstol; it->reltol = reltol; it->initially_zero = initially_zero;
    const size_t es = mik_dtype_size(A_loc->dtype);
    const int64_t nseg = A_loc->dtype == MIK_F64 ? mik_nseg<double>(n) : mik_nseg<float>(n);
    const int64_t nb = mik_spmv_nwg(n);
    hipError_t e;
    (void)hipSetDevice(ctx->device);
    auto fail = [&](const char *what, hipError_t err) {
        int rc = mik_fail(ctx, MIK_ERR_NOMEM, "mik_cgd_create: %s: %s", what, hipGetErrorString(err));
        mik_cgd_destroy(it);
        return rc;
    };
    if ((e = hipMalloc(&bs.dev, 256)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc(&bs.seg_spmv, es * (size_t)std::max<int64_t>(nb, 1))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc(&bs.seg_vec, es * (size_t)std::max<int64_t>(nseg, 1))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc(&bs.hist, es * 1024)) != hipSuccess) return fail("hipMalloc", e);
    bs.hist_cap = 1024;
    if ((e = hipMalloc(&bs.fin, 256)) != hipSuccess || (e = hipMemsetAsync(bs.fin, 0, 256, ctx->stream)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipHostMalloc((void **)&bs.mirror, sizeof(CgMirror), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return fail("hipHostMalloc", e);
    memset(bs.mirror, 0, sizeof(CgMirror));
    if ((e = hipMemsetAsync(bs.dev, 0, 256, ctx->stream)) != hipSuccess) return fail("hipMemsetAsync", e);
    *out = it;
    return MIK_OK;
}

extern "C" int mik_cgd_set_interior(mik_cgd *it, int64_t rb_begin, int64_t rb_end)
{
    if (!it) return MIK_ERR_INVALID;
    const int64_t nb = mik_spmv_nwg(it->base.n);
    if (rb_begin < 0 || rb_end < rb_begin || rb_end > nb) return mik_fail(it->base.ctx, MIK_ERR_INVALID, "mik_cgd_set_interior: bad range");
    if (!mik_spmv_can_split(it->base.A)) return mik_fail(it->base.ctx, MIK_ERR_NOTIMPL, "mik_cgd_set_interior: the operator's layout cannot be launched over a row-block range");
    it->int_begin = rb_begin;
    it->int_end = rb_end;
    return MIK_OK;
}

void mik_cgd_group_forget(mik_cgd *it);   // mik_comm.hip

extern "C" int mik_cgd_destroy(mik_cgd *it)
{
    if (!it) return MIK_OK;
    mik_cgd_group_forget(it);
    mik_cg &bs = it->base;
    if (bs.ctx) (void)hipStreamSynchronize(bs.ctx->stream);
    if (it->link) { (void)mik_plink_destroy(it->link); it->link = nullptr; }
    if (bs.dev) (void)hipFree(bs.dev);
    if (bs.hist) (void)hipFree(bs.hist);
    if (bs.fin) (void)hipFree(bs.fin);
    if (bs.seg_spmv) (void)hipFree(bs.seg_spmv);
    if (bs.seg_vec) (void)hipFree(bs.seg_vec);
    if (bs.mirror) (void)hipHostFree(bs.mirror);
    delete it;
    return MIK_OK;
}

template <typename T> static int cgd_phase_impl(mik_cgd *it, int phase, int64_t iteration)
{
    mik_cg &bs = it->base;
    mik_ctx *ctx = bs.ctx;
    CgDev<T> *d = (CgDev<T> *)bs.dev;
    const int *done = &d->done;
    const int64_t n = bs.n, nseg = mik_nseg<T>(n), nb = mik_spmv_nwg(n);
    T *x = (T *)bs.x, *u = (T *)it->u_ext, *r = (T *)bs.r, *c = (T *)bs.c;
    const T *b = (const T *)bs.b;
    T *dot_slot = (T *)it->dot_all + it->rank, *rr_slot = (T *)it->rr_all + it->rank;
    const bool vec = mik_aligned16(x) && mik_aligned16(u) && mik_aligned16(r) && mik_aligned16(c) && mik_aligned16(b);
    switch (phase) {
    case 10:   // init A
        if (!it->initially_zero) {
            MIK_HIP(ctx, hipMemcpyAsync(u, x, sizeof(T) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
            MIK_TRY(gather_launch<T>(ctx, it->n_send, it->send_idx, u, (T *)it->send_buf, nullptr));
        }
        return MIK_OK;
    case 11: { // init B
        if (it->initially_zero) {
            bs.mv_products = 0;
            OpSubNrm<T> op{b, nullptr, r};
            MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)bs.seg_vec, nullptr)));
        } else {
            bs.mv_products = 1;
            MIK_TRY(mik_spmv_launch<T>(ctx, bs.A, u, c, false, nullptr, nullptr));
            OpSubNrm<T> op{b, c, r};
            MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)bs.seg_vec, nullptr)));
        }
        hipLaunchKernelGGL((k_finalize_store<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)bs.seg_vec, nseg, (int64_t)0, rr_slot,
                           (const int *)nullptr);
        MIK_LAUNCH_CHECK(ctx);
        OpFill<T> z{u, T(0)};   // u .= 0 (src/cg.jl:129), halo included
        return launch_map<T>(ctx, it->n_ext, z, mik_aligned16(u), (T *)nullptr, nullptr);
    }
    case 12:   // init C
        bs.seq += 1;
        hipLaunchKernelGGL((k_cgd_fin_init<T>), dim3(1), dim3(1), 0, ctx->stream, (const T *)it->rr_all, it->nranks, d, (T)it->reltol, (T)it->abstol,
                           (long long)bs.maxiter, bs.mirror, bs.seq);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    case 0: {  // step A
        if (bs.fuse_x) {   // ... with x .+= alpha .* u of the previous step on the u this sweep reads anyway (OpXpbyX)
            OpXpbyX<T> op{r, u, x, coef_ptr<T>(&d->beta), coef_ptr<T>(&d->alpha), done, &d->x_pending, cg_stream_hints(ctx, true, bs.A) & 15};
            MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)nullptr, (const int *)nullptr)));
        } else {
            OpXpby<T> op{r, u, coef_ptr<T>(&d->beta), cg_stream_hints(ctx) & 7};
            MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)nullptr, done)));
        }
        return gather_launch<T>(ctx, it->n_send, it->send_idx, u, (T *)it->send_buf, done);
    }
    case 7:    // step A, early part: u (and the pending x update) on the rows the neighbours need, then pack -- the halo leaves first
    case 8: {  // step A, bulk: the same sweep on all other rows, while the halo is on the wire
        if (it->n_early <= 0) return mik_fail(ctx, MIK_ERR_INVALID, "mik_cgd_phase: no early rows (mik_cgd_set_halo_plan)");
        int64_t lo[3], hi[3];
        int nr = 0;
        if (phase == 7) {
            for (int q = 0; q < it->n_early; ++q) { lo[nr] = it->early_a[q]; hi[nr] = it->early_b[q]; ++nr; }
        } else {
            int64_t at = 0;
            for (int q = 0; q < it->n_early; ++q) {
                if (it->early_a[q] > at) { lo[nr] = at; hi[nr] = it->early_a[q]; ++nr; }
                at = it->early_b[q];
            }
            if (at < n) { lo[nr] = at; hi[nr] = n; ++nr; }
        }
        for (int q = 0; q < nr; ++q) {
            const int64_t o = lo[q], len = hi[q] - lo[q];
            const bool v2 = vec && (o % VT<T>::W == 0);
            if (bs.fuse_x) {
                OpXpbyX<T> op{r + o, u + o, x + o, coef_ptr<T>(&d->beta), coef_ptr<T>(&d->alpha), done, &d->x_pending, cg_stream_hints(ctx, true, bs.A) & 15};
                MIK_TRY((launch_map<T>(ctx, len, op, v2, (T *)nullptr, (const int *)nullptr)));
            } else {
                OpXpby<T> op{r + o, u + o, coef_ptr<T>(&d->beta), cg_stream_hints(ctx) & 7};
                MIK_TRY((launch_map<T>(ctx, len, op, v2, (T *)nullptr, done)));
            }
        }
        if (phase == 7) return gather_launch<T>(ctx, it->n_send, it->send_idx, u, (T *)it->send_buf, done);
        return MIK_OK;
    }
    case 9: {  // step A, early part as one launch (every send index occurs once: mik_cgd_set_halo_plan)
        if (it->n_early <= 0 || !it->early_merged) return mik_fail(ctx, MIK_ERR_INVALID, "mik_cgd_phase: no merged early part");
        const int grid = (int)std::min<int64_t>((it->n_send + MIK_BLOCK - 1) / MIK_BLOCK, mik_max_grid(ctx));
        hipLaunchKernelGGL((k_cgd_early<T>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, it->n_send, it->send_idx, (const T *)r, u, x, (const T *)&d->beta,
                           (const T *)&d->alpha, done, (const int *)&d->x_pending, bs.fuse_x ? 1 : 0, (T *)it->send_buf);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    case 6: {  // the x update that is still due when no head follows (end of a call without look-ahead)
        OpXFlush<T> op{u, x, coef_ptr<T>(&d->alpha), &d->x_pending};
        MIK_TRY((launch_map<T>(ctx, n, op, mik_aligned16(x) && mik_aligned16(u), (T *)nullptr, (const int *)nullptr)));
        hipLaunchKernelGGL((k_cg_clear_pending<T>), dim3(1), dim3(1), 0, ctx->stream, d);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    case 1:    // step B
        { CgProfileScope ps(&bs, 0); MIK_TRY(mik_spmv_launch<T>(ctx, bs.A, u, c, true, (T *)bs.seg_spmv, done)); }
        hipLaunchKernelGGL((k_cgd_fin_slot<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)bs.seg_spmv, nb, dot_slot, done, (FinScratch<T> *)bs.fin);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    case 4:    // step B1: the row-blocks that reference no halo column -- runs while the halo is in flight
        if (it->int_end <= it->int_begin) return mik_fail(ctx, MIK_ERR_INVALID, "mik_cgd_phase: no interior range set (mik_cgd_set_interior)");
        { CgProfileScope ps(&bs, 0); return mik_spmv_launch_range<T>(ctx, bs.A, u, c, true, (T *)bs.seg_spmv, done, (int)it->int_begin, (int)(it->int_end - it->int_begin)); }
    case 5:    // step B2: the row-blocks before and after the interior range, then the local dot(u, c) as in step B
        { CgProfileScope ps(&bs, 0); MIK_TRY(mik_spmv_launch_outside<T>(ctx, bs.A, u, c, true, (T *)bs.seg_spmv, done, (int)it->int_begin, (int)it->int_end)); }
        hipLaunchKernelGGL((k_cgd_fin_slot<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)bs.seg_spmv, nb, dot_slot, done, (FinScratch<T> *)bs.fin);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    case 2: {  // step C
        if (bs.fuse_x && (ctx->tuning[MIK_KNOB_CG_STEP] & 8) == 0) {   // alpha inside the sweep, its bookkeeping inside the finaliser (MIK_KNOB_CG_STEP bit 3: the separate k_cgd_alpha)
            OpCgUpdateR<T, CoefAlphaRanks<T>> up{r, c, CoefAlphaRanks<T>{(const T *)it->dot_all, it->nranks, &d->res}, cg_stream_hints(ctx, true, bs.A) >> 3};
            MIK_TRY((launch_map<T>(ctx, n, up, vec, (T *)bs.seg_vec, done)));
            hipLaunchKernelGGL((k_cgd_fin_slot_alpha<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)bs.seg_vec, nseg, rr_slot, done,
                               (FinScratch<T> *)bs.fin, (const T *)it->dot_all, it->nranks, d);
            MIK_LAUNCH_CHECK(ctx);
            return MIK_OK;
        }
        hipLaunchKernelGGL((k_cgd_alpha<T>), dim3(1), dim3(1), 0, ctx->stream, (const T *)it->dot_all, it->nranks, d);
        MIK_LAUNCH_CHECK(ctx);
        if (bs.fuse_x) {
            OpCgUpdateR<T> up{r, c, coef_ptr<T>(&d->alpha), cg_stream_hints(ctx, true, bs.A) >> 3};
            MIK_TRY((launch_map<T>(ctx, n, up, vec, (T *)bs.seg_vec, done)));
        } else {
            OpCgUpdate<T> up{x, r, u, c, coef_ptr<T>(&d->alpha), cg_stream_hints(ctx) >> 3};
            MIK_TRY((launch_map<T>(ctx, n, up, vec, (T *)bs.seg_vec, done)));
        }
        hipLaunchKernelGGL((k_cgd_fin_slot<T>), dim3(MIK_FIN_WGS), dim3(64), 0, ctx->stream, (const T *)bs.seg_vec, nseg, rr_slot, done, (FinScratch<T> *)bs.fin);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    case 3:    // step D
        if (it->hist_total >= bs.hist_cap) return mik_fail(ctx, MIK_ERR_INVALID, "mik_cgd_phase: more than %lld steps enqueued without mik_cgd_wait", (long long)bs.hist_cap);
        it->hist_total += 1;
        bs.seq += 1;
        hipLaunchKernelGGL((k_cgd_fin_res<T>), dim3(1), dim3(1), 0, ctx->stream, (const T *)it->rr_all, it->nranks, d, (T *)bs.hist,
                           (long long)(iteration + 1), (long long)bs.maxiter, bs.mirror, bs.seq, (int)(it->hist_total - 1), bs.fuse_x ? 1 : 0);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    case 20: {  // scaled norm of r over the partition, pass 1: this rank's max |r_i| -> rr_all[rank]   (mik_safe_norm_slow, per rank)
        const int grid = (int)std::min<int64_t>((std::max<int64_t>(n, 1) + MIK_BLOCK - 1) / MIK_BLOCK, 1024);
        MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(std::max<int64_t>(nseg, grid), 1)));
        hipLaunchKernelGGL((k_amax<T>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, (const T *)r, (T *)ctx->partials);
        MIK_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL((k_amax<T>), dim3(1), dim3(MIK_BLOCK), 0, ctx->stream, (int64_t)grid, (const T *)ctx->partials, rr_slot);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    case 22: {  // pass 2: this rank's tree sum of (r_i * s)^2, s = the common power of two (it->norm_scale) -> rr_all[rank]
        MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(nseg, 1)));
        OpScaledSq<T> op{(const T *)r, (T)it->norm_scale};
        MIK_TRY((launch_map<T>(ctx, n, op, mik_aligned16(r), (T *)ctx->partials, nullptr)));
        hipLaunchKernelGGL((k_finalize_store<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, nseg, (int64_t)0, rr_slot,
                           (const int *)nullptr);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    case 23:    // the frozen step closes with the scaled norm (it->norm_res), as k_cg_fix_res does on one GPU
        bs.seq += 1;
        it->hist_total = it->norm_fix_index + 1;
        hipLaunchKernelGGL((k_cg_fix_res<T>), dim3(1), dim3(1), 0, ctx->stream, d, (T)it->norm_res, (T *)bs.hist, (long long)it->norm_it_next,
                           (long long)bs.maxiter, bs.mirror, bs.seq, it->norm_fix_index);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    case 13: {  // step B2 without the finaliser (the mailbox transport finalises and exchanges in one kernel: csrc/mik_comm.hip)
        CgProfileScope ps(&bs, 0);
        return mik_spmv_launch_outside<T>(ctx, bs.A, u, c, true, (T *)bs.seg_spmv, done, (int)it->int_begin, (int)it->int_end);
    }
    case 14: {  // step B without the finaliser
        CgProfileScope ps(&bs, 0);
        return mik_spmv_launch<T>(ctx, bs.A, u, c, true, (T *)bs.seg_spmv, done);
    }
    case 16: {  // step C without alpha formation and without the finaliser: alpha is the stored scalar (k_cgd_fin_dot_mail)
        if (bs.fuse_x) {
            OpCgUpdateR<T> up{r, c, coef_ptr<T>(&d->alpha), cg_stream_hints(ctx, true, bs.A) >> 3};
            return launch_map<T>(ctx, n, up, vec, (T *)bs.seg_vec, done);
        }
        OpCgUpdate<T> up{x, r, u, c, coef_ptr<T>(&d->alpha), cg_stream_hints(ctx) >> 3};
        return launch_map<T>(ctx, n, up, vec, (T *)bs.seg_vec, done);
    }
    case 25:    // the pending x update of a frozen step has been applied (by a no-op head, the head ahead or phase 6): drop the flag
        hipLaunchKernelGGL((k_cg_clear_pending<T>), dim3(1), dim3(1), 0, ctx->stream, d);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    case 24:    // cg_iterator! closes with the scaled norm
        bs.seq += 1;
        hipLaunchKernelGGL((k_cgd_fix_init<T>), dim3(1), dim3(1), 0, ctx->stream, d, (T)it->norm_res, (T)it->reltol, (T)it->abstol, (long long)bs.maxiter,
                           bs.mirror, bs.seq);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    default:
        return mik_fail(ctx, MIK_ERR_INVALID, "mik_cgd_phase: unknown phase %d", phase);
    }
}

extern "C" int mik_cgd_phase(mik_cgd *it, int phase, int64_t iteration)
{
    if (!it || iteration < 0) return MIK_ERR_INVALID;
    return it->base.dtype == MIK_F64 ? cgd_phase_impl<double>(it, phase, iteration) : cgd_phase_impl<float>(it, phase, iteration);
}

int cgd_wait_raw(mik_cgd *it, CgMirror *m)
{
    MIK_TRY(cg_wait_mirror(&it->base));
    *m = *it->base.mirror;
    return MIK_OK;
}

int cgd_collect(mik_cgd *it, const CgMirror &m, double *residual, double *tol, int *done, double *history, int64_t cap, int64_t *steps)
{
    mik_cg &bs = it->base;
    mik_ctx *ctx = bs.ctx;
    const int64_t nd = m.nhist;
    if (history && nd == 1 && cap >= 1) {
        history[0] = m.res;                                   // the mirror carries the only residual: no copy
    } else if (history && nd > 0) {
        const int64_t take = std::min<int64_t>(nd, cap);
        const size_t es = mik_dtype_size(bs.dtype);
        std::vector<unsigned char> tmp((size_t)take * es);
        MIK_HIP(ctx, hipMemcpyAsync(tmp.data(), bs.hist, es * (size_t)take, hipMemcpyDeviceToHost, ctx->stream));
        MIK_HIP(ctx, mik_wait(ctx));
        for (int64_t j = 0; j < take; ++j)
            history[j] = bs.dtype == MIK_F64 ? ((const double *)tmp.data())[j] : (double)((const float *)tmp.data())[j];
    }
    if (bs.profile) (void)cg_profile_collect(&bs);           // the bracketed SpMV launches of the steps the host has now seen (mik_cgd_profile)
    // start a fresh history window for the next batch of steps (the device is idle: the host owns the mirror)
    bs.mirror->nhist = 0;
    it->hist_total = 0;
    bs.residual = m.res;
    bs.prev_residual = m.prev_res;
    if (m.tol_valid) bs.tol = m.tol;
    bs.mv_products += nd;
    if (residual) *residual = m.res;
    if (tol) *tol = bs.tol;
    if (done) *done = m.done;
    if (steps) *steps = nd;
    return MIK_OK;
}

// Block until every phase enqueued so far has run; returns the scalars of the last step and the
// residuals recorded since the previous wait (history[0..*steps-1], at most `cap`).  For hosts that drive the phases and the
// exchanges themselves: a norm outside the safe range is reported as MIK_ERR_RANGE (the step's x and r are final and x is
// flushed); mik_cgd_init / mik_cgd_iterate_many and the group calls recompute it with a common scale instead.
extern "C" int mik_cgd_wait(mik_cgd *it, double *residual, double *tol, int *done, double *history, int64_t cap, int64_t *steps)
{
    if (!it) return MIK_ERR_INVALID;
    CgMirror m;
    MIK_TRY(cgd_wait_raw(it, &m));
    if (m.range) {
        mik_ctx *ctx = it->base.ctx;
        if (it->base.fuse_x && it->initialised) {              // the frozen step's x .+= alpha .* u must not stay pending
            (void)mik_cgd_phase(it, 6, 0);
            (void)mik_wait(ctx);
        }
        return mik_fail(ctx, MIK_ERR_RANGE, "cg (row-partitioned, host-driven phases): |r|^2 left the range of a safe norm; use mik_cgd_init / "
                                            "mik_cgd_iterate_many (or the group calls), which rescale across the ranks");
    }
    return cgd_collect(it, m, residual, tol, done, history, cap, steps);
}

// =============================================================================================
// helpers for BiCGStab(l) (src/bicgstabl.jl): batched dot to the host, small dense LU solve
// =============================================================================================

// h = V[:, 1:k]' * w -- mul!(h, adjoint(V), w) (src/orthogonalize.jl:15) and, column by column, the
// Gram matrix mul!(M, adjoint(rs), rs) of src/bicgstabl.jl:121.  h: HOST array of k scalars.
template <typename T> static int gemv_t_impl(mik_ctx *ctx, int64_t n, int k, const T *V, int64_t ldv, const T *w, T *h_host)
{
    if (k == 0) return MIK_OK;
    if ((size_t)k * sizeof(T) > mik_ctx::COEF_SAFE_SLOT) return mik_fail(ctx, MIK_ERR_NOTIMPL, "mik_gemv_t: k = %d too large", k);
    const int64_t nseg = mik_nseg<T>(n);
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(nseg, 1) * (size_t)k));
    MIK_TRY(multidot<T>(ctx, n, k, V, ldv, w, (T *)ctx->coef));
    return coef_download<T>(ctx, 0, h_host, k);
}

extern "C" int mik_gemv_t(mik_ctx *ctx, int dtype, int64_t n, int k, const void *V, int64_t ldv, const void *w, void *h)
{
    if (!ctx || n < 0 || k < 0 || (k && (!h || !V || ldv < n)) || (n && k && !w)) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return gemv_t_impl<double>(ctx, n, k, (const double *)V, ldv, (const double *)w, (double *)h);
    if (dtype == MIK_F32) return gemv_t_impl<float>(ctx, n, k, (const float *)V, ldv, (const float *)w, (float *)h);
    return MIK_ERR_INVALID;
}

// M = V' * V for k <= 5 columns in one pass over V -- src/bicgstabl.jl:120 (M = rs' * rs)
template <typename T, int K> static int gram_launch(mik_ctx *ctx, int64_t n, const T *V, int64_t ldv)
{
    const int64_t nseg = mik_nseg<T>(n);
    const int grid = (int)std::min<int64_t>(nseg, mik_max_grid(ctx));
    const bool vec = mik_aligned16(V) && (ldv % VT<T>::W == 0);
    if (vec) hipLaunchKernelGGL((k_gram<T, true, K>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, V, ldv, (T *)ctx->partials);
    else hipLaunchKernelGGL((k_gram<T, false, K>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, V, ldv, (T *)ctx->partials);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

template <typename T> static int gram_impl(mik_ctx *ctx, int64_t n, int k, const T *V, int64_t ldv, T *M)
{
    const int np = k * (k + 1) / 2;
    const int64_t nseg = mik_nseg<T>(n);
    T *hd = (T *)ctx->coef;
    if (nseg == 0) {
        for (int i = 0; i < k * k; ++i) M[i] = T(0);
        return MIK_OK;
    }
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)nseg * (size_t)np));
    switch (k) {
    case 1: MIK_TRY((gram_launch<T, 1>(ctx, n, V, ldv))); break;
    case 2: MIK_TRY((gram_launch<T, 2>(ctx, n, V, ldv))); break;
    case 3: MIK_TRY((gram_launch<T, 3>(ctx, n, V, ldv))); break;
    case 4: MIK_TRY((gram_launch<T, 4>(ctx, n, V, ldv))); break;
    default: MIK_TRY((gram_launch<T, 5>(ctx, n, V, ldv))); break;
    }
    MIK_TRY(finalize_store<T>(ctx, nseg, np, hd));
    std::vector<T> out((size_t)np);
    MIK_TRY(coef_download<T>(ctx, 0, out.data(), np));
    int p = 0;
    for (int r = 0; r < k; ++r)
        for (int c = r; c < k; ++c) { M[(size_t)c * k + r] = out[p]; M[(size_t)r * k + c] = out[p]; ++p; }
    return MIK_OK;
}

extern "C" int mik_gram(mik_ctx *ctx, int dtype, int64_t n, int k, const void *V, int64_t ldv, void *M)
{
    if (!ctx || n < 0 || k < 1 || !M || (n && (!V || ldv < n))) return MIK_ERR_INVALID;
    if (k > 5) return mik_fail(ctx, MIK_ERR_NOTIMPL, "mik_gram: k = %d (at most 5 columns; use mik_gemv_t per column)", k);
    if (dtype == MIK_F64) return gram_impl<double>(ctx, n, k, (const double *)V, ldv, (double *)M);
    if (dtype == MIK_F32) return gram_impl<float>(ctx, n, k, (const float *)V, ldv, (float *)M);
    return MIK_ERR_INVALID;
}

// src/bicgstabl.jl:127-132 in one sweep (k_bicg_mr); *out = norm(rs[:, 1])
template <typename T>
static int bicg_mr_impl(mik_ctx *ctx, int64_t n, int l, T *us, int64_t ldu, T *rs, int64_t ldr, T *x, const T *gamma, T *out)
{
    const int64_t nseg = mik_nseg<T>(n);
    if (nseg == 0) { *out = T(0); return MIK_OK; }
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)nseg));
    BicgGamma<T> gm{};
    for (int j = 0; j < l; ++j) gm.g[j] = gamma[j];
    const int grid = (int)std::min<int64_t>(nseg, mik_max_grid(ctx));
    const bool vec = mik_aligned16(us) && mik_aligned16(rs) && mik_aligned16(x) && (ldu % VT<T>::W == 0) && (ldr % VT<T>::W == 0);
    if (vec) hipLaunchKernelGGL((k_bicg_mr<T, true>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, l, us, ldu, rs, ldr, x, gm, (T *)ctx->partials, (const T *)nullptr);
    else hipLaunchKernelGGL((k_bicg_mr<T, false>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, nseg, l, us, ldu, rs, ldr, x, gm, (T *)ctx->partials, (const T *)nullptr);
    MIK_LAUNCH_CHECK(ctx);
    MIK_TRY(finalize_store<T>(ctx, nseg, 1, (T *)ctx->coef));
    T ss;
    MIK_TRY(coef_download<T>(ctx, 0, &ss, 1));
    if (mik_nrm_in_range(ss)) { *out = (T)std::sqrt(ss); return MIK_OK; }
    return mik_safe_norm_slow<T>(ctx, n, rs, out);                       // norm(rs[:, 1]) of a badly scaled residual
}

extern "C" int mik_bicgstab_mr_update(mik_ctx *ctx, int dtype, int64_t n, int l, void *us, int64_t ldu, void *rs, int64_t ldr, void *x,
                                      const void *gamma, void *out)
{
    if (!ctx || n < 0 || l < 1 || l > 8 || !gamma || !out || (n && (!us || !rs || !x || ldu < n || ldr < n))) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return bicg_mr_impl<double>(ctx, n, l, (double *)us, ldu, (double *)rs, ldr, (double *)x, (const double *)gamma, (double *)out);
    if (dtype == MIK_F32) return bicg_mr_impl<float>(ctx, n, l, (float *)us, ldu, (float *)rs, ldr, (float *)x, (const float *)gamma, (float *)out);
    return MIK_ERR_INVALID;
}

// Solve A x = b for a small dense column-major A (n x n, leading dimension lda) by LU with partial
// pivoting -- F = lu!(view(M, L, L)); ldiv!(gamma, F, view(M, L, 1)) at src/bicgstabl.jl:124-125.
// A is overwritten by its factors, b by the solution.  Returns MIK_ERR_SINGULAR on an exactly
// singular pivot (the reference throws SingularException).
template <typename T> static int lu_solve(T *A, int64_t lda, int n, T *b)
{
    auto at = [&](int i, int j) -> T & { return A[(size_t)j * lda + i]; };
    for (int j = 0; j < n; ++j) {
        int p = j;
        T best = std::fabs(at(j, j));
        for (int i = j + 1; i < n; ++i) { const T a = std::fabs(at(i, j)); if (a > best) { best = a; p = i; } }
        if (best == T(0)) return MIK_ERR_SINGULAR;
        if (p != j) {
            for (int c = 0; c < n; ++c) std::swap(at(j, c), at(p, c));
            std::swap(b[j], b[p]);
        }
        const T piv = at(j, j);
        for (int i = j + 1; i < n; ++i) {
            const T m = at(i, j) / piv;
            at(i, j) = m;
            for (int c = j + 1; c < n; ++c) at(i, c) = at(i, c) - m * at(j, c);
            b[i] = b[i] - m * b[j];
        }
    }
    for (int j = n - 1; j >= 0; --j) {
        b[j] = b[j] / at(j, j);
        const T t = b[j];
        for (int i = 0; i < j; ++i) b[i] = b[i] - t * at(i, j);
    }
    return MIK_OK;
}

extern "C" int mik_lu_solve(int dtype, void *A, int64_t lda, int n, void *b)
{
    if (!A || !b || n < 0 || lda < n) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return lu_solve<double>((double *)A, lda, n, (double *)b);
    if (dtype == MIK_F32) return lu_solve<float>((float *)A, lda, n, (float *)b);
    return MIK_ERR_INVALID;
}

// LinearAlgebra.givensAlgorithm(f, g) -> (c, s, r) on the host -- used at src/hessenberg.jl:24 and
// src/minres.jl:129.  out = {c, s, r} of dtype.
extern "C" int mik_givens(int dtype, const void *f, const void *g, void *out)
{
    if (!f || !g || !out) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) { double *o = (double *)out; givens_algorithm<double>(*(const double *)f, *(const double *)g, o[0], o[1], o[2]); return MIK_OK; }
    if (dtype == MIK_F32) { float *o = (float *)out; givens_algorithm<float>(*(const float *)f, *(const float *)g, o[0], o[1], o[2]); return MIK_OK; }
    return MIK_ERR_INVALID;
}
