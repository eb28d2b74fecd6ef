/* A plain-C host of libmik.so -- no Python, no PyTorch: the drop-in boundary as a C program would use it
 * (and as Julia's ccall does).  Builds the 3D 7-point Laplacian of test/laplace_matrix.jl as SparseMatrixCSC
 * fields (1-based Int64 colptr / rowval + nzval), the hashed right-hand side of SURVEY.md section 8d, runs
 *      x, history = cg(A, b; log = true)          (src/cg.jl:162, :209-242)
 * through mik_cg_create / mik_cg_iterate and prints every residual as a C99 hex float, one per line, followed by
 * "iters <k> converged <0|1>".  Exit status 3 = no usable HIP device (nothing computed).
 *
 *      gcc -std=c99 -Wall -pedantic -I include tests/c_host/cg_host.c -o cg_host -L iterativesolvers.jl_amd -l:libmik.so
 *      ./cg_host 16            (grid points per dimension)
 *      ./cg_host 12 gmres      x, history = gmres(A, b; restart = 10, log = true) on the same operator (src/gmres.jl:143,184-222)
 *      ./cg_host 12 cgop       cg with the operator handed over as a C CALLBACK (mik_cg_create_op: "any A with mul!",
 *                              docs/src/getting_started.md:25-30) and all iterations in ONE call (mik_cg_iterate_many)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mik.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != MIK_OK) {                                                                 \
            fprintf(stderr, "%s -> status %d: %s\n", #call, rc_, mik_last_error(ctx));      \
            return rc_ == MIK_ERR_HIP && !ctx ? 3 : 1;                                       \
        }                                                                                    \
    } while (0)

/* mul!(y, A, x) of a user-defined operator: here it forwards to the library's own SpMV, a Julia host would pass an
 * @cfunction around its LinearMap */
struct my_operator { mik_ctx *ctx; const mik_csr *A; long calls; };
static int my_mul(void *user, const void *x, void *y)
{
    struct my_operator *op = (struct my_operator *)user;
    op->calls += 1;
    return mik_spmv(op->ctx, op->A, x, y);
}

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 16;
    const int64_t n = N * N * N, nnz = 7 * n - 6 * N * N;
    mik_ctx *ctx = NULL;
    CHECK(mik_ctx_create(0, &ctx));
    {   /* the machine the library found (mik_ctx_info): nothing in it is assumed */
        mik_device_info di;
        CHECK(mik_ctx_info(ctx, &di));
        printf("machine %s cus %d xcds %d wave %d lds %lld xcd_maps %d gs_cap %d\n", di.arch, di.compute_units, di.xcds, di.wavefront_size,
               (long long)di.lds_bytes_per_cu, di.xcd_maps, di.resident_workgroup_cap);
    }

    /* laplace_matrix(Float64, N, 3): column j = x + N (y + N z); symmetric, so column j lists rows j -+ N^2, j -+ N, j -+ 1, j */
    int64_t *colptr = malloc(sizeof(int64_t) * (size_t)(n + 1)), *rowval = malloc(sizeof(int64_t) * (size_t)nnz);
    double *nzval = malloc(sizeof(double) * (size_t)nnz), *b = malloc(sizeof(double) * (size_t)n);
    if (!colptr || !rowval || !nzval || !b) return 1;
    int64_t k = 0;
    for (int64_t j = 0; j < n; ++j) {
        const int64_t x = j % N, y = (j / N) % N, z = j / (N * N);
        colptr[j] = k + 1;
        if (z > 0)     { rowval[k] = j - N * N + 1; nzval[k++] = -1.0; }
        if (y > 0)     { rowval[k] = j - N + 1;     nzval[k++] = -1.0; }
        if (x > 0)     { rowval[k] = j - 1 + 1;     nzval[k++] = -1.0; }
        rowval[k] = j + 1; nzval[k++] = 6.0;
        if (x < N - 1) { rowval[k] = j + 1 + 1;     nzval[k++] = -1.0; }
        if (y < N - 1) { rowval[k] = j + N + 1;     nzval[k++] = -1.0; }
        if (z < N - 1) { rowval[k] = j + N * N + 1; nzval[k++] = -1.0; }
    }
    colptr[n] = k + 1;
    if (k != nnz) { fprintf(stderr, "nnz mismatch\n"); return 1; }
    for (int64_t i = 1; i <= n; ++i) b[i - 1] = (double)(((uint64_t)i * 2654435761ull) & 0xffffffffull) / 4294967296.0 - 0.5;

    mik_csr *A = NULL;
    CHECK(mik_csr_create(ctx, MIK_F64, n, n, nnz, colptr, rowval, nzval, 1 /* index_base */, 1 /* is_csc */, &A));
    int layout = -1;
    CHECK(mik_csr_layout(A, &layout));

    const size_t bytes = sizeof(double) * (size_t)n;
    void *dx, *db, *du, *dr, *dc;
    CHECK(mik_malloc(ctx, bytes, &dx)); CHECK(mik_malloc(ctx, bytes, &db)); CHECK(mik_malloc(ctx, bytes, &du));
    CHECK(mik_malloc(ctx, bytes, &dr)); CHECK(mik_malloc(ctx, bytes, &dc));
    CHECK(mik_memcpy_h2d(ctx, db, b, bytes));
    const double zero = 0.0;
    CHECK(mik_fill(ctx, MIK_F64, n, &zero, dx));                       /* zerox(A, b): src/common.jl:18-23 */

    if (argc > 2 && strcmp(argv[2], "gmres") == 0) {
        mik_gmres *g = NULL;                                            /* gmres_iterable!(x, A, b; restart = 10, initially_zero = true) */
        CHECK(mik_gmres_create(ctx, A, dx, db, NULL, NULL, 0.0, 1.4901161193847656e-8, 10, n, 1, MIK_MGS, &g));
        int64_t iteration = 0;
        for (;;) {
            double residual;
            int done;
            CHECK(mik_gmres_iterate(g, iteration, &residual, &done));
            if (done) break;
            printf("%a\n", residual);
            ++iteration;
        }
        double res, tol, beta;
        int kk, conv;
        int64_t mv;
        CHECK(mik_gmres_state(g, &res, &tol, &beta, &kk, &mv, &conv));
        printf("iters %lld converged %d layout %d mvps %lld\n", (long long)iteration, conv, layout, (long long)mv);
        double *xg = malloc(bytes);
        CHECK(mik_memcpy_d2h(ctx, xg, dx, bytes));
        double sg = 0.0;
        for (int64_t i = 0; i < n; ++i) sg += xg[i];
        printf("sum_x %a\n", sg);
        CHECK(mik_gmres_destroy(g));
        CHECK(mik_csr_destroy(A));
        CHECK(mik_ctx_destroy(ctx));
        return 0;
    }

    if (argc > 2 && strcmp(argv[2], "cgop") == 0) {
        struct my_operator mine = {ctx, A, 0};
        mik_operator op = {MIK_F64, n, NULL, my_mul, &mine};
        mik_cg *itop = NULL;
        CHECK(mik_cg_create_op(ctx, &op, NULL /* Pl = Identity() */, dx, db, du, dr, dc, 0.0, 1.4901161193847656e-8, n, 1, &itop));
        double *hist = malloc(sizeof(double) * (size_t)n);
        int64_t steps = 0;
        CHECK(mik_cg_iterate_many(itop, 0, n, hist, &steps));           /* the device-side stopping test ends the batch */
        for (int64_t i = 0; i < steps; ++i) printf("%a\n", hist[i]);
        double res, prev, tol;
        int64_t maxiter, mv;
        int conv;
        CHECK(mik_cg_state(itop, &res, &prev, &tol, &maxiter, &mv, &conv));
        printf("iters %lld converged %d layout %d mvps %lld\n", (long long)steps, conv, layout, (long long)mv);
        printf("callback_calls %ld\n", mine.calls);
        double *xo = malloc(bytes);
        CHECK(mik_memcpy_d2h(ctx, xo, dx, bytes));
        double so = 0.0;
        for (int64_t i = 0; i < n; ++i) so += xo[i];
        printf("sum_x %a\n", so);
        CHECK(mik_cg_destroy(itop));
        CHECK(mik_csr_destroy(A));
        CHECK(mik_ctx_destroy(ctx));
        return 0;
    }

    mik_cg *it = NULL;                                                  /* cg_iterator!(x, A, b; initially_zero = true) */
    CHECK(mik_cg_create(ctx, A, dx, db, du, dr, dc, NULL, 0.0 /* abstol */, 1.4901161193847656e-8 /* sqrt(eps) */, n /* maxiter */,
                        1, &it));
    int64_t iteration = 0;
    for (;;) {                                                          /* for (iteration, item) in enumerate(iterable) */
        double residual;
        int done;
        CHECK(mik_cg_iterate(it, iteration, &residual, &done));
        if (done) break;
        printf("%a\n", residual);
        ++iteration;
    }
    double res, prev, tol;
    int64_t maxiter, mv;
    int conv;
    CHECK(mik_cg_state(it, &res, &prev, &tol, &maxiter, &mv, &conv));
    printf("iters %lld converged %d layout %d mvps %lld\n", (long long)iteration, conv, layout, (long long)mv);

    double *x = malloc(bytes);
    CHECK(mik_memcpy_d2h(ctx, x, dx, bytes));
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += x[i];
    printf("sum_x %a\n", s);

    CHECK(mik_cg_destroy(it));
    CHECK(mik_csr_destroy(A));
    CHECK(mik_free(ctx, dx)); CHECK(mik_free(ctx, db)); CHECK(mik_free(ctx, du)); CHECK(mik_free(ctx, dr)); CHECK(mik_free(ctx, dc));
    CHECK(mik_ctx_destroy(ctx));
    free(colptr); free(rowval); free(nzval); free(b); free(x);
    return 0;
}
