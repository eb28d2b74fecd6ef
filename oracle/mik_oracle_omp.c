/*
 * mik_oracle_omp.c -- multi-threaded CPU baseline for bench.py's `cpu_baseline_omp` leg ONLY
 * (TEST / MEASUREMENT INFRASTRUCTURE, never on the product path, never used as a parity checker).
 *
 * BASELINE.md section 3 plans two CPU modes: `cpu_ref_serial` (what Julia does: oracle/mik_oracle.c,
 * mode SEQ) and `cpu_ref_omp`, a best-effort host baseline with the same algorithm and stopping rule
 * (src/cg.jl:43-66,120-155): row-parallel CSR SpMV with Int32 indices and OpenMP reductions on all
 * host cores.  Summation order differs from the serial oracle (per-thread partial sums), so only the
 * iteration count / convergence are checked against it, never bits.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int orc_omp_threads(void) { return omp_get_max_threads(); }
void orc_omp_set_threads(int t) { if (t > 0) omp_set_num_threads(t); }

/* cg! on a CSR matrix (rowptr/col Int32, 0-based), x0 = 0.  Returns iterations done. */
int64_t orc_omp_cg_f64(int64_t n, const int32_t *rowptr, const int32_t *col, const double *val, const double *b,
                       double *x, double abstol, double reltol, int64_t maxiter, double *resnorm)
{
    double *u = (double *)malloc(sizeof(double) * (size_t)n);
    double *r = (double *)malloc(sizeof(double) * (size_t)n);
    double *c = (double *)malloc(sizeof(double) * (size_t)n);
    /* NUMA first touch: the caller's arrays were written by one thread (numpy); give every thread a local
     * copy of its own rows of the operator before timing-relevant work starts */
    const int64_t nnz = rowptr[n];
    int32_t *rp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t *cl = (int32_t *)malloc(sizeof(int32_t) * (size_t)nnz);
    double *vl = (double *)malloc(sizeof(double) * (size_t)nnz);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        rp[i] = rowptr[i];
        u[i] = 0.0;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) { cl[k] = col[k]; vl[k] = val[k]; }
    }
    rp[n] = rowptr[n];
    rowptr = rp; col = cl; val = vl;
    double rr = 0.0;
#pragma omp parallel for reduction(+ : rr) schedule(static)
    for (int64_t i = 0; i < n; ++i) { x[i] = 0.0; r[i] = b[i]; rr += b[i] * b[i]; }
    double residual = sqrt(rr), prev = 1.0;
    const double tol = fmax(reltol * residual, abstol);
    int64_t it = 0;
    while (it < maxiter && residual > tol) {
        const double beta = (residual * residual) / (prev * prev);
        double uc = 0.0;
#pragma omp parallel
        {
#pragma omp for schedule(static)
            for (int64_t i = 0; i < n; ++i) u[i] = r[i] + beta * u[i];
#pragma omp for reduction(+ : uc) schedule(static)
            for (int64_t i = 0; i < n; ++i) {
                double s = 0.0;
                for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += val[k] * u[col[k]];
                c[i] = s;
                uc += u[i] * s;
            }
        }
        const double alpha = (residual * residual) / uc;
        rr = 0.0;
#pragma omp parallel for reduction(+ : rr) schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            x[i] += alpha * u[i];
            r[i] -= alpha * c[i];
            rr += r[i] * r[i];
        }
        prev = residual;
        residual = sqrt(rr);
        if (resnorm) resnorm[it] = residual;
        ++it;
    }
    free(u); free(r); free(c); free(rp); free(cl); free(vl);
    return it;
}
