/*
 * mik_oracle.c -- CPU oracle for the cg! / gmres! hot path of IterativeSolvers.jl v0.9.4.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain-C restatement of the reference
 * algorithm (Julia cannot run in this environment: no `julia` binary, no network), used as the
 * checker by tests/, by __graft_entry__.smoke() and as bench.py's `cpu_baseline` ("port") leg.
 * The product path (libmik.so, HIP) never links, imports or calls it.
 *
 * Pinning status: PARITY PARTIALLY PINNED.  The reference's tests hold no stored residual
 * histories; its random inputs come from Julia's RNG and cannot be regenerated outside Julia
 * (test/cg.jl:22, test/gmres.jl:13).  What the reference DOES pin for this path -- the literal
 * Hessenberg matrices of test/hessenberg.jl:10-26, the identity case of test/gmres.jl:68-73, the
 * tridiagonal termination cases of test/cg.jl:98-122 and test/gmres.jl:75-99, the zero-rhs case of
 * test/cg.jl:49-51 and the invariants of test/orthogonalize.jl:25-34 -- is checked against this
 * oracle in tests/test_oracle_pinning.py.  Bit-level residual histories are "parity unpinned".
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: Julia does not contract a*b+c).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SEQ 0
#define ORC_PAIR 1
#define ORC_TREE 2
#define ORC_BLAS 3

#define ORC_MGS 0
#define ORC_CGS 1
#define ORC_DGKS 2

/* optional row partition for ORC_TREE reductions over full-length vectors (multi-GPU layout) */
#define ORC_MAX_PARTS 64
static int g_orc_nparts = 0;
static int64_t g_orc_part[ORC_MAX_PARTS + 1];
int orc_set_partition(int nparts, const int64_t *offsets)
{
    if (nparts < 0 || nparts > ORC_MAX_PARTS) return 1;
    g_orc_nparts = nparts;
    for (int i = 0; i <= nparts && nparts > 0; ++i) g_orc_part[i] = offsets[i];
    return 0;
}

/* optional device row-sum shape for long rows (0 = off: every row strictly sequential) */
static int64_t g_orc_long_row = 0;
static int64_t g_orc_long_seg = 0;     /* rows longer than this are summed segment by segment (0 = never cut) */
static int64_t g_orc_long_grp = 1;     /* entries per lane and group of the long-row shape (mik_spmv_long_group(); 1 = lane-strided single entries) */
void orc_set_long_group(int64_t group) { g_orc_long_grp = group > 0 ? group : 1; }
void orc_set_long_row(int64_t threshold) { g_orc_long_row = threshold > 0 ? threshold : 0; }
void orc_set_long_segment(int64_t segment) { g_orc_long_seg = segment > 0 ? segment : 0; }

/* ORC_BLAS: entry points of the host's OpenBLAS (CBLAS interface, 32-bit ints), bound by oracle/orc.py from the
 * library NumPy / SciPy ship -- the same library family LinearAlgebra.dot / norm / mul! reach in the reference. */
typedef double (*orc_ddot_fn)(int, const double *, int, const double *, int);
typedef double (*orc_dnrm2_fn)(int, const double *, int);
typedef void (*orc_dgemv_fn)(int, int, int, int, double, const double *, int, const double *, int, double, double *, int);
typedef float (*orc_sdot_fn)(int, const float *, int, const float *, int);
typedef float (*orc_snrm2_fn)(int, const float *, int);
typedef void (*orc_sgemv_fn)(int, int, int, int, float, const float *, int, const float *, int, float, float *, int);
static orc_ddot_fn g_blas_dot_f64;
static orc_dnrm2_fn g_blas_nrm2_f64;
static orc_dgemv_fn g_blas_gemv_f64;
static orc_sdot_fn g_blas_dot_f32;
static orc_snrm2_fn g_blas_nrm2_f32;
static orc_sgemv_fn g_blas_gemv_f32;
int orc_set_blas(void *ddot, void *dnrm2, void *dgemv, void *sdot, void *snrm2, void *sgemv)
{
    if (!ddot || !dnrm2 || !dgemv || !sdot || !snrm2 || !sgemv) return 1;
    g_blas_dot_f64 = (orc_ddot_fn)ddot; g_blas_nrm2_f64 = (orc_dnrm2_fn)dnrm2; g_blas_gemv_f64 = (orc_dgemv_fn)dgemv;
    g_blas_dot_f32 = (orc_sdot_fn)sdot; g_blas_nrm2_f32 = (orc_snrm2_fn)snrm2; g_blas_gemv_f32 = (orc_sgemv_fn)sgemv;
    return 0;
}
int orc_have_blas(void) { return g_blas_dot_f64 != 0; }

/* ---- fp64 instantiation ---- */
#define T double
#define F(x) x##_f64
#define SQRT_f64 sqrt
#define HYPOT_f64 hypot
#define FABS_f64 fabs
#define POW_f64 pow
#define LOG_f64 log
#define EPS_f64 DBL_EPSILON /* Julia's eps(Float64) = 2^-52: LinearAlgebra.floatmin2 uses it, not LAPACK's unit roundoff */
#define TMIN_f64 DBL_MIN
#define TMAX_f64 DBL_MAX
#define NRM_LO_f64 0x1p-900     /* safe range of a sum of squares, see safe_nrm_ */
#define NRM_HI_f64 0x1p+900
#define NRM_EC_f64 1022
#include "orc_impl.inc"
#undef T
#undef F

/* ---- fp32 instantiation ---- */
#define T float
#define F(x) x##_f32
#define SQRT_f32 sqrtf
#define HYPOT_f32 hypotf
#define FABS_f32 fabsf
#define POW_f32 powf
#define LOG_f32 logf
#define EPS_f32 FLT_EPSILON
#define TMIN_f32 FLT_MIN
#define TMAX_f32 FLT_MAX
#define NRM_LO_f32 0x1p-70f
#define NRM_HI_f32 0x1p+100f
#define NRM_EC_f32 126
#include "orc_impl.inc"
#undef T
#undef F

int orc_abi_version(void) { return 1; }

/* ------------------------------------------------------------------------------------------ */
/* fixtures                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* nnz of laplace_matrix(T, N, dims): dims*N^dims stencil arms minus the clipped ones. */
int64_t orc_laplace_nnz(int64_t N, int dims)
{
    int64_t n = 1;
    for (int d = 0; d < dims; ++d) n *= N;
    /* each dimension contributes 2 neighbours per point except on its two faces */
    return n + (int64_t)dims * 2 * (n - n / N);
}

/* laplace_matrix(Float64, N, dims) -- test/laplace_matrix.jl:1-12.
 * D = SymTridiagonal(2, -1); A <- kron(A, I_N) + kron(I, D) per extra dimension, so the newest
 * dimension is the fastest index: row = c_dims + N*(c_{dims-1} + N*(...)).  Diagonal = 2*dims,
 * off-diagonals -1 at +-1 (not across line ends), +-N, +-N^2.  Emits SparseMatrixCSC arrays with
 * `index_base`-based Int64 indices, rows ascending within a column (as Julia stores them). */
void orc_laplace_csc(int64_t N, int dims, int index_base, int64_t *colptr, int64_t *rowval,
                     double *nzval)
{
    int64_t n = 1, stride[3] = {1, 1, 1};
    for (int d = 0; d < dims; ++d) { stride[d] = n; n *= N; }
    int64_t k = 0;
    double diag = 0.0;
    for (int d = 0; d < dims; ++d) diag = diag + 2.0;
    for (int64_t j = 0; j < n; ++j) {
        colptr[j] = k + index_base;
        for (int d = dims - 1; d >= 0; --d) {          /* rows below the diagonal, ascending */
            int64_t c = (j / stride[d]) % N;
            if (c > 0) { rowval[k] = j - stride[d] + index_base; nzval[k] = -1.0; ++k; }
        }
        rowval[k] = j + index_base; nzval[k] = diag; ++k;
        for (int d = 0; d < dims; ++d) {
            int64_t c = (j / stride[d]) % N;
            if (c < N - 1) { rowval[k] = j + stride[d] + index_base; nzval[k] = -1.0; ++k; }
        }
    }
    colptr[n] = k + index_base;
}

/* advection_dominated(; N, beta) -- benchmark/advection_diffusion.jl:3-30.
 * h = 1/(N+1); A = laplace_matrix(Float64, N, 3) ./ -h^2 + kron(I_{N^2}, dx1d),
 * dx1d = spdiagm(-1 => -beta/2h, 1 => beta/2h)  (`2h` binds tighter than `/`).
 * b[x + N*(y + N*z)] = f(xs[x], xs[y], xs[z]), f = exp(x*y*z)*sin(pi*x)*sin(pi*y)*sin(pi*z),
 * xs = range(0, stop=1, length=N+2)[2:N+1].  A is nonsymmetric, emitted as CSC like Julia. */
void orc_advdiff_csc(int64_t N, double beta, int index_base, int64_t *colptr, int64_t *rowval,
                     double *nzval, double *b)
{
    const int64_t n = N * N * N;
    const double h = 1.0 / (double)(N + 1);
    const double mh2 = -(h * h);
    const double lap_diag = 6.0 / mh2;
    const double lap_off = -1.0 / mh2;
    const double dx_sub = (-beta) / (2.0 * h);   /* A[i, i-1] */
    const double dx_sup = beta / (2.0 * h);      /* A[i, i+1] */
    const int64_t stride[3] = {1, N, N * N};
    int64_t k = 0;
    for (int64_t j = 0; j < n; ++j) {
        colptr[j] = k + index_base;
        for (int d = 2; d >= 0; --d) {
            int64_t c = (j / stride[d]) % N;
            if (c > 0) {
                /* entry (row i = j - stride, col j): for d == 0 it is A[i, i+1] = super-diagonal */
                rowval[k] = j - stride[d] + index_base;
                nzval[k] = (d == 0) ? lap_off + dx_sup : lap_off;
                ++k;
            }
        }
        rowval[k] = j + index_base; nzval[k] = lap_diag; ++k;
        for (int d = 0; d < 3; ++d) {
            int64_t c = (j / stride[d]) % N;
            if (c < N - 1) {
                /* entry (row i = j + stride, col j): for d == 0 it is A[i, i-1] = sub-diagonal */
                rowval[k] = j + stride[d] + index_base;
                nzval[k] = (d == 0) ? lap_off + dx_sub : lap_off;
                ++k;
            }
        }
    }
    colptr[n] = k + index_base;
    if (b) {
        const double pi = 3.14159265358979323846;
        for (int64_t z = 0; z < N; ++z)
            for (int64_t y = 0; y < N; ++y)
                for (int64_t x = 0; x < N; ++x) {
                    const double xs = (double)(x + 1) / (double)(N + 1);
                    const double ys = (double)(y + 1) / (double)(N + 1);
                    const double zs = (double)(z + 1) / (double)(N + 1);
                    double v = exp(xs * ys * zs);
                    v = v * sin(pi * xs);
                    v = v * sin(pi * ys);
                    v = v * sin(pi * zs);
                    b[x + N * (y + N * z)] = v;
                }
    }
}

/* Language-independent right-hand side:
 * b[i] = ((i * 2654435761) mod 2^32) / 2^32 - 0.5, i = 1..n, exact in uint64/double. */
void orc_hashed_rhs(int64_t n, double *b)
{
    for (int64_t i = 1; i <= n; ++i) {
        uint64_t hsh = ((uint64_t)i * 2654435761ULL) & 0xFFFFFFFFULL;
        b[i - 1] = (double)hsh / 4294967296.0 - 0.5;
    }
}
