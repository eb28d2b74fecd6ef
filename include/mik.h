/*
 * mik.h -- C ABI of libmik.so: the MI355X (gfx950) Krylov inner loop behind IterativeSolvers.jl's
 *          cg! / gmres! iteration path.
 *
 * This is the drop-in boundary.  IterativeSolvers.jl (v0.9.4, pure Julia) has no FFI of its own;
 * its "plugin mechanism" is multiple dispatch on the operator/vector types
 * (docs/src/getting_started.md:25-30).  A Julia host binds these entry points with `ccall` behind
 * `mul!`, `dot`, `norm`, broadcast and `iterate` methods for a device vector / CSR operator type
 * (INTEGRATION.md shows the binding); the Python harness in this repo binds the same symbols with
 * ctypes.  Each entry point cites the reference call it replaces (file:line relative to the
 * reference checkout).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only.  Every function returns an int status
 *    (MIK_OK = 0); no C++ exception crosses the boundary; mik_last_error() gives the text.
 *  - `dtype` is MIK_F64 or MIK_F32 (the arithmetic type of the whole path; indices are Int32 on
 *    the device).  Scalars cross the boundary as `const void*` / `void*` to a host value of that
 *    dtype, or as double where stated.
 *  - Vector arguments are raw DEVICE pointers (from mik_malloc, or from any HIP allocator of the
 *    same device -- e.g. a torch tensor's data_ptr or an AMDGPU.jl ROCArray).  16-byte aligned
 *    pointers take the vectorised kernels, others a scalar-load variant with identical results.
 *  - Host pointers are borrowed for the duration of the call.  Calls that return a scalar to the
 *    host synchronise the context's stream; everything else is asynchronous on that stream.
 *  - One mik_ctx = one device + one HIP stream; calls on a ctx are not re-entrant (the reference
 *    is single-threaded).  Multi-GPU = one process (and one ctx) per GPU.
 *
 * Reduction semantics (what makes results reproducible and checkable bit-for-bit)
 *  Every dot / norm on the device is a fixed-shape two-level tree that depends only on (n, dtype):
 *   level 1: the vector is cut into segments of 256*W*L elements (mik_reduce_shape()); virtual
 *            thread t of a segment sums its elements e -> (e/W)*(256*W) + W*t + e%W in ascending
 *            e, then a wave-64 shuffle-down tree (offsets 32..1), then the 4 wave sums left to
 *            right;  the dot fused into the SpMV (CG's dot(u, c)) uses W = 1 (one row per
 *            thread) and L = mik_spmv_dot_shape() consecutive 256-row blocks per segment;
 *   level 2: 1024 virtual threads; thread t sums segment sums t, t+1024, ... ascending, wave tree
 *            per 64, then the 16 wave sums left to right.
 *  Multiply and add are never contracted into an FMA (the library is built -ffp-contract=off) and
 *  the SpMV sums each row serially in ascending column order, exactly like the reference's CSC
 *  column scatter (rows longer than mik_spmv_long_row() entries: one wave per row segment, lane l sums the
 *  products of the groups l, l+64, ... of 4 consecutive entries in order, then the wave tree: mik_spmv_long_row below).  oracle/mik_oracle.c (mode ORC_TREE) restates the same tree on the CPU.
 *
 * Norms (the reference's norm() is BLAS nrm2 / generic_norm2: over- and underflow-safe)
 *  norm(x) = sqrt(t), t = tree sum of x_i^2, whenever t lies in [2^-900, 2^900] (fp32: [2^-70, 2^100]) -- then no
 *  square that matters has underflowed and nothing has overflowed.  Outside that range (t = 0, denormal, Inf or NaN
 *  although x may be finite and non-zero: a badly scaled system) the norm is recomputed with scaling: amax = max |x_i|
 *  (0, Inf, NaN are returned as they are); s = 2^-e with amax = f * 2^e, f in [0.5, 1), e clamped to +-1022 (+-126);
 *  t' = the same tree over (x_i * s)^2; norm = sqrt(t') / s.  mik_nrm2, the fused sweeps and the single-GPU iterables
 *  (CG residual, GMRES beta and the Gram-Schmidt norms) all do this, so a solve on a system scaled by 1e-200 takes
 *  the iterations of the unscaled one instead of "converging" on a residual that underflowed to 0.  The row-partitioned
 *  iterables do the same ACROSS the ranks: every rank sees the same out-of-range total, the ranks exchange their local max |x_i|
 *  (mik_cgd_init / mik_cgd_iterate_many / the group calls: one more scalar gather; mik_gmres_create_partitioned: through the
 *  reduce() callback, every rank contributing its value in its own slot of a zero vector), scale by the common power of two,
 *  add the local tree sums in rank order and divide -- so switching to N GPUs does not change what converges.  Only the legacy
 *  host-driven phases (mik_cgd_phase + mik_cgd_wait) still report MIK_ERR_RANGE.  The oracle mirrors it.
 */
#ifndef MIK_H
#define MIK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIK_ABI_VERSION 6   /* 6 (round 6): mik_ctx_info / mik_device_info (the machine is queried, not assumed); mik_plink_exchange;
                             *   mik_comm_allgather_sum also over a mailbox-only communicator.
                             *   5 (round 5): mik_partition gained `link`; the pushed halo lands in library-owned buffers (mik_plink_*, mik_cgd_ghost_export;
                             *   mik_cgd_connect_ghosts takes ghost counts instead of byte offsets; mik_mem_export is gone); mik_cgd_profile.
                             *   4 (round 4): MIK_ERR_SINGULAR replaces MIK_ERR_INVALID for an exactly singular pivot (mik_lu_solve, mik_bicgstab_step);
                             *   the scalar mailbox transport (mik_mailbox_*).  3 (round 3): mik_csr_pack and the knob setters left this header
                             *   (include/mik_dev.h) */

/* status codes */
enum {
    MIK_OK = 0,
    MIK_ERR_INVALID = 1,     /* invalid argument */
    MIK_ERR_HIP = 2,         /* HIP runtime error (text in mik_last_error) */
    MIK_ERR_MISMATCH = 3,    /* dimension / dtype mismatch */
    MIK_ERR_NOMEM = 4,       /* out of (device or host) memory */
    MIK_ERR_NOTIMPL = 5,     /* not implemented (e.g. nnz >= 2^31) */
    MIK_ERR_CALLBACK = 6,    /* a mik_partition / operator / preconditioner callback returned non-zero */
    MIK_ERR_RANGE = 7,       /* a norm left the safe range while the HOST drives the phases of a row-partitioned step itself (see "Norms") */
    MIK_ERR_SINGULAR = 8     /* lu! met an exactly singular pivot (the reference throws SingularException, src/bicgstabl.jl:124): mik_lu_solve,
                              * mik_bicgstab_step -- a code of its own, so that an invalid handle is never reported as a singular matrix */
};

enum { MIK_F64 = 0, MIK_F32 = 1 };

/* orthogonalisation methods -- src/orthogonalize.jl:4-7 */
enum { MIK_MGS = 0, MIK_CGS = 1, MIK_DGKS = 2 };

typedef struct mik_ctx mik_ctx;       /* device + stream + reduction workspace */
typedef struct mik_csr mik_csr;       /* device CSR operator (the `A` of mul!(y, A, x)) */
typedef struct mik_cg mik_cg;         /* CGIterable            -- src/cg.jl:5-16 */
typedef struct mik_gmres mik_gmres;   /* GMRESIterable + ArnoldiDecomp + Residual -- src/gmres.jl:5-49 */

/* ---- library / context ------------------------------------------------------------------ */
int mik_abi_version(void);
int mik_device_count(int *count);
int mik_ctx_create(int device, mik_ctx **out);
int mik_ctx_destroy(mik_ctx *ctx);
/* The machine behind a context, as hipDeviceGetAttribute reported it in mik_ctx_create, and what the library's selection paths derive
 * from it: nothing assumes "256 compute units in 8 XCDs".  The workgroup -> XCD maps of the banded SpMV kernels and the XCD-local
 * Gram-Schmidt are written for the 8-XCD round-robin dispatch of an unpartitioned MI355X; on any other shape (xcd_maps = 0) the
 * identity map and the device-wide forms run -- same bits (src/orthogonalize.jl:67-79 has one result whatever the form). */
typedef struct mik_device_info {
    int device;                          /* HIP ordinal */
    int compute_units;                   /* hipDeviceAttributeMultiprocessorCount */
    int xcds;                            /* hipDeviceAttributeNumberOfXccs (1 if the runtime does not know the attribute) */
    int wavefront_size;                  /* 64 (mik_ctx_create refuses anything else) */
    int64_t lds_bytes_per_cu, l2_bytes, hbm_bytes;
    char arch[64];                       /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    /* derived (what launch and selection code reads): */
    int planned_compute_units, planned_xcds;   /* = the two above unless a development override is set (include/mik_dev.h MIK_KNOB_MACHINE) */
    int xcd_maps;                        /* 1: XCD-aware workgroup maps in use (planned_xcds == 8) */
    int resident_workgroup_cap;          /* workgroups of a launch whose workgroups wait for each other: one per compute unit */
    int gs_single_launch_max_segments;   /* orthogonalize_and_normalize! as ONE launch up to this many reduction segments (8 per workgroup) */
    int gs_xcd_local_max_workgroups;     /* ... in its XCD-local form up to this many workgroups (0: form not available on this shape) */
    int sweep_grid_cap;                  /* grid cap of the grid-stride vector sweeps */
    int mgs_resident_max_segments;       /* ... and ModifiedGramSchmidt beyond that, with w resident in registers / LDS, up to this many (86 per workgroup: while at
                                          * least 0.6 of w fits on the chip; 0: the device reports less than 160 KB of LDS per compute unit) */
    int reserved[7];
} mik_device_info;
int mik_ctx_info(const mik_ctx *ctx, mik_device_info *out);
/* Adopt an external hipStream_t (e.g. torch's current stream); NULL restores the ctx's own. */
int mik_ctx_set_stream(mik_ctx *ctx, void *hip_stream);
int mik_ctx_synchronize(mik_ctx *ctx);
const char *mik_last_error(mik_ctx *ctx);   /* ctx may be NULL: last error of a failed create */
/* (W, L) of the level-1 reduction tree for `dtype` (see "Reduction semantics"). */
int mik_reduce_shape(int dtype, int *W, int *L);
/* (W, L) of the dot(u, c) fused into the SpMV of the CG step: one row per thread (W = 1), L
 * consecutive 256-row blocks per workgroup. */
int mik_spmv_dot_shape(int *W, int *L);
/* Rows with more than *threshold stored entries are summed with the wave shape: the row's entries are taken in GROUPS of *group
 * (= 4) consecutive entries; lane l of a wave adds, from +0 and in ascending order, the products of the groups l, l + 64, ...
 * (entry e belongs to lane (e / 4) % 64 -- the shape of one 64-thread segment of a dot product read with 16-byte loads), then the
 * wave-64 tree; shorter rows strictly in column order like the reference.  None of the reference's fixtures has such rows. */
int mik_spmv_long_row(int *threshold);
int mik_spmv_long_group(int *group);
/* Rows with more than *segment entries (a multiple of the group) are cut into segments of that many consecutive entries; every
 * segment is summed with the wave shape above and the segment sums are added left to right (one wave per row left a
 * 20,000-entry row to a single wave). */
int mik_spmv_long_segment(int *segment);

/* ---- device memory (similar / zero / copyto! / fill! of the vector interface) ----------- */
int mik_malloc(mik_ctx *ctx, size_t bytes, void **dptr);            /* similar(x)             */
int mik_free(mik_ctx *ctx, void *dptr);
int mik_memcpy_h2d(mik_ctx *ctx, void *dst, const void *src, size_t bytes);   /* synchronous  */
int mik_memcpy_d2h(mik_ctx *ctx, void *dst, const void *src, size_t bytes);   /* synchronous  */
int mik_copy(mik_ctx *ctx, int dtype, int64_t n, const void *x, void *y);     /* copyto!(y, x) src/cg.jl:130 */
int mik_fill(mik_ctx *ctx, int dtype, int64_t n, const void *value, void *x); /* x .= value    src/cg.jl:129 */

/* ---- operator ---------------------------------------------------------------------------- */
/* Upload a SparseMatrixCSC (is_csc = 1: ptr = colptr, idx = rowval -- the layout of
 * test/laplace_matrix.jl:12) or a CSR matrix (is_csc = 0).  Host arrays, Int64 indices with
 * `index_base` (1 for Julia).  The arrays are copied to the device as they are; validation, the conversion to 0-based
 * Int32 CSR (CSC -> CSR transpose, columns ascending within a row = the order Julia's column scatter reaches a row), the
 * operator statistics and the layout analysis run there as kernels (csrc/mik_upload.hip; 0.06 s for the 256^3 Laplacian).
 * Matrices with rows longer than mik_spmv_long_row() or with duplicate (row, column) entries take the host path
 * (single-threaded counting-sort transpose + builders), as does everything when the development knob MIK_KNOB_UPLOAD is set. */
int mik_csr_create(mik_ctx *ctx, int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz,
                   const int64_t *ptr, const int64_t *idx, const void *val, int index_base,
                   int is_csc, mik_csr **out);
/* The same for SparseMatrixCSC{T, Int32} (the reference's tests run Ti in (Int64, Int32): test/gmres.jl:38, test/cg.jl:57): ptr / idx
 * are 32-bit.  Host or device arrays. */
int mik_csr_create_i32(mik_ctx *ctx, int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *ptr, const int32_t *idx,
                       const void *val, int index_base, int is_csc, mik_csr **out);
int mik_csr_destroy(mik_csr *A);
/* Release the CSR arrays (rowptr / col / val: 12 B per entry at fp64) of an operator whose active layout is one of the sliced
 * forms (1, 2, 4, 5) -- mik_spmv never reads them then; they are only what mik_csr_set_layout(A, 0) and the development knobs
 * (csrc/mik_dev.h) fall back to, which have no effect on this operator afterwards.  MIK_ERR_NOTIMPL (nothing released) for an
 * operator that runs on its CSR arrays or has split-off long rows. */
int mik_csr_compact(mik_csr *A);
/* Device layout mik_spmv uses for this operator (chosen at upload from the sparsity pattern; results are
 * bit-identical across layouts): 0 = CSR row-blocks (LDS tile filled by LDS-DMA; product tile for uneven rows; any matrix),
 * 1 = jagged slices (one row per lane, groups of 16 B / sizeof(T) consecutive entries stored lane-interleaved per 64-row
 * slice: long near-uniform rows -- finite-element matrices, variable-coefficient stencils), (2, 3: retired), 4 = sliced-ELL values with per-slice offsets and one presence-mask byte per row (every
 * 256-row slice uses <= 8 distinct offsets: stencils on structured grids), 5 = the same with slice-CONSTANT slot values
 * (within a slice every row that has a slot carries the same value there -- constant-coefficient stencils): the slice
 * stores its <= 8 values once and a row is one mask byte, 6 = the same idea for up to 32 offsets per slice with one 32-bit mask per
 * row (9-point 2-D, 13 / 19 / 27-point 3-D constant-coefficient stencils). */
int mik_csr_layout(const mik_csr *A, int *layout);
/* Choose the layout mik_spmv and the iterables created AFTERWARDS use for this operator: layout = 0 runs it on its plain CSR
 * arrays (k_spmv_rowgather / k_spmv_rowblock -- what any matrix can run on; bench.py measures the north star's CSR figure this
 * way on the same operator), layout = -1 returns to the automatic choice.  MIK_ERR_NOTIMPL for any other value or after
 * mik_csr_compact released the CSR arrays.  Results are bit-identical either way. */
int mik_csr_set_layout(mik_csr *A, int layout);
/* Name of the kernel mik_spmv launches for this operator now (layout, operator properties, development knobs): for
 * profiles and the bench line.  Layout 5 runs k_spmv_sdiab (x through buffer loads: a slot a row does not have reads 0.0 by
 * the descriptor's range check) when every slice value is finite and the offsets fit 32 bits, else k_spmv_sdiac. */
int mik_spmv_kernel(const mik_csr *A, char *name, int len);
/* Bytes of operator data (values, indices / codes, pointers) one mik_spmv launch streams in that layout. */
int mik_csr_stored_bytes(const mik_csr *A, int64_t *bytes);
/* size(A, d), nnz, eltype(A) */
int mik_csr_info(const mik_csr *A, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int *dtype);

/* ---- L1 operator / vector interface ------------------------------------------------------ */
/* mul!(y, A, x) -- SparseArrays mul!, called at src/cg.jl:54,137; src/gmres.jl:245,287 */
int mik_spmv(mik_ctx *ctx, const mik_csr *A, const void *x, void *y);
/* dot(x, y) -- src/cg.jl:55, src/orthogonalize.jl:71.  *out is a host scalar of `dtype`. */
int mik_dot(mik_ctx *ctx, int dtype, int64_t n, const void *x, const void *y, void *out);
/* norm(x) -- src/cg.jl:62,140; src/orthogonalize.jl:75; src/gmres.jl:252 */
int mik_nrm2(mik_ctx *ctx, int dtype, int64_t n, const void *x, void *out);
/* y .+= alpha .* x  -- src/cg.jl:58 (alpha) / :59 (-alpha) */
int mik_axpy(mik_ctx *ctx, int dtype, int64_t n, const void *alpha, const void *x, void *y);
/* y .= x .+ beta .* y -- src/cg.jl:51 (u .= r .+ beta .* u) */
int mik_xpby(mik_ctx *ctx, int dtype, int64_t n, const void *x, const void *beta, void *y);
/* y .-= x -- src/cg.jl:138, src/gmres.jl:246 */
int mik_sub(mik_ctx *ctx, int dtype, int64_t n, const void *x, void *y);
/* x .*= alpha -- src/orthogonalize.jl:76, src/gmres.jl:253 */
int mik_scal(mik_ctx *ctx, int dtype, int64_t n, const void *alpha, void *x);
/* y .= x ./ d -- ldiv!(y, P::JacobiPrec, x) of test/cg.jl:18 (diagonal left preconditioner) */
int mik_divide(mik_ctx *ctx, int dtype, int64_t n, const void *x, const void *d, void *y);

/* ---- L2 Krylov helpers ------------------------------------------------------------------- */
/* orthogonalize_and_normalize!(V[:, 1:k], w, h, method) -> nrm -- src/orthogonalize.jl:13-79.
 * V: device, column-major, leading dimension ldv (elements); w: device n-vector (updated in
 * place, normalised); h: HOST array of k scalars (written); nrm: HOST scalar (written). */
int mik_orthogonalize(mik_ctx *ctx, int dtype, int64_t n, int k, const void *V, int64_t ldv,
                      void *w, void *h, void *nrm, int method);
/* orthogonalize_and_normalize!(V::Vector{Vector}, w, h, ModifiedGramSchmidt()) -> nrm -- src/orthogonalize.jl:53-65: the basis as
 * k separate device n-vectors (V: HOST array of k device pointers).  Same arithmetic and bits as mik_orthogonalize with MIK_MGS
 * on a matrix holding those columns.  (The reference defines this method for ModifiedGramSchmidt only.) */
int mik_orthogonalize_vectors(mik_ctx *ctx, int dtype, int64_t n, int k, const void *const *V, void *w, void *h, void *nrm);
/* mul!(y, V[:, 1:k], c, alpha, 1) -- src/gmres.jl:275 (alpha = 1), src/orthogonalize.jl:16 (-1).
 * c: HOST array of k scalars. */
int mik_gemv_n(mik_ctx *ctx, int dtype, int64_t n, int k, const void *V, int64_t ldv,
               const void *c, const void *alpha, void *y);

/* h = V[:, 1:k]' * w -- mul!(h, adjoint(V), w): src/orthogonalize.jl:15,43 and (column by column) the
 * Gram matrix of src/bicgstabl.jl:121.  One sweep reads w once and every column once.  h: HOST. */
int mik_gemv_t(mik_ctx *ctx, int dtype, int64_t n, int k, const void *V, int64_t ldv, const void *w,
               void *h);
/* Host: solve the small dense system A x = b (column-major n x n) by LU with partial pivoting --
 * lu! + ldiv! at src/bicgstabl.jl:124-125.  A is overwritten by its factors, b by x; MIK_ERR_SINGULAR if A is
 * exactly singular. */
int mik_lu_solve(int dtype, void *A, int64_t lda, int n, void *b);

/* ---- L3 iterables ------------------------------------------------------------------------ */
/* cg_iterator!(x, A, b, Pl; abstol, reltol, maxiter, statevars, initially_zero)
 *   -- src/cg.jl:120-155.  x, b and the CGStateVariables u, r, c (src/cg.jl:114-118) are device
 * n-vectors owned by the caller.  `jacobi_diag` (device n-vector or NULL) selects the
 * PCGIterable with a diagonal left preconditioner (src/cg.jl:18-30,72-100); NULL = Identity(). */
int mik_cg_create(mik_ctx *ctx, const mik_csr *A, void *x, const void *b, void *u, void *r,
                  void *c, const void *jacobi_diag, double abstol, double reltol,
                  int64_t maxiter, int initially_zero, mik_cg **out);
int mik_cg_destroy(mik_cg *it);
/* iterate(it, iteration) -- src/cg.jl:43-66 / :72-100.  *done = 1 and nothing is computed when
 * done(it, iteration) (src/cg.jl:36); otherwise one step runs and *residual = it.residual.
 * r and the residual are those of the step just returned, and so is x for every reader ordered after the
 * ctx stream (mik_memcpy_d2h, kernels on that stream, or after mik_synchronize): with a CSR operator the
 * update x .+= alpha .* u of a step is carried out by the sweep over u that opens the next step (same
 * operands, same rounding), which is on the stream before this call returns.  The search direction u and c = A u are
 * the iterable's scratch: with a CSR operator and no host callbacks the library enqueues the first half of
 * the NEXT step (u = r + beta u, c = A u, alpha) before it waits for this step's residual, so on return
 * u and c may already belong to step iteration + 1 (results of any call sequence are unchanged; the
 * device-side stopping flag turns that half into a no-op once the iteration has stopped). */
int mik_cg_iterate(mik_cg *it, int64_t iteration, double *residual, int *done);
/* 1 if this iterable applies x .+= alpha .* u in the sweep over u that opens the next step (CSR operator, no
 * preconditioner callback; see mik_cg_iterate), 0 if in the step's own update sweep.  Results are bit-identical either way. */
int mik_cg_fused_x(const mik_cg *it, int *fused);
/* Up to max_steps consecutive iterate() calls with ONE host synchronisation: the stopping test
 * of src/cg.jl:36 is evaluated on the device after every step and later steps become no-ops.
 * residuals[0..*steps_done-1] receive it.residual after each executed step. */
int mik_cg_iterate_many(mik_cg *it, int64_t iteration, int64_t max_steps, double *residuals,
                        int64_t *steps_done);
/* fields of CGIterable read by cg! (src/cg.jl:227,232,238): mv_products, residual, tol,
 * converged(it) */
int mik_cg_state(const mik_cg *it, double *residual, double *prev_residual, double *tol,
                 int64_t *maxiter, int64_t *mv_products, int *converged);

/* gmres_iterable!(x, A, b; Pl, Pr, abstol, reltol, restart, maxiter, initially_zero, orth_meth)
 * -- src/gmres.jl:108-136.  x, b: device n-vectors owned by the caller.  pl_diag / pr_diag: NULL =
 * Identity(), or a device n-vector d making the preconditioner ldiv!(y, P, x) = y .= x ./ d (the three
 * expand! methods src/gmres.jl:285-304, init! :249 and update_solution! :278-283 are followed).  The Krylov basis V (n x (restart+1), src/gmres.jl:13) lives on the device and
 * the Hessenberg matrix, Givens least squares and null-vector residual recurrence
 * (src/gmres.jl:224-233,262-271; src/hessenberg.jl:15-46) on the host, inside the handle. */
int mik_gmres_create(mik_ctx *ctx, const mik_csr *A, void *x, const void *b, const void *pl_diag,
                     const void *pr_diag, double abstol, double reltol, int restart, int64_t maxiter,
                     int initially_zero, int orth_method, mik_gmres **out);
int mik_gmres_destroy(mik_gmres *it);
/* iterate(g, iteration) -- src/gmres.jl:57-106 */
int mik_gmres_iterate(mik_gmres *it, int64_t iteration, double *residual, int *done);
/* Up to max_steps consecutive iterate() calls in one entry (the loop of gmres!, src/gmres.jl:207-214): residuals[j] =
 * residual.current after step j; *steps_done < max_steps iff done(g, iteration + *steps_done). */
int mik_gmres_iterate_many(mik_gmres *it, int64_t iteration, int64_t max_steps, double *residuals, int64_t *steps_done);
/* fields read by gmres! (src/gmres.jl:210,218): mv_products, residual.current, tol, k, beta */
int mik_gmres_state(const mik_gmres *it, double *residual, double *tol, double *beta, int *k,
                    int64_t *mv_products, int *converged);

/* ---- any operator, any preconditioner on the fused iterables -------------------------------------------- */
/* The reference asks of A only mul!(y, A, v) and of Pl / Pr only ldiv!(y, P, x) (docs/src/getting_started.md:25-30,
 * docs/src/preconditioning.md:5-14; exercised with a LinearMap at test/gmres.jl:59-66 and with lu(A) as Pl at
 * test/gmres.jl:28-35, test/cg.jl:71-77).  The callbacks below carry that contract across the C ABI: x and y are DEVICE
 * n-vectors of the handle's dtype; the callback must leave its work ordered on the context's stream (enqueue kernels on
 * it -- mik_ctx_set_stream adopts a host framework's stream -- or finish before returning) and return 0.  A non-zero
 * return surfaces as MIK_ERR_CALLBACK.  With a callback operator the CG step keeps its fused vector sweeps and reduces
 * dot(u, c) with the (W, L) shape of mik_reduce_shape instead of inside the SpMV epilogue. */
typedef int (*mik_mul_fn)(void *user, const void *x, void *y);      /* y = A * x      -- mul!(y, A, x)   */
typedef int (*mik_ldiv_fn)(void *user, void *y, const void *x);     /* y = P \ x      -- ldiv!(y, P, x); y may alias x */
typedef struct mik_operator {
    int dtype;                  /* MIK_F64 / MIK_F32 (ignored when csr is given) */
    int64_t n;                  /* size(A, 1) = size(A, 2) (ignored when csr is given) */
    const mik_csr *csr;         /* a device CSR operator: the fully fused path ...                            */
    mik_mul_fn mul;             /* ... or any operator as a callback (csr == NULL)                           */
    void *user;
} mik_operator;
typedef struct mik_precond {    /* all NULL = Identity() (src/common.jl:28-32) */
    const void *diag;           /* device n-vector d: ldiv!(y, P, x) = y .= x ./ d, fused into the sweeps       */
    mik_ldiv_fn ldiv;           /* or any preconditioner as a callback                                        */
    void *user;
} mik_precond;
/* cg_iterator!(x, A, b, Pl; ...) -- src/cg.jl:120-155 with any A / Pl; mik_cg_iterate* and mik_cg_state work as usual. */
int mik_cg_create_op(mik_ctx *ctx, const mik_operator *A, const mik_precond *Pl, void *x, const void *b, void *u, void *r,
                     void *c, double abstol, double reltol, int64_t maxiter, int initially_zero, mik_cg **out);
/* gmres_iterable!(x, A, b; Pl, Pr, ...) -- src/gmres.jl:108-136 with any A / Pl / Pr (all three expand! methods,
 * src/gmres.jl:285-304). */
int mik_gmres_create_op(mik_ctx *ctx, const mik_operator *A, const mik_precond *Pl, const mik_precond *Pr, void *x, const void *b,
                        double abstol, double reltol, int restart, int64_t maxiter, int initially_zero, int orth_method,
                        mik_gmres **out);

/* ---- fused sweeps for the other solvers of the package ------------------------------------------- */
/* Each call is several consecutive reference statements executed as ONE pass over the vectors, with the
 * same per-element operations in the same order (bit-identical to issuing the L1 calls one by one).  `hints` is a
 * bit mask of operands the caller will not touch again soon: they are streamed past the caches (non-temporal
 * loads / stores) so that the operands that ARE re-read stay resident; results never depend on it (0 = none). */
/* y .+= alpha .* x (x NULL: no update); then *out = dot(z, y), or norm(y) when z is NULL
 *   -- src/minres.jl:104+107 and :109+112 */
int mik_axpy_dot(mik_ctx *ctx, int dtype, int64_t n, const void *alpha, const void *x, void *y, const void *z, void *out,
                 int hints /* 1: x */);
/* x .+= alpha .* u; r .-= alpha .* c; *out = norm(r)   -- src/chebyshev.jl:51-54 (same shape as src/cg.jl:58-62) */
int mik_axpy2_nrm2(mik_ctx *ctx, int dtype, int64_t n, const void *alpha, const void *u, void *x, const void *c, void *r, void *out,
                   int hints /* 1: x, 2: c, 4: u */);
/* c = Pl \ r (pl_diag NULL = Identity); u .= c when `first`, else u .= c .+ beta .* c   -- src/chebyshev.jl:35-45 */
int mik_cheb_direction(mik_ctx *ctx, int dtype, int64_t n, const void *r, const void *pl_diag, const void *beta, int first, void *u);
/* LSQR / LSMR (src/lsqr.jl, src/lsmr.jl), groups of consecutive statements as single sweeps with the same per-element operations:
 *   mik_xpby_nrm2     y .= x .+ beta .* y; *out = norm(y)           u .= -alpha .* u .+ tmpm; beta = norm(u)  (src/lsqr.jl:151-152, :159-160;
 *                                                                   src/lsmr.jl:161-162, :167-168)
 *   mik_lsqr_update   x .+= t1*w; w = t2 .* w .+ v; wrho .= w .* inv_rho (never stored); *out = norm(wrho)              (src/lsqr.jl:189-192)
 *   mik_lsmr_update   hbar .= hbar .* c1 .+ h; x .+= c2 * hbar; h .= h .* c3 .+ v; *out = norm(x)                        (src/lsmr.jl:199-201, :242)
 * Scalars: HOST values of `dtype`. */
int mik_xpby_nrm2(mik_ctx *ctx, int dtype, int64_t n, const void *x, const void *beta, void *y, void *out);
int mik_lsqr_update(mik_ctx *ctx, int dtype, int64_t n, const void *t1, const void *t2, const void *inv_rho, void *x, void *w, const void *v, void *out);
int mik_lsmr_update(mik_ctx *ctx, int dtype, int64_t n, const void *c1, const void *c2, const void *c3, void *hbar, void *h, void *x, const void *v, void *out);
/* QMR (src/qmr.jl), groups of consecutive statements as single sweeps with the same per-element operations:
 *   mik_axpy2_dot    y .+= a .* x1; y .+= b .* x2 (x2 may be NULL); *out = dot(y, z) (z may be NULL: no reduction, out untouched)
 *                    -- the two axpy! of a Lanczos vector (:70-72 / :76-78) and, for the second vector, vw = dot(v_next, w_next) (:81)
 *   mik_scal2        x .*= a; y .*= b                                                                             (:90-91)
 *   mik_qmr_update   p = v - h1 p_curr - h0 p_prev (either may be NULL), p .*= inv, x .+= g .* p; p is stored at p_out, the next p_curr
 *                    (p_out may be p_prev's storage: the caller rotates names instead of the two copies of :196-197)  (:188-197)
 * Scalars: HOST values of `dtype` (neg_h1 = -H[2], neg_h0 = -H[1]). */
int mik_axpy2_dot(mik_ctx *ctx, int dtype, int64_t n, const void *a, const void *x1, const void *b, const void *x2, void *y, const void *z, void *out);
int mik_scal2(mik_ctx *ctx, int dtype, int64_t n, const void *a, void *x, const void *b, void *y);
int mik_qmr_update(mik_ctx *ctx, int dtype, int64_t n, const void *v, const void *neg_h1, const void *p_curr, const void *neg_h0, const void *p_prev,
                   const void *inv, const void *g, void *x, void *p_out);
/* v_next .*= inv_h3; w_next .= v_curr .+ neg_h1 .* w_curr .+ neg_h0 .* w_prev (each term skipped when its vector
 * is NULL); w_next .*= inv_h2; x .+= rhs0 .* w_next   -- src/minres.jl:113, :136-142 */
int mik_minres_update(mik_ctx *ctx, int dtype, int64_t n, const void *inv_h3, void *v_next, const void *v_curr, const void *neg_h1,
                      const void *w_curr, const void *neg_h0, const void *w_prev, const void *inv_h2, void *w_next,
                      const void *rhs0, void *x, int hints /* 1: x, 2: w_prev */);

/* M = V' * V for k <= 5 columns in ONE pass over V (all pairwise dots; M is k x k, column-major, host) -- the
 * Gram matrix of src/bicgstabl.jl:120; entry (r, c) equals mik_dot(V[:, r], V[:, c]) bit for bit. */
int mik_gram(mik_ctx *ctx, int dtype, int64_t n, int k, const void *V, int64_t ldv, void *M);
/* BiCGStab(l) minimal-residual update, src/bicgstabl.jl:127-132, as one sweep: us[:, 1] -= us[:, 2:l+1] * gamma;
 * x += rs[:, 1:l] * gamma; rs[:, 1] -= rs[:, 2:l+1] * gamma; *out = norm(rs[:, 1]).  l <= 8; gamma: l host scalars. */
int mik_bicgstab_mr_update(mik_ctx *ctx, int dtype, int64_t n, int l, void *us, int64_t ldu, void *rs, int64_t ldr, void *x,
                           const void *gamma, void *out);

/* One whole outer iteration of BiCGStab(l) -- iterate(::BiCGStabIterable), src/bicgstabl.jl:79-134 -- per call, with rho, beta,
 * sigma, alpha, the Gram matrix, gamma and omega kept on the device: the host waits once, for the residual norm it returns
 * (the statement-by-statement form through mik_dot / mik_xpby / mik_spmv / mik_axpy / mik_gram / mik_lu_solve /
 * mik_bicgstab_mr_update waits 2 l + 3 times; the same bits wherever mik_bicgstab_dot_shape reports the vector shape).  The caller owns x, the residual block rs and the
 * search block us (n x (l + 1), column-major, leading dimensions ldr / ldu) as set up by bicgstabl_iterator!
 * (src/bicgstabl.jl:25-73: rs[:, 1] = Pl \ (b - A x), us = 0) and the shadow residual r_shadow (:38); pl_diag: the diagonal
 * of a Jacobi Pl (ldiv! = elementwise division, :98, :108) or NULL for Identity.  omega = sigma = 1 at creation (:59).
 * l = 1 ... 4.  Between two steps of a handle the blocks are the handle's state (as the fields of the reference's iterable are): the MR sweep of a
 * step leaves the segment sums of dot(r_shadow, rs[:, 1]) -- rho of the next step's first column (:89) -- with the handle.
 * MIK_ERR_SINGULAR when lu! meets an exactly singular pivot (the reference throws SingularException); the handle is
 * then latched: every later mik_bicgstab_step reports MIK_ERR_SINGULAR again (its device scalars are not steppable).  Handles a host
 * never destroys are freed by mik_ctx_destroy of their context (a finalizer that finds the context closed must skip the destroy call). */
typedef struct mik_bicgstab mik_bicgstab;
int mik_bicgstab_create(mik_ctx *ctx, const mik_csr *A, int l, void *x, void *rs, int64_t ldr, void *us, int64_t ldu,
                        const void *r_shadow, const void *pl_diag, mik_bicgstab **out);
int mik_bicgstab_step(mik_bicgstab *it, void *residual);          /* residual: one scalar of A's element type */
/* (W, L) of the reduction tree of sigma = dot(r_shadow, A u) (src/bicgstabl.jl:100) and of rho = dot(r_shadow, rs[:, j]) for j >= 2 (:89)
 * inside mik_bicgstab_step: where the operator's SpMV kernel takes a dot epilogue and Pl = Identity, both are formed in the SpMV launch
 * that produces the vector, one partial per 256-row block (mik_spmv_dot_shape); else the vector shape (mik_reduce_shape), which rho of
 * the first column and the Gram matrix always have.  The oracle takes it. */
int mik_bicgstab_dot_shape(const mik_bicgstab *it, int *W, int *L);
int mik_bicgstab_destroy(mik_bicgstab *it);

/* One whole iteration of MINRES -- iterate(::MINRESIterable), src/minres.jl:95-159 -- per call: mul!, the Lanczos step with its
 * projection (:104-107), the orthogonalisation with its norm (:109-112), both Givens rotations and the right-hand side (:116-133),
 * and the tail sweep (:113, :136-142), the scalars on the device; the host waits once, for the residual norm |rhs[2]| (:154) it
 * returns (the statement-by-statement form through mik_spmv / mik_axpy_dot / mik_givens / mik_minres_update waits twice inside
 * the iteration; same bits).  The caller owns x and the six work vectors as minres_iterable! sets them up (:38-87:
 * v_curr = (b - A x) / resnorm0, w's = 0); the handle rotates its three v and three w pointers after every step exactly as
 * :145-146 do, and the caller rotates its own names with it.  iteration counts from 1 (:91). */
typedef struct mik_minres mik_minres;
int mik_minres_create(mik_ctx *ctx, const mik_csr *A, void *x, void *v_prev, void *v_curr, void *v_next, void *w_prev, void *w_curr,
                      void *w_next, double resnorm0, int skew_hermitian, mik_minres **out);
int mik_minres_step(mik_minres *it, int64_t iteration, void *resnorm);   /* resnorm: one scalar of A's element type */
/* (W, L) of the reduction tree of proj = dot(v_curr, v_next) (src/minres.jl:107) inside mik_minres_step: where the operator's SpMV
 * kernel takes the Lanczos step as its epilogue (y = A x - H[2] v_prev stored once, the dot formed in the same launch) it is the
 * shape of the dot fused into the CG SpMV (mik_spmv_dot_shape), else the vector shape (mik_reduce_shape).  The oracle takes it. */
int mik_minres_proj_shape(const mik_minres *it, int *W, int *L);
int mik_minres_destroy(mik_minres *it);

/* One step of IDR(s) -- iterate(::IDRSIterable, (iter, step)), src/idrs.jl:164-272 -- per call.  step = 1..s: the small triangular solve (:187,
 * host), V / Q / ldiv!(Pl, V) / U[k] in ONE sweep (:188-202), mul! (:203), the bi-orthogonalisation against P[1..k-1] as a chain of sweeps whose
 * coefficients never leave the device (:207-211), M[k..s, k] as one batched dot (:215-217), and beta / R / X / norm(R) in one sweep (:221-225);
 * step = s + 1: the polynomial step with omega(Q, R) (:242-256).  Residual smoothing (:226-235) when X_s / R_s are given.  The host waits once per
 * step (twice in steps 1 and s + 1).  The caller owns the vectors as idrs_iterable! sets them up (:116-147): R = C - A X, U = G = 0 (n x s,
 * column-major), X_s = copy(X) and R_s = copy(R) for smoothing (both NULL otherwise), and P: the s shadow vectors (the reference fills them
 * with rand!, :136).  pl_diag: diagonal of a Jacobi Pl or NULL (Identity).  M (= I), f, c and omega (= 1) live in the handle.  s <= 32.
 * normR0 = norm(R).  mik_idrs_step returns it.normR of the reference after the step (norm(R), or norm(R_s) with smoothing); the caller copies
 * X_s into X on termination as :171-173 do. */
typedef struct mik_idrs mik_idrs;
int mik_idrs_create(mik_ctx *ctx, const mik_csr *A, int s, void *x, void *r, const void *P, int64_t ldp, void *U, int64_t ldu, void *G,
                    int64_t ldg, const void *pl_diag, void *x_s, void *r_s, double normR0, mik_idrs **out);
int mik_idrs_step(mik_idrs *it, int step, void *normR);            /* normR: one scalar of A's element type */
int mik_idrs_state(const mik_idrs *it, void *omega, void *M, void *f);   /* host copies (any may be NULL): omega, M (s x s column-major), f (s) */
int mik_idrs_destroy(mik_idrs *it);

/* ---- row-partitioned GMRESIterable: one process per GPU ---------------------------------------- */
/* The same iterable (src/gmres.jl:57-106) over a contiguous row block.  The Arnoldi basis, x, b and
 * the diagonal preconditioners are this rank's n_loc rows; A_loc is the block as an n_loc x n_ext
 * operator whose columns [n_loc, n_ext) are halo entries (same layout as mik_cgd_create).  The two
 * places where ranks couple are handed to the caller:
 *   halo(user):   called with send_buf[i] = v[send_idx[i]] packed (enqueued on the ctx stream) for the
 *                 vector v about to be multiplied; must leave x_ext[n_loc .. n_ext) filled with the
 *                 neighbours' entries, ordered after that packing on the ctx stream (RCCL send/recv on
 *                 the same stream, or a host-staged copy after a stream synchronisation);
 *   reduce(user, dtype, count, values):  values[0..count) are this rank's partial sums (host scalars of
 *                 `dtype`); on return they must hold  ((p_0 + p_1) + p_2) + ...  over ranks 0..P-1 in rank
 *                 order, identically on every rank (dot / norm^2 of src/orthogonalize.jl:71,75,
 *                 src/gmres.jl:252; CGS / DGKS pass all k projections in one call).
 * Either callback returns 0 on success.  Every rank must make the same sequence of calls.  Hessenberg
 * matrix, Givens solve and stopping test are replicated on every rank from identical scalars. */
typedef int (*mik_halo_fn)(void *user);
typedef int (*mik_reduce_fn)(void *user, int dtype, int count, void *values);
typedef struct mik_partition {
    int rank, nranks;
    int64_t n_ext;              /* n_loc + number of halo columns of A_loc */
    void *x_ext;                /* device n_ext-vector: SpMV input; the library writes [0, n_loc) */
    const int32_t *send_idx;    /* device: local indices packed for the neighbours */
    int64_t n_send;
    void *send_buf;             /* device: n_send packed entries */
    mik_halo_fn halo;
    mik_reduce_fn reduce;
    void *user;
    struct mik_plink *link;     /* NULL: the two callbacks above couple the ranks.  A connected link ("Transport 3" below; created on a communicator
                                 * of THIS ctx): the library exchanges by itself, on the device -- halo pushed into the neighbours' landing buffers,
                                 * every projection / norm summed over the ranks inside the kernel that finalises it (rank order, same bits) -- and
                                 * never calls halo / reduce: no host round trip between the k + 1 reductions of an Arnoldi column
                                 * (src/orthogonalize.jl:69-76), one wait per inner iteration as on a single GPU. */
} mik_partition;
int mik_gmres_create_partitioned(mik_ctx *ctx, const mik_csr *A_loc, void *x, const void *b, const void *pl_diag,
                                 const void *pr_diag, double abstol, double reltol, int restart, int64_t maxiter,
                                 int initially_zero, int orth_method, const mik_partition *part, mik_gmres **out);

/* ---- row-partitioned CGIterable: one process per GPU (new design; the reference is serial) --- */
/* out[i] = x[idx[i]], i < m (idx: device Int32) -- packs halo entries for the neighbour ranks. */
int mik_gather(mik_ctx *ctx, int dtype, int64_t m, const int32_t *idx, const void *x, void *out);
/* Rank `rank` of `nranks` owns a contiguous block of n_loc rows.  A_loc is that block as an
 * n_loc x (n_loc + n_ghost) operator: columns [0, n_loc) are the owned entries of a vector, columns
 * [n_loc, n_loc + n_ghost) the halo entries the host receives from the neighbours into the tail of
 * u_ext before phase 1 / 11.  x, b, r, c: device n_loc-vectors; u_ext: device (n_loc + n_ghost)-
 * vector; send_idx / send_buf: local indices to pack and the packed buffer (n_send entries);
 * dot_all / rr_all: device arrays of nranks scalars -- every rank writes slot [rank], the host
 * all-gathers them between phases (RCCL over xGMI via torch.distributed, gloo in CPU tests).
 * Sums over ranks run in rank order on every rank, so all ranks hold identical scalars. */
typedef struct mik_cgd mik_cgd;
int mik_cgd_create(mik_ctx *ctx, const mik_csr *A_loc, void *x, const void *b, void *u_ext, void *r,
                   void *c, const int32_t *send_idx, int64_t n_send, void *send_buf, void *dot_all,
                   void *rr_all, int rank, int nranks, double abstol, double reltol, int64_t maxiter,
                   int initially_zero, mik_cgd **out);
int mik_cgd_destroy(mik_cgd *it);
/* Enqueue one phase (no host synchronisation).  cg_iterator! (src/cg.jl:120-155): 10 = pack x's
 * halo, [exchange], 11 = r = b - A x and local |r|^2, [all-gather rr], 12 = residual, tolerance.
 * iterate (src/cg.jl:43-66): 0 = u = r + beta u and pack, [exchange], 1 = c = A u with local
 * dot(u, c), [all-gather dot], 2 = alpha, x += alpha u, r -= alpha c, local |r|^2, [all-gather rr],
 * 3 = residual, beta, stopping test of src/cg.jl:36 for iteration + 1 (later steps become no-ops). */
int mik_cgd_phase(mik_cgd *it, int phase, int64_t iteration);
/* Overlap of the halo exchange with the SpMV: row-blocks (256 rows each) [rb_begin, rb_end) of this rank contain no
 * row that references a halo column.  Step B may then be issued as phase 4 (those row-blocks; needs no halo, so
 * it runs while the exchange is in flight) followed by phase 5 (the remaining row-blocks + the local dot) -- same
 * results as phase 1.  MIK_ERR_NOTIMPL if the operator's layout cannot be launched over a range. */
int mik_cgd_set_interior(mik_cgd *it, int64_t rb_begin, int64_t rb_end);
/* Wait for everything enqueued; residual / tol / done of the last step and the residuals of the
 * steps executed since the previous wait (at most 1024 steps may be enqueued between waits). */
int mik_cgd_wait(mik_cgd *it, double *residual, double *tol, int *done, double *history, int64_t cap,
                 int64_t *steps);

/* ---- the exchanges of the row-partitioned CGIterable inside the library ----------------------------------- */
/* With the calls below the host no longer drives phases and collectives itself: after mik_cgd_create it registers the
 * halo plan and a transport once, and every batch of iterate() calls is ONE entry (a Julia host: one ccall).
 * Halo plan: rank `rank` receives recv_cnt[i] entries from rank recv_peer[i] into the ghost tail of u_ext at element
 * offset recv_off[i] (relative to n_loc), and sends send_cnt[i] entries of send_buf starting at send_off[i] to rank
 * send_peer[i] (the packing order of send_idx).  Arrays are copied. */
int mik_cgd_set_halo_plan(mik_cgd *it, int n_recv, const int *recv_peer, const int64_t *recv_off, const int64_t *recv_cnt,
                          int n_send, const int *send_peer, const int64_t *send_off, const int64_t *send_cnt);
/* What the plan made of the step: *runs = number of contiguous row runs (0, 1 or 2) that are updated, packed and put on the
 * wire before the bulk of the u = r + beta u sweep (0: the halo follows the whole sweep), *rows = their total length,
 * *merged = 1 if update and pack of those rows are one kernel.  Any pointer may be NULL. */
int mik_cgd_halo_early(const mik_cgd *it, int *runs, int64_t *rows, int *merged);

/* Transport 1 -- RCCL over xGMI, one process per GPU.  librccl is bound at run time (dlopen; a process that already
 * carries RCCL, e.g. PyTorch-ROCm, shares that copy); MIK_ERR_NOTIMPL if it cannot be loaded.  Rank 0 obtains the
 * 128-byte ncclUniqueId with mik_comm_unique_id and hands it to the other ranks by whatever channel the host has (MPI.jl
 * bcast, a file, torch.distributed); every rank then calls mik_comm_create (collective).  id128 = NULL gives a communicator
 * without RCCL: a world of one needs nothing else, more ranks connect mailboxes and landing buffers (transport 3 below). */
typedef struct mik_comm mik_comm;
int mik_comm_unique_id(void *id128);
int mik_comm_create(mik_ctx *ctx, const void *id128, int rank, int nranks, mik_comm **out);
int mik_comm_destroy(mik_comm *comm);
int mik_comm_info(const mik_comm *comm, int *rank, int *nranks, int *uses_rccl);
/* values[0..count) (host scalars of dtype, count <= 256): this rank's partial sums in, ((p_0 + p_1) + p_2) + ... over the
 * ranks out, identical on every rank -- exactly the mik_reduce_fn contract, so a row-partitioned GMRES host passes a
 * two-line callback around it.  Blocks (ncclAllGather on the ctx stream + one read-back; a communicator without RCCL sends the
 * values through the vector slots of its connected mailboxes -- same bits, bounded wait). */
int mik_comm_allgather_sum(mik_comm *comm, int dtype, int count, void *values);
/* The mik_halo_fn of a row-partitioned GMRES over RCCL: ncclSend / ncclRecv of the packed buffer into the ghost region
 * on the ctx stream; segments as in mik_cgd_set_halo_plan. */
int mik_comm_halo(mik_comm *comm, int dtype, const void *send_buf, void *ghost, int n_recv, const int *recv_peer,
                  const int64_t *recv_off, const int64_t *recv_cnt, int n_send, const int *send_peer, const int64_t *send_off,
                  const int64_t *send_cnt);
int mik_cgd_set_comm(mik_cgd *it, mik_comm *comm);
/* cg_iterator! (src/cg.jl:120-155) over the partition: init phases + exchanges + one wait. */
int mik_cgd_init(mik_cgd *it, double *residual, double *tol);
/* Up to max_steps iterate() calls (src/cg.jl:43-66) with ONE host wait; collective: every rank makes the same call.
 * Per step: pack, halo (ncclSend / ncclRecv on a side stream -- the interior row-blocks of the SpMV run meanwhile when
 * mik_cgd_set_interior was called), c = A u with the local dot, ncclAllGather of one scalar per rank, update,
 * ncclAllGather, stopping test; the P partial sums are added in rank order on every device.  At most 1024 steps. */
int mik_cgd_iterate_many(mik_cgd *it, int64_t iteration, int64_t max_steps, double *residuals, int64_t *steps_done);

/* Transport 2 -- in-process group: ONE host thread drives all P ranks (its[p] = rank p, each created on its own ctx;
 * the contexts may sit on different GPUs -- halos and scalars then move by peer copies over xGMI -- or share one, which
 * is how the step routine is verified on a single-GPU box).  Same steps, same bits as transport 1. */
int mik_cgd_group_init(mik_cgd **its, int P, double *residual, double *tol);
int mik_cgd_group_iterate_many(mik_cgd **its, int P, int64_t iteration, int64_t max_steps, double *residuals, int64_t *steps_done);
int mik_cgd_group_release(mik_cgd **its, int P);    /* frees the group's events / side streams (also done by mik_cgd_destroy) */

/* Transport 3 -- peer-mapped mailbox over xGMI (SURVEY.md section 5 "backend B", section 8e): no collective launch in the step.
 * Every communicator owns a mailbox in fine-grained device memory.  mik_comm_mailbox_export gives its 64-byte HIP IPC handle; the
 * host gathers the handles of all ranks (rank order) over whatever channel it has and every rank calls mik_comm_mailbox_connect
 * (ranks may even share one GPU: HIP IPC has no one-rank-per-device rule).  From then on the two scalars of a step travel as
 * {value, sequence number} stores into every peer's mailbox, issued by the kernel that finalised the reduction; every rank adds the
 * P values in rank order -- the bits of transports 1 and 2.
 * The halo travels through LANDING BUFFERS: fine-grained device memory owned by the library (one per link, two halves used
 * alternately).  A push kernel stores the packed send buffer straight into the neighbours' landing buffers over xGMI and posts the
 * exchange number in their mailboxes; on the receiving side one kernel waits for the flags and copies the landed entries into the
 * ghost tail of the extended vector with system-scope loads.  The SpMV that follows therefore reads halo data its OWN device wrote --
 * no memory of the host (a tensor of a pooling allocator) is ever mapped into a peer, and nothing depends on which peer-written
 * lines a cache of the receiving device still holds.
 * A communicator created with id128 = NULL and nranks > 1 has no RCCL at all and needs both mailboxes and links; one created with a
 * ncclUniqueId may connect mailboxes only (scalars by mailbox, halo by ncclSend / ncclRecv).
 * Waits are bounded (MIK_MAILBOX_TIMEOUT_MS, default 10000): MIK_ERR_HIP instead of a hung queue. */
int mik_comm_mailbox_export(mik_comm *comm, void *handle64);
int mik_comm_mailbox_connect(mik_comm *comm, const void *handles /* nranks x 64 bytes, rank order; this rank's entry is ignored */);
int mik_comm_mailbox_info(const mik_comm *comm, int *connected, int *finegrained);

/* The links of ONE row partition on a communicator: its halo plan (segments as in mik_cgd_set_halo_plan: offsets into the ghost region
 * of n_ghost entries / into the packed send buffer, at most 8 per direction), its landing buffer and the mappings of the neighbours'.
 *   mik_plink_export   64-byte HIP IPC handle of this rank's landing buffer;
 *   mik_plink_connect  collective, after the host gathered handle and n_ghost of every rank: handles / ghost_counts per RANK (this
 *                      rank's entries are ignored; a rank may be its own neighbour), dst_elem per SEND segment = the element of the
 *                      receiver's ghost region at which the segment lands (the offset of its matching receive segment).
 * A row-partitioned GMRES iterable takes a connected link in mik_partition.link; the row-partitioned CG iterable builds its own from the
 * plan it was given (mik_cgd_ghost_export = create + export, mik_cgd_connect_ghosts = connect). */
typedef struct mik_plink mik_plink;
int mik_plink_create(mik_comm *comm, int dtype, int64_t n_ghost, int n_recv, const int *recv_peer, const int64_t *recv_off, const int64_t *recv_cnt,
                     int n_send, const int *send_peer, const int64_t *send_off, const int64_t *send_cnt, mik_plink **out);
int mik_plink_export(mik_plink *link, void *handle64);
int mik_plink_connect(mik_plink *link, const void *handles, const int64_t *ghost_counts, const int64_t *dst_elem);
int mik_plink_info(const mik_plink *link, int *connected, int *finegrained, int64_t *n_ghost);
/* One halo exchange through a connected link, blocking (collective: every rank of the plan calls it): send_buf (device, packed as the
 * plan's send segments say) is stored into the neighbours' landing buffers, the entries that landed here are copied into ghost (device,
 * n_ghost entries).  What a host that drives a row-partitioned operator itself passes as its mik_halo_fn, and what the transport
 * self-test of bench.py --gpus N times.  A peer that never arrives: MIK_ERR_HIP after MIK_MAILBOX_TIMEOUT_MS.
 * Invariants of the landing buffers (two halves used alternately, no consumed-acknowledgement; one push ticket per communicator):
 *   - ONE exchange in flight per communicator: links of one communicator are used from its context's stream, one after the other;
 *   - a rank must not run two exchanges ahead of a neighbour.  A plan whose send peers equal its receive peers guarantees that by itself;
 *     any other plan only when a sum over all ranks separates two exchanges -- true for cg! / gmres! (a dot or norm follows every
 *     product), so the iterables accept such plans; mik_plink_exchange refuses them (MIK_ERR_NOTIMPL). */
int mik_plink_exchange(mik_plink *link, const void *send_buf, void *ghost);
int mik_plink_destroy(mik_plink *link);
/* After mik_cgd_set_halo_plan and mik_cgd_set_comm (transport "mailbox"): */
int mik_cgd_ghost_export(mik_cgd *it, void *handle64);
int mik_cgd_connect_ghosts(mik_cgd *it, const void *handles, const int64_t *ghost_counts, const int64_t *dst_elem);

/* ---- Hessenberg least squares (host) ------------------------------------------------------ */
/* ldiv!(FastHessenberg(H), rhs) -- src/hessenberg.jl:15-46.  Host arrays of `dtype`; H is
 * (width+1) x width column-major with leading dimension ldh, overwritten by R; rhs has width+1
 * entries, overwritten by [y; residual]. */
int mik_hessenberg_ldiv(int dtype, void *H, int64_t ldh, int width, void *rhs);

/* Host: LinearAlgebra.givensAlgorithm(f, g) -> out = {c, s, r} with [c s; -s c] [f; g] = [r; 0]
 * (src/hessenberg.jl:24, src/minres.jl:129).  f, g, out: host scalars / 3-array of dtype. */
int mik_givens(int dtype, const void *f, const void *g, void *out);

/* ---- measurement -------------------------------------------------------------------------- */
/* Time `reps` back-to-back launches of the SpMV (optionally with the fused dot epilogue used by
 * the CG step) with HIP events on the ctx stream; returns average milliseconds per launch. */
int mik_time_spmv(mik_ctx *ctx, const mik_csr *A, const void *x, void *y, int fused_dot, int reps,
                  double *avg_ms);
/* In-loop timing of the SpMV launch inside mik_cg_iterate / mik_cg_iterate_many: a HIP event pair
 * on the ctx stream brackets every SpMV launch of the CG step.  First reports the totals gathered
 * so far (either pointer may be NULL), then: enable = 1 (or 2, see below) resets the totals and switches timing on,
 * 0 switches it off, -1 leaves the mode unchanged. */
int mik_cg_profile(mik_cg *it, int enable, double *spmv_ms_total, int64_t *spmv_launches);
/* enable = 2 in mik_cg_profile brackets all three streaming launches of the step; totals per kernel:
 * [0] the SpMV (src/cg.jl:54), [1] u .= r .+ beta .* u (:51), [2] x / r update + |r|^2 (:58-62).  ms_total / launches: 3 entries each. */
int mik_cg_profile_kernels(const mik_cg *it, double *ms_total, int64_t *launches);
/* The same bracket around every SpMV launch of the row-partitioned iterable's steps (mik_cgd_iterate_many, mik_cgd_phase): the in-loop
 * SpMV time of a rank of BASELINE.json configs[3] (bench.py --gpus N: `roofline`).  Same protocol as mik_cg_profile (enable 1 / 0 / -1). */
int mik_cgd_profile(mik_cgd *it, int enable, double *spmv_ms_total, int64_t *spmv_launches);

#ifdef __cplusplus
}
#endif
#endif /* MIK_H */
