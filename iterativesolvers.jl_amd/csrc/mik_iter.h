// mik_iter.h -- host structures of the L3 iterables (CGIterable and its row-partitioned form), shared by
// mik_krylov.hip (kernels, single-GPU protocol) and mik_comm.hip (RCCL / in-process transports).
#pragma once
#include "mik_internal.h"

template <typename T> struct CgDev {
    T res, prev_res, alpha, beta, dot_uc, rr, tol, rho;
    T beta_rho;            // PCG with the fused tail: rho / rho_prev of the step being closed (becomes beta when the step's norm is known)
    int done, nhist;
    int x_pending, pad_;   // x .+= alpha .* u of the last step has not been applied yet (it rides on the next sweep over u)
};

// Host-mapped (pinned, device-visible) mirror of the scalars the host needs after a step.  The
// closing finalise kernel of every step stores it with system scope and then publishes `seq`;
// the host polls `seq` instead of paying a D2H copy kernel + hipStreamSynchronize per iteration.
struct CgMirror {
    double res, prev_res, tol;
    int done, nhist;
    int tol_valid;
    int range;            // 1: the sum of squares of the step after `nhist` left the safe range; the host finishes that step
    unsigned long long seq;
};

struct mik_cg {
    mik_ctx *ctx = nullptr;
    const mik_csr *A = nullptr;
    int dtype = MIK_F64;
    int64_t n = 0;
    void *x = nullptr, *u = nullptr, *r = nullptr, *c = nullptr;
    const void *b = nullptr, *diag = nullptr;
    mik_mul_fn op_mul = nullptr;     // A as a callback (A == nullptr): mik_cg_create_op
    void *op_user = nullptr;
    mik_ldiv_fn pl_fn = nullptr;     // Pl as a callback
    void *pl_user = nullptr;
    void *dev = nullptr;       // CgDev<T>
    void *fin = nullptr;       // FinScratch<T>: wave sums + ticket of the spread level-2 reductions
    void *hist = nullptr;      // device history of one iterate_many call
    int64_t hist_cap = 0;
    void *seg_spmv = nullptr;  // one partial per row-block
    void *seg_vec = nullptr;   // one partial per vector segment
    void *seg_vec2 = nullptr;  // second reduction of the fused PCG tail (dot(Pl \\ r, r))
    bool pcg_fused = false;    // diagonal Pl on a CSR operator: c = Pl \\ r, rho ride on the tail; the head recomputes r ./ d (OpPcgUpdateR / OpPcgXpbyX)
    double residual = 0, prev_residual = 1, tol = 0;
    int64_t maxiter = 0, mv_products = 0;
    CgMirror *mirror = nullptr;      // host-mapped; same pointer is valid on the device
    unsigned long long seq = 0;      // steps enqueued so far (published by k_cg_fin_res)
    bool dev_done = false;           // device stopping flag known to be set
    bool head_ahead = false;         // the head of the next step (u, c, alpha) is on the stream already
    bool fuse_x = false;             // x .+= alpha .* u rides on the next u = r + beta u sweep (plain / Jacobi CG on a CSR operator)
    // optional in-loop timing of the SpMV launch (HIP events on the ctx stream)
    int profile = 0;               // 0 off, 1 = the SpMV launch, 2 = SpMV + the two vector sweeps of the step
    std::vector<hipEvent_t> ev;    // pairs (start, stop), recycled
    std::vector<int> ev_kind;      // per pair: 0 = SpMV, 1 = u = r + beta u, 2 = x / r update
    size_t ev_used = 0;
    double kern_ms[3] = {0, 0, 0};
    int64_t kern_launches[3] = {0, 0, 0};
};

#ifdef __HIPCC__
// Level 2 of a reduction spread over MIK_FIN_WGS single-wave workgroups (the single 1024-thread workgroup of
// level2_sum pulls its 64 k partials through ONE CU: ~10 us at 256^3).  Workgroup w plays virtual threads
// 64 w .. 64 w + 63 of the same 1024-thread shape (serial stride-1024 sums, then the wave tree); the workgroup that
// arrives last at the ticket adds the 16 wave sums left to right -- the order block_tree_1024 uses -- so the
// total is bit-identical.  Returns true in lane 0 of that last workgroup only.
constexpr int MIK_FIN_WGS = MIK_FIN_THREADS / 64;
template <typename T> struct FinScratch { T ws[MIK_FIN_WGS]; unsigned ticket; };

template <typename T> __device__ __forceinline__ bool level2_sum_spread(const T *__restrict__ S, int64_t m, FinScratch<T> *fs, T &tot)
{
    const int w = blockIdx.x, lane = threadIdx.x;              // blockDim.x == 64, gridDim.x == MIK_FIN_WGS
    T acc = T(0);
    int64_t j = 64 * (int64_t)w + lane;
    // (batches of 64 / 16 / 8 loads in flight, added in index order: 65,536 SpMV partials are ONE round trip per lane, the 16,384 of
    //  a vector sweep too -- a single wave per workgroup has the registers for it)
    for (; j + 63 * (int64_t)MIK_FIN_THREADS < m; j += 64 * (int64_t)MIK_FIN_THREADS) {
        T v[64];
#pragma unroll
        for (int q = 0; q < 64; ++q) v[q] = S[j + q * (int64_t)MIK_FIN_THREADS];
#pragma unroll
        for (int q = 0; q < 64; ++q) acc = acc + v[q];
    }
    for (; j + 15 * (int64_t)MIK_FIN_THREADS < m; j += 16 * (int64_t)MIK_FIN_THREADS) {
        T v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = S[j + q * (int64_t)MIK_FIN_THREADS];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = acc + v[q];
    }
    for (; j + 7 * (int64_t)MIK_FIN_THREADS < m; j += 8 * (int64_t)MIK_FIN_THREADS) {
        T v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = S[j + q * (int64_t)MIK_FIN_THREADS];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = acc + v[q];
    }
    for (; j < m; j += MIK_FIN_THREADS) acc = acc + S[j];
    acc = wave_tree(acc);
    bool last = false;
    if (lane == 0) {
        // hand-off without cache-wide fences (as longrow_store, mik_spmv.h): one write-through store, drained, then a relaxed ticket;
        // the last arrival reads the sums with loads that are served past its L1
        __hip_atomic_store(&fs->ws[w], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tk = __hip_atomic_fetch_add(&fs->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk == (unsigned)gridDim.x - 1u) {
            T t = __hip_atomic_load(&fs->ws[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 1; q < MIK_FIN_WGS; ++q) t = t + __hip_atomic_load(&fs->ws[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&fs->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tot = t;
            last = true;
        }
    }
    return last;
}


// residual = norm(r) of a row-partitioned step from the rank-ordered total of |r|^2 (src/cg.jl:61-62), history, beta and the stopping test
// of :36 for the next iterate() call; one thread.  A total outside the range of a safe sqrt(sum of squares) freezes the batch on every
// rank alike (identical totals): x and r of the step are final, the hosts finish it with the scaled norm over the partition.
template <typename T>
__device__ __forceinline__ void cgd_close_step(CgDev<T> *d, T tot, T *__restrict__ hist, long long it_next, long long maxiter, CgMirror *mirror,
                                               unsigned long long seq, int hist_index, int fuse_x)
{
    if (fuse_x) d->x_pending = 1;              // r of this step is final; its x update rides on the next sweep over u
    if (!mik_nrm_in_range(tot)) {
        d->done = 1; mirror->done = 0; mirror->nhist = hist_index; mirror->range = 1;
        __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const T prev = d->res;
    const T res = mik_sqrt(tot);
    d->rr = tot; d->prev_res = prev; d->res = res;
    d->beta = (res * res) / (prev * prev);
    hist[hist_index] = res;                   // step `hist_index` since the last wait (steps behind a stop are no-ops)
    const int nh = hist_index + 1;
    const int dn = (it_next >= maxiter || res <= d->tol) ? 1 : 0;
    if (dn) d->done = 1;
    mirror->res = (double)res; mirror->prev_res = (double)prev; mirror->done = dn; mirror->nhist = nh;
    __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif  // __HIPCC__

// Wait until the device has published step `it->seq` in the host-mapped mirror (bounded spin).
int cg_wait_mirror(mik_cg *it);

struct mik_comm;
struct mik_plink;
struct mik_cgd {
    mik_cg base;                 // reuses the single-GPU handle's buffers / mirror / scalars
    int rank = 0, nranks = 1;
    int64_t n_send = 0;
    const int *send_idx = nullptr;   // device: local indices to pack for the neighbours
    void *send_buf = nullptr;        // device: packed halo values (caller-owned, n_send entries)
    void *u_ext = nullptr;           // device: n_loc + n_ghost entries (u and its halo)
    int64_t n_ext = 0;
    void *dot_all = nullptr, *rr_all = nullptr;   // device: nranks scalars each (caller-owned comm buffers)
    double abstol = 0, reltol = 0;
    int initially_zero = 1;
    int64_t hist_total = 0;
    int64_t int_begin = 0, int_end = 0;   // row-blocks [int_begin, int_end) reference no halo column (mik_cgd_set_interior)
    // transport owned by the library (mik_cgd_set_halo_plan / mik_cgd_set_comm, csrc/mik_comm.hip)
    struct HaloSeg { int peer; int64_t off, cnt; };
    std::vector<HaloSeg> recv, send;      // offsets into the ghost tail of u_ext / into send_buf, in elements
    mik_comm *comm = nullptr;
    struct mik_plink *link = nullptr;     // landing buffer + peer mappings of the pushed halo (mik_cgd_ghost_export / mik_cgd_connect_ghosts; csrc/mik_comm.hip)
    bool ghosts = false;                  // the halo is pushed into the peers' landing buffers (mailbox transport) instead of ncclSend / ncclRecv
    bool initialised = false;             // mik_cgd_init ran
    // rows the neighbours need (send_idx) as at most two contiguous runs [a, b): when they are, u is updated there FIRST, packed and
    // put on the wire before the bulk of the u = r + beta u sweep runs (mik_cgd_set_halo_plan decides; n_early = 0: not applicable)
    int n_early = 0;
    bool early_merged = false;            // every send index occurs once: update + pack are one launch (k_cgd_early)
    int64_t early_a[2] = {0, 0}, early_b[2] = {0, 0};
    // the over-/underflow-safe norm across the partition (phases 20-24; csrc/mik_comm.hip cgd_norm_stage)
    double norm_scale = 1.0, norm_res = 0.0;
    int norm_fix_index = 0;
    int64_t norm_it_next = 0;
};

// waits for the last enqueued phase and copies the mirror; a frozen step (range) is reported in the copy, not as an error
int cgd_wait_raw(mik_cgd *it, struct CgMirror *m);
// what mik_cgd_wait does after a successful wait: history of the steps since the previous wait, handle scalars
int cgd_collect(mik_cgd *it, const struct CgMirror &m, double *residual, double *tol, int *done, double *history, int64_t cap, int64_t *steps);


// Device-driven links of a row partition (csrc/mik_comm.hip), as the row-partitioned GMRES uses them (csrc/mik_krylov.hip):
bool plink_ready(const mik_plink *pl);                   // connected, on a communicator whose mailboxes are connected
const mik_ctx *plink_ctx(const mik_plink *pl);
int plink_rank(const mik_plink *pl);
int plink_nranks(const mik_plink *pl);
int plink_check(mik_plink *pl, const char *who);         // MIK_ERR_HIP if a bounded wait of an earlier exchange expired
int plink_halo(mik_plink *pl, const void *send_buf, void *ghost);                                   // behind the pack kernel, on the ctx stream
int plink_fin_sum(mik_plink *pl, const void *partials, int64_t nseg, void *out_dev, int mode);      // level 2 + sum over the ranks (+ sqrt, inverse)
int plink_sum_vec(mik_plink *pl, void *vals_dev, int count);                                        // in place, rank order
int plink_gather(mik_plink *pl, void *all_dev);                                                     // all[rank] -> all[0 .. P)
// Modified Gram-Schmidt as the launch-lean chain over a link (n <= 1024 segments; k + 2 launches): h -> hd[0, k), nrm -> hd[k], 1 / nrm -> hd[k + 1]
int plink_mgs_lean(mik_plink *pl, int64_t n, int k, const void *V, int64_t ldv, void *w, void *hd, void *partials, bool vec, bool vecw, int hints);
