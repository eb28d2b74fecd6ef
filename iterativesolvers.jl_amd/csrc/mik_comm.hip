// mik_comm.hip -- the exchanges of the row-partitioned iterables INSIDE libmik.so.
//
// The reference is a serial library; the partition is new design: contiguous row blocks, one
// halo exchange of u per SpMV and two sums of one scalar per rank per CG step.  Two transports behind one step
// routine, so that a host (Julia through ccall, C, Python) runs multi-GPU cg! with ONE call per batch of steps:
//
//  * RCCL over xGMI, one process per GPU (mik_comm_create from an ncclUniqueId): the halo is ncclSend / ncclRecv on
//    a side stream (ordered by events, so the interior row-blocks of the SpMV run while it is in flight), the scalars
//    are ncclAllGather of one element per rank on the compute stream; every rank then adds the P partial sums in
//    rank order on the device (k_cgd_alpha / k_cgd_fin_res), so all ranks hold identical bits.  librccl is bound at
//    run time (dlopen): hosts that never go multi-GPU do not need it, and a process that already carries RCCL
//    (PyTorch-ROCm bundles one) shares that copy.
//  * an in-process group (mik_cgd_group_*): one host thread drives P ranks, each with its own ctx / device; halos and
//    scalars move by peer copies ordered with events (xGMI P2P when the ranks sit on different GPUs).  This is also
//    how the step routine is verified on a single-GPU box: P virtual ranks on one device, bit-exact against the
//    partition-aware oracle.
#include <dlfcn.h>

#include <algorithm>
#include <new>

#include "mik_kernels.h"
#include "mik_iter.h"
#include "mik_mail.h"

// ---------------------------------------------------------------------------------------------
// RCCL, bound at run time
// ---------------------------------------------------------------------------------------------
namespace {
struct NcclId { char internal[128]; };
using nccl_comm_t = void *;
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, NcclId, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};

Rccl *rccl()
{
    static Rccl R;
    static bool tried = false;
    if (tried) return R.h ? &R : nullptr;
    tried = true;
    // MIK_RCCL_LIB names the library explicitly (a host whose RCCL lives elsewhere); otherwise the copy the process already carries
    // (PyTorch-ROCm bundles one) or the system's
    const char *forced = getenv("MIK_RCCL_LIB");
    if (forced && *forced) R.h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (R.h) break;
        R.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!R.h) { const char *de = dlerror(); R.err = de ? de : "librccl.so not found"; return nullptr; }
    auto sym = [&](const char *s) { void *p = dlsym(R.h, s); if (!p) R.err = std::string("missing symbol ") + s; return p; };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.Send = (decltype(R.Send))sym("ncclSend");
    R.Recv = (decltype(R.Recv))sym("ncclRecv");
    R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
    R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.Send || !R.Recv || !R.AllGather || !R.GroupStart || !R.GroupEnd) {
        dlclose(R.h);
        R.h = nullptr;
        return nullptr;
    }
    return &R;
}
constexpr int NCCL_F32 = 7, NCCL_F64 = 8;   // ncclFloat32 / ncclFloat64 (rccl.h)
}  // namespace

// ---------------------------------------------------------------------------------------------
// Transport 3: the peer-mapped mailbox (SURVEY.md section 5 "backend B", section 8e)
// ---------------------------------------------------------------------------------------------
// A CG step couples the ranks at three points: the halo of u and two sums of one scalar per rank.  Over RCCL each of them is a
// collective launch (ncclSend/ncclRecv on a side stream ordered by two events; two dependent one-element ncclAllGather on the
// compute stream).  Here every rank owns a MAILBOX in fine-grained device memory that its peers map (hipIpcOpenMemHandle between
// processes) and write to over xGMI with ordinary stores:
//   * a scalar: the kernel that finalises the producing reduction stores {value, sequence number} into slot [kind][rank] of EVERY
//     peer's mailbox, then waits for the P slots of its own mailbox to carry this sequence number and adds the values in rank
//     order -- every rank the same additions, bit-identical scalars, no collective launch, no host;
//   * the halo: a push kernel copies the packed send buffer into the neighbours' LANDING BUFFERS (fine-grained memory owned by the library,
//     mik_plink below) and then publishes the exchange number in their mailboxes; on the receiving side k_halo_land waits for the flags and
//     copies the landed entries into the ghost tail of the extended vector with system-scope loads -- the SpMV reads what its own device wrote.
// Sequence numbers only grow, every rank enqueues the same sequence of exchanges, and two exchanges of one kind are always
// separated by one of another kind that needs every rank's contribution -- a slot is never overwritten before its reader took it
// (two parities per kind are kept anyway).  Every wait is bounded (MIK_MAILBOX_TIMEOUT_MS, default 10 s): a peer that died turns
// into MIK_ERR_HIP on the host instead of a hung queue.
struct mik_comm {
    mik_ctx *ctx = nullptr;
    int rank = 0, nranks = 1;
    nccl_comm_t nccl = nullptr;          // NULL: a world of one without the library
    hipStream_t side = nullptr;          // halo transfers (so that the interior SpMV overlaps them)
    hipEvent_t ev_packed = nullptr, ev_halo = nullptr;
    void *scratch = nullptr;             // device: nranks * MAX_COUNT scalars for mik_comm_allgather_sum
    void *scratch_host = nullptr;        // pinned mirror
    static constexpr int MAX_COUNT = 256;
    // mailbox transport
    MailBox *mail = nullptr;             // this rank's mailbox (fine-grained device memory)
    MailBox **peers_dev = nullptr;       // device: peer q's mailbox as mapped into this process (q = rank: mail)
    std::vector<MailBox *> peers;
    std::vector<void *> ipc_open;        // mappings to close
    // the peers' landing buffers as mapped into this process (mik_plink_connect).  Keyed by the 64-byte IPC handle, not by the rank: every link
    // on this communicator (a second iterable, a second solve) has a landing buffer of its own on every peer; and a handle must not be opened
    // twice in one process.
    struct GhostMap { int rank; unsigned char handle[64]; void *base; };
    std::vector<GhostMap> ghost_maps;
    bool mail_finegrained = false;       // hipExtMallocWithFlags(hipDeviceMallocFinegrained) succeeded for the mailbox
    std::vector<struct mik_plink *> links;   // live links on this communicator: orphaned (cm = NULL) when it is destroyed first
    bool mail_ready = false;
    unsigned long long mseq[MIK_MAIL_KINDS] = {0, 0, 0};   // exchanges enqueued so far, per kind
    unsigned long long vseq = 0;         // vector exchanges enqueued so far
    unsigned long long halo_no = 0;
    unsigned *mail_err = nullptr;        // pinned, device-mapped: a wait timed out
    unsigned long long timeout_ticks = 0;   // of the 100 MHz wall clock
    unsigned *push_ticket = nullptr;     // device: arrival counter of k_halo_push
};

namespace {
// all[rank] (this rank's partial, written by the kernel before on the stream) -> all[0 .. P) on every rank: what ncclAllGather of one
// element per rank did, as one single-wave launch
template <typename T>
__global__ __launch_bounds__(64) void k_mail_gather(MailBox *const *__restrict__ peers, int P, int rank, int kind, unsigned long long seq,
                                                    T *__restrict__ all, unsigned long long ticks, unsigned *__restrict__ err)
{
    const T mine = all[rank];
    (void)mail_exchange<T>(peers, P, rank, kind, seq, mine, all, ticks, err);
}

// The two scalar exchanges of a step INSIDE the kernels that finalise the producing reductions (no gather launch at all):
// level 2 of this rank's partials over MIK_FIN_WGS single-wave workgroups with a ticket, exactly as k_cgd_fin_slot; the wave that
// arrives last posts the rank's sum to every peer, collects the P sums and adds them in rank order.
//   k_cgd_fin_dot_mail: ... then dot(u, c) and alpha = residual^2 / dot(u, c) (src/cg.jl:55) -- what k_cgd_alpha / CoefAlphaRanks did;
//   k_cgd_fin_rr_mail:  ... then residual, beta, history and the stopping test -- what k_cgd_fin_res did (cgd_close_step).
template <typename T>
__global__ __launch_bounds__(64) void k_cgd_fin_dot_mail(const T *__restrict__ S, int64_t m, FinScratch<T> *fs, CgDev<T> *d, T *__restrict__ dot_all,
                                                          MailBox *const *__restrict__ peers, int P, int rank, unsigned long long seq,
                                                          unsigned long long ticks, unsigned *__restrict__ err)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) d->x_pending = 0;      // the sweep over u of this step's head applied it (OpXpbyX)
    if (d->done) return;                                             // identical on every rank: nobody posts, nobody waits
    T tot = T(0);
    int last = level2_sum_spread(S, m, fs, tot) ? 1 : 0;
    last = __shfl(last, 0);
    if (!last) return;
    tot = __shfl(tot, 0);
    const T v = mail_exchange<T>(peers, P, rank, 0, seq, tot, dot_all, ticks, err);
    const T sum = mail_rank_sum(v, P);
    if (threadIdx.x == 0) { d->dot_uc = sum; d->alpha = (d->res * d->res) / sum; }
}

template <typename T>
__global__ __launch_bounds__(64) void k_cgd_fin_rr_mail(const T *__restrict__ S, int64_t m, FinScratch<T> *fs, CgDev<T> *d, T *__restrict__ rr_all,
                                                         MailBox *const *__restrict__ peers, int P, int rank, unsigned long long seq,
                                                         unsigned long long ticks, unsigned *__restrict__ err, T *__restrict__ hist, long long it_next,
                                                         long long maxiter, CgMirror *mirror, unsigned long long step_seq, int hist_index, int fuse_x)
{
    if (d->done) {                                                   // a no-op step (the stopping test fired earlier in this batch): still publish
        if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&mirror->seq, step_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    T tot = T(0);
    int last = level2_sum_spread(S, m, fs, tot) ? 1 : 0;
    last = __shfl(last, 0);
    if (!last) return;
    tot = __shfl(tot, 0);
    const T v = mail_exchange<T>(peers, P, rank, 1, seq, tot, rr_all, ticks, err);
    const T sum = mail_rank_sum(v, P);
    if (threadIdx.x == 0) cgd_close_step<T>(d, sum, hist, it_next, maxiter, mirror, step_seq, hist_index, fuse_x);
}

// behind a kernel on the same stream: the kernel boundary orders that kernel's writes before the flag (no release fence: it would write
// this XCD's L2 back, ~4 us on the compute stream)
__global__ void k_mail_mark(unsigned long long *flag, unsigned long long v)
{
    __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_mail_wait_flag(const unsigned long long *flag, unsigned long long want, unsigned long long ticks, unsigned *err)
{
    if (!mail_wait(flag, want, ticks)) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the halo of this rank's neighbours: segment i of the packed send buffer -> dst[i] (a pointer into the peer's landing buffer), 16 bytes
// per lane where the alignment allows; the last workgroup to finish publishes the exchange number in the receivers' mailboxes
struct PushSegs {
    static constexpr int MAX = 8;
    void *dst[MAX];
    long long off[MAX], cnt[MAX];
    int peer[MAX];
    int n;
};
template <typename T>
__global__ __launch_bounds__(MIK_BLOCK) void k_halo_push(const T *__restrict__ send_buf, PushSegs sg, MailBox *const *__restrict__ peers, int rank,
                                                         unsigned long long halo_no, unsigned *__restrict__ ticket)
{
    constexpr int W = 16 / (int)sizeof(T);
    for (int i = 0; i < sg.n; ++i) {
        const T *src = send_buf + sg.off[i];
        T *dst = (T *)sg.dst[i];
        const long long cnt = sg.cnt[i];
        const bool wide = ((((size_t)src) | ((size_t)dst)) & 15) == 0;
        if (wide) {
            const long long nv = cnt / W, stride = (long long)gridDim.x * MIK_BLOCK;
            long long j = (long long)blockIdx.x * MIK_BLOCK + threadIdx.x;
            for (; j + 3 * stride < nv; j += 4 * stride) {          // four independent 16-byte copies in flight per lane
                const uint4 a = reinterpret_cast<const uint4 *>(src)[j], b2 = reinterpret_cast<const uint4 *>(src)[j + stride];
                const uint4 c2 = reinterpret_cast<const uint4 *>(src)[j + 2 * stride], d2 = reinterpret_cast<const uint4 *>(src)[j + 3 * stride];
                reinterpret_cast<uint4 *>(dst)[j] = a; reinterpret_cast<uint4 *>(dst)[j + stride] = b2;
                reinterpret_cast<uint4 *>(dst)[j + 2 * stride] = c2; reinterpret_cast<uint4 *>(dst)[j + 3 * stride] = d2;
            }
            for (; j < nv; j += stride) reinterpret_cast<uint4 *>(dst)[j] = reinterpret_cast<const uint4 *>(src)[j];
            for (long long k2 = nv * W + (long long)blockIdx.x * MIK_BLOCK + threadIdx.x; k2 < cnt; k2 += stride) dst[k2] = src[k2];
        } else {
            for (long long j = (long long)blockIdx.x * MIK_BLOCK + threadIdx.x; j < cnt; j += (long long)gridDim.x * MIK_BLOCK) dst[j] = src[j];
        }
    }
    __threadfence_system();                              // this thread's stores have reached the peers
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (tk == gridDim.x - 1u) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i < sg.n; ++i)
                __hip_atomic_store(&peers[sg.peer[i]]->halo_seq[rank], halo_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // behind every workgroup's fence
        }
    }
}

// vals[0 .. count) (this rank's partial sums, written by the kernel before on the stream) -> the sums over the ranks in rank order, in place
template <typename T>
__global__ __launch_bounds__(64) void k_mail_sum_vec(MailBox *const *__restrict__ peers, int P, int rank, unsigned long long seq0, T *__restrict__ vals, int count,
                                                     unsigned long long ticks, unsigned *__restrict__ err)
{
    for (int base = 0, c = 0; base < count; base += MIK_MAIL_VEC, ++c) {
        const int j = base + (int)threadIdx.x, cnt = min(MIK_MAIL_VEC, count - base);
        const T mine = j < count ? vals[j] : T(0);
        const T sum = mail_exchange_vec<T>(peers, P, rank, seq0 + (unsigned long long)c, mine, cnt, ticks, err);
        if (j < count) vals[j] = sum;
    }
}

// Level 2 of this rank's segment sums (the fixed 1024-thread shape of k_finalize_store), the exchange of the rank totals and their sum in rank
// order in ONE launch: what finalize -> D2H -> host all-gather -> H2D did for every projection of the row-partitioned GMRES.
// mode 0: out[0] = sum;  mode 1: out[0] = nrm = sqrt(sum), out[1] = 1 / nrm  (k_finalize_nrm_inv: NaN / 1 outside the safe range -- the host recomputes)
template <typename T>
__global__ __launch_bounds__(MIK_FIN_THREADS) void k_fin_sum_mail(const T *__restrict__ S, int64_t m, T *__restrict__ out, int mode, MailBox *const *__restrict__ peers,
                                                                   int P, int rank, unsigned long long seq, T *__restrict__ all, unsigned long long ticks,
                                                                   unsigned *__restrict__ err)
{
    __shared__ T lds16[16];
    __shared__ T tot_s;
    const T tot = level2_sum(S, m, lds16);
    if (threadIdx.x == 0) tot_s = tot;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const T v = mail_exchange<T>(peers, P, rank, 2, seq, tot_s, all, ticks, err);
    const T sum = mail_rank_sum(v, P);
    if (threadIdx.x == 0) {
        if (mode == 0) out[0] = sum;
        else {
            T nrm = mik_sqrt(sum);
            T inv = T(1) / nrm;
            if (!mik_nrm_in_range(sum)) { nrm = __builtin_nan(""); inv = T(1); }
            out[0] = nrm;
            out[1] = inv;
        }
    }
}

struct WaitPeers { int peer[PushSegs::MAX]; int n; };
// The halo has landed: wait for the senders' flags, then copy this exchange's half of the LANDING BUFFER into the ghost tail of the
// extended vector.  The landing buffer is fine-grained device memory owned by the library and written by the peers over xGMI; it is
// read here with system-scope loads (never served from a cache of this device), and the ghost tail is written by THIS device -- so
// the SpMV that follows reads halo data through the ordinary kernel-to-kernel visibility of its own device, whatever a cache of this
// device still holds of the previous exchange (peer-written coarse-grained memory became visible only if the next kernel's
// start invalidated the right lines).  U = the element's bit type (uint64 for fp64, uint32 for fp32).
template <typename U>
__global__ __launch_bounds__(MIK_BLOCK) void k_halo_land(const MailBox *__restrict__ mine, WaitPeers wp, unsigned long long halo_no, unsigned long long ticks,
                                                         unsigned *__restrict__ err, const U *__restrict__ land, U *__restrict__ ghost, long long count)
{
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    if ((int)threadIdx.x < wp.n && !mail_wait(&mine->halo_seq[wp.peer[threadIdx.x]], halo_no, ticks)) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        bad = 1;
    }
    __syncthreads();
    if (bad) return;
    const long long stride = (long long)gridDim.x * MIK_BLOCK;
    long long j = (long long)blockIdx.x * MIK_BLOCK + threadIdx.x;
    for (; j + 7 * stride < count; j += 8 * stride) {
        U v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __hip_atomic_load(land + j + q * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int q = 0; q < 8; ++q) ghost[j + q * stride] = v[q];
    }
    for (; j < count; j += stride) ghost[j] = __hip_atomic_load(land + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void k_halo_wait(const MailBox *__restrict__ mine, WaitPeers wp, unsigned long long halo_no, unsigned long long ticks,
                                                  unsigned *__restrict__ err)
{
    const int i = threadIdx.x;
    if (i < wp.n && !mail_wait(&mine->halo_seq[wp.peer[i]], halo_no, ticks)) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

#define MIK_NCCL(ctx, call)                                                                                        \
    do {                                                                                                           \
        int r_ = (call);                                                                                           \
        if (r_ != 0) {                                                                                             \
            Rccl *R_ = rccl();                                                                                     \
            return mik_fail((ctx), MIK_ERR_HIP, "%s failed: %s (%s:%d)", #call,                                    \
                            (R_ && R_->GetErrorString) ? R_->GetErrorString(r_) : "rccl error", __FILE__, __LINE__); \
        }                                                                                                          \
    } while (0)

static int mailbox_alloc(mik_comm *cm);
static int mailbox_check(mik_comm *cm, const char *who);

extern "C" int mik_comm_unique_id(void *id128)
{
    if (!id128) return MIK_ERR_INVALID;
    Rccl *R = rccl();
    if (!R) return mik_fail(nullptr, MIK_ERR_NOTIMPL, "mik_comm_unique_id: RCCL is not available in this process");
    NcclId id;
    if (R->GetUniqueId(&id) != 0) return mik_fail(nullptr, MIK_ERR_HIP, "ncclGetUniqueId failed");
    memcpy(id128, id.internal, 128);
    return MIK_OK;
}

extern "C" int mik_comm_create(mik_ctx *ctx, const void *id128, int rank, int nranks, mik_comm **out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return mik_fail(ctx, MIK_ERR_INVALID, "mik_comm_create: bad rank %d of %d", rank, nranks);
    // id128 = NULL with nranks > 1: a communicator without RCCL -- its ranks connect mailboxes (mik_comm_mailbox_export / _connect)
    // and landing buffers (mik_cgd_ghost_export / mik_cgd_connect_ghosts, mik_plink_*) instead
    mik_comm *cm = new (std::nothrow) mik_comm();
    if (!cm) return mik_fail(ctx, MIK_ERR_NOMEM, "mik_comm_create: host allocation failed");
    cm->ctx = ctx; cm->rank = rank; cm->nranks = nranks;
    auto bail = [&](int rc) { mik_comm_destroy(cm); return rc; };
    hipError_t e;
    (void)hipSetDevice(ctx->device);
    if ((e = hipStreamCreateWithFlags(&cm->side, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&cm->ev_packed, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&cm->ev_halo, hipEventDisableTiming)) != hipSuccess ||
        (e = hipMalloc(&cm->scratch, 8 * (size_t)nranks * mik_comm::MAX_COUNT)) != hipSuccess ||
        (e = hipHostMalloc(&cm->scratch_host, 8 * (size_t)nranks * mik_comm::MAX_COUNT, hipHostMallocDefault)) != hipSuccess)
        return bail(mik_fail(ctx, MIK_ERR_HIP, "mik_comm_create: %s", hipGetErrorString(e)));
    if (id128) {
        Rccl *R = rccl();
        if (!R) return bail(mik_fail(ctx, MIK_ERR_NOTIMPL, "mik_comm_create: RCCL is not available in this process (dlopen librccl.so failed)"));
        NcclId id;
        memcpy(id.internal, id128, 128);
        int r = R->CommInitRank(&cm->nccl, nranks, id, rank);
        if (r != 0) return bail(mik_fail(ctx, MIK_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, R->GetErrorString ? R->GetErrorString(r) : "?"));
    }
    // the local mailbox (the flags that order the side stream, at least).  Not fatal for a communicator that has RCCL: without it the
    // side stream is ordered by events, as until round 4
    if (nranks <= MIK_MAIL_MAXP) {
        const int rc = mailbox_alloc(cm);
        if (rc && !cm->nccl && nranks > 1) return bail(rc);      // (more ranks and neither RCCL nor a mailbox: nothing to talk through)
        if (rc) cm->mail_ready = false;                          // (mailbox_alloc has released whatever it had allocated)
    }
    *out = cm;
    return MIK_OK;
}

static void plink_orphan(struct mik_plink *pl);     // (defined with the struct below)

extern "C" int mik_comm_destroy(mik_comm *cm)
{
    if (!cm) return MIK_OK;
    for (struct mik_plink *pl : cm->links) plink_orphan(pl);    // a host may destroy the communicator before the iterables that hold links on it
    cm->links.clear();
    if (cm->ctx) { (void)hipSetDevice(cm->ctx->device); (void)hipStreamSynchronize(cm->ctx->stream); }
    if (cm->side) (void)hipStreamSynchronize(cm->side);
    if (cm->nccl) { Rccl *R = rccl(); if (R) (void)R->CommDestroy(cm->nccl); }
    if (cm->side) (void)hipStreamDestroy(cm->side);
    if (cm->ev_packed) (void)hipEventDestroy(cm->ev_packed);
    if (cm->ev_halo) (void)hipEventDestroy(cm->ev_halo);
    if (cm->scratch) (void)hipFree(cm->scratch);
    if (cm->scratch_host) (void)hipHostFree(cm->scratch_host);
    for (void *p : cm->ipc_open) (void)hipIpcCloseMemHandle(p);
    if (cm->mail) (void)hipFree(cm->mail);
    if (cm->peers_dev) (void)hipFree(cm->peers_dev);
    if (cm->push_ticket) (void)hipFree(cm->push_ticket);
    if (cm->mail_err) (void)hipHostFree(cm->mail_err);
    delete cm;
    return MIK_OK;
}

extern "C" int mik_comm_info(const mik_comm *cm, int *rank, int *nranks, int *uses_rccl)
{
    if (!cm) return MIK_ERR_INVALID;
    if (rank) *rank = cm->rank;
    if (nranks) *nranks = cm->nranks;
    if (uses_rccl) *uses_rccl = cm->nccl ? 1 : 0;
    return MIK_OK;
}

// values[0..count): this rank's partial sums (host scalars) -> ((p_0 + p_1) + p_2) + ... over the ranks, identically
// on every rank: the mik_reduce_fn contract (dot / norm^2 of a row-partitioned GMRES).  Blocks.
template <typename T> static int allgather_sum_impl(mik_comm *cm, int count, T *values)
{
    mik_ctx *ctx = cm->ctx;
    if (cm->nranks == 1 && !cm->nccl) return MIK_OK;
    T *dev = (T *)cm->scratch, *host = (T *)cm->scratch_host;
    MIK_HIP(ctx, mik_wait(ctx));                                       // the pinned staging buffer must be idle
    memcpy(host, values, sizeof(T) * count);
    if (!cm->nccl) {
        // a communicator without RCCL ("Transport 3"): the partial sums travel through the vector slots of the connected mailboxes and are
        // added in rank order by the one-wave exchange kernel -- the same bits; a bounded wait (MIK_ERR_HIP, never a hung queue)
        if (!cm->mail_ready) return mik_fail(ctx, MIK_ERR_INVALID, "mik_comm_allgather_sum: this communicator has neither RCCL nor connected mailboxes");
        MIK_HIP(ctx, hipMemcpyAsync(dev, host, sizeof(T) * count, hipMemcpyHostToDevice, ctx->stream));
        const unsigned long long seq0 = cm->vseq + 1;
        cm->vseq += (unsigned long long)((count + MIK_MAIL_VEC - 1) / MIK_MAIL_VEC);
        hipLaunchKernelGGL((k_mail_sum_vec<T>), dim3(1), dim3(64), 0, ctx->stream, (MailBox *const *)cm->peers_dev, cm->nranks, cm->rank, seq0, dev, count,
                           cm->timeout_ticks, cm->mail_err);
        MIK_LAUNCH_CHECK(ctx);
        MIK_HIP(ctx, hipMemcpyAsync(host, dev, sizeof(T) * count, hipMemcpyDeviceToHost, ctx->stream));
        MIK_HIP(ctx, mik_wait(ctx));
        MIK_TRY(mailbox_check(cm, "mik_comm_allgather_sum"));
        memcpy(values, host, sizeof(T) * count);
        return MIK_OK;
    }
    Rccl *R = rccl();
    MIK_HIP(ctx, hipMemcpyAsync(dev + (size_t)cm->rank * count, host, sizeof(T) * count, hipMemcpyHostToDevice, ctx->stream));
    MIK_NCCL(ctx, R->AllGather(dev + (size_t)cm->rank * count, dev, (size_t)count, sizeof(T) == 8 ? NCCL_F64 : NCCL_F32, cm->nccl, ctx->stream));
    MIK_HIP(ctx, hipMemcpyAsync(host, dev, sizeof(T) * count * cm->nranks, hipMemcpyDeviceToHost, ctx->stream));
    MIK_HIP(ctx, mik_wait(ctx));
    for (int j = 0; j < count; ++j) {
        T s = host[j];
        for (int p = 1; p < cm->nranks; ++p) s = s + host[(size_t)p * count + j];
        values[j] = s;
    }
    return MIK_OK;
}

extern "C" int mik_comm_allgather_sum(mik_comm *cm, int dtype, int count, void *values)
{
    if (!cm || count < 0 || (count && !values)) return MIK_ERR_INVALID;
    if (count > mik_comm::MAX_COUNT) return mik_fail(cm->ctx, MIK_ERR_NOTIMPL, "mik_comm_allgather_sum: at most %d values per call", mik_comm::MAX_COUNT);
    if (count == 0) return MIK_OK;
    if (dtype == MIK_F64) return allgather_sum_impl<double>(cm, count, (double *)values);
    if (dtype == MIK_F32) return allgather_sum_impl<float>(cm, count, (float *)values);
    return MIK_ERR_INVALID;
}

// Halo exchange of a packed send buffer into a ghost region on the ctx stream (the mik_halo_fn of a row-partitioned
// GMRES handle): segments as in mik_cgd_set_halo_plan.
extern "C" int mik_comm_halo(mik_comm *cm, int dtype, const void *send_buf, void *ghost, int n_recv, const int *recv_peer,
                             const int64_t *recv_off, const int64_t *recv_cnt, int n_send, const int *send_peer, const int64_t *send_off,
                             const int64_t *send_cnt)
{
    if (!cm || n_recv < 0 || n_send < 0) return MIK_ERR_INVALID;
    if (n_recv + n_send == 0) return MIK_OK;
    if (!cm->nccl) return mik_fail(cm->ctx, MIK_ERR_INVALID, "mik_comm_halo: a world of one has no neighbours");
    Rccl *R = rccl();
    const size_t es = mik_dtype_size(dtype);
    const int nt = dtype == MIK_F64 ? NCCL_F64 : NCCL_F32;
    MIK_NCCL(cm->ctx, R->GroupStart());
    for (int i = 0; i < n_recv; ++i)
        MIK_NCCL(cm->ctx, R->Recv((unsigned char *)ghost + es * (size_t)recv_off[i], (size_t)recv_cnt[i], nt, recv_peer[i], cm->nccl, cm->ctx->stream));
    for (int i = 0; i < n_send; ++i)
        MIK_NCCL(cm->ctx, R->Send((const unsigned char *)send_buf + es * (size_t)send_off[i], (size_t)send_cnt[i], nt, send_peer[i], cm->nccl, cm->ctx->stream));
    MIK_NCCL(cm->ctx, R->GroupEnd());
    return MIK_OK;
}

// ---------------------------------------------------------------------------------------------
// row-partitioned CG driven by the library
// ---------------------------------------------------------------------------------------------
// (a rank may be its own neighbour -- a periodic direction cut into one slab: RCCL sends to self inside a group)
extern "C" int mik_cgd_set_halo_plan(mik_cgd *it, int n_recv, const int *recv_peer, const int64_t *recv_off, const int64_t *recv_cnt,
                                     int n_send, const int *send_peer, const int64_t *send_off, const int64_t *send_cnt)
{
    if (!it || n_recv < 0 || n_send < 0 || (n_recv && (!recv_peer || !recv_off || !recv_cnt)) || (n_send && (!send_peer || !send_off || !send_cnt)))
        return MIK_ERR_INVALID;
    mik_ctx *ctx = it->base.ctx;
    const int64_t n_ghost = it->n_ext - it->base.n;
    if (it->link) { (void)mik_plink_destroy(it->link); it->link = nullptr; it->ghosts = false; }     // a new plan: the landing buffer and the peers' targets follow it
    it->recv.clear();
    it->send.clear();
    for (int i = 0; i < n_recv; ++i) {
        if (recv_peer[i] < 0 || recv_peer[i] >= it->nranks || recv_off[i] < 0 || recv_cnt[i] < 0 || recv_off[i] + recv_cnt[i] > n_ghost)
            return mik_fail(ctx, MIK_ERR_INVALID, "mik_cgd_set_halo_plan: receive segment %d out of range", i);
        it->recv.push_back({recv_peer[i], recv_off[i], recv_cnt[i]});
    }
    for (int i = 0; i < n_send; ++i) {
        if (send_peer[i] < 0 || send_peer[i] >= it->nranks || send_off[i] < 0 || send_cnt[i] < 0 || send_off[i] + send_cnt[i] > it->n_send)
            return mik_fail(ctx, MIK_ERR_INVALID, "mik_cgd_set_halo_plan: send segment %d out of range", i);
        it->send.push_back({send_peer[i], send_off[i], send_cnt[i]});
    }
    // Do the rows the neighbours need form at most two contiguous runs (the bottom and top planes of a slab)?  Then the step
    // updates and packs them first and puts the halo on the wire before the bulk of the sweep over u (cgd_enqueue_head).
    it->n_early = 0;
    it->early_merged = false;
    if (it->n_send > 0 && it->n_send <= (1 << 26) && (it->base.ctx->tuning[MIK_KNOB_CG_STEP] & 4) == 0) {       // development knob MIK_KNOB_CG_STEP bit 2: halo after the whole sweep
        std::vector<int> idx((size_t)it->n_send);
        if (hipMemcpy(idx.data(), it->send_idx, sizeof(int) * idx.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            std::vector<int> srt(idx);
            std::sort(srt.begin(), srt.end());
            srt.erase(std::unique(srt.begin(), srt.end()), srt.end());
            int runs = 0;
            int64_t a[2] = {0, 0}, b[2] = {0, 0};
            bool ok = !srt.empty() && srt.front() >= 0 && (int64_t)srt.back() < it->base.n;
            for (size_t q = 0; ok && q < srt.size();) {
                size_t e = q + 1;
                while (e < srt.size() && srt[e] == srt[e - 1] + 1) ++e;
                if (runs == 2) { ok = false; break; }
                a[runs] = srt[q]; b[runs] = (int64_t)srt[e - 1] + 1; ++runs;
                q = e;
            }
            if (ok && runs > 0 && (b[0] - a[0]) + (b[1] - a[1]) <= it->base.n / 4) {
                it->n_early = runs;
                it->early_merged = (int64_t)srt.size() == it->n_send;      // no index sent twice
                for (int q = 0; q < 2; ++q) { it->early_a[q] = a[q]; it->early_b[q] = b[q]; }
            }
        } else {
            (void)hipGetLastError();
        }
    }
    return MIK_OK;
}

extern "C" int mik_cgd_halo_early(const mik_cgd *it, int *runs, int64_t *rows, int *merged)
{
    if (!it) return MIK_ERR_INVALID;
    int64_t total = 0;
    for (int q = 0; q < it->n_early; ++q) total += it->early_b[q] - it->early_a[q];
    if (runs) *runs = it->n_early;
    if (rows) *rows = total;
    if (merged) *merged = it->early_merged ? 1 : 0;
    return MIK_OK;
}

extern "C" int mik_cgd_set_comm(mik_cgd *it, mik_comm *cm)
{
    if (!it) return MIK_ERR_INVALID;
    if (cm && (cm->rank != it->rank || cm->nranks != it->nranks || cm->ctx != it->base.ctx))
        return mik_fail(it->base.ctx, MIK_ERR_MISMATCH, "mik_cgd_set_comm: communicator is rank %d of %d, the iterable rank %d of %d (or another ctx)",
                        cm->rank, cm->nranks, it->rank, it->nranks);
    it->comm = cm;
    return MIK_OK;
}

// ---- mailbox: allocation, export, connection ----------------------------------------------------------------------------------
static int mailbox_alloc(mik_comm *cm)
{
    if (cm->mail) return MIK_OK;
    mik_ctx *ctx = cm->ctx;
    if (cm->nranks > MIK_MAIL_MAXP) return mik_fail(ctx, MIK_ERR_NOTIMPL, "mailbox transport: at most %d ranks", MIK_MAIL_MAXP);
    (void)hipSetDevice(ctx->device);
    // a failed attempt leaves nothing behind (a later mik_comm_mailbox_export re-runs this and must not leak the first attempt's pieces)
    auto undo = [&](int rc) {
        (void)hipGetLastError();
        if (cm->mail) { (void)hipFree(cm->mail); cm->mail = nullptr; }
        if (cm->peers_dev) { (void)hipFree(cm->peers_dev); cm->peers_dev = nullptr; }
        if (cm->push_ticket) { (void)hipFree(cm->push_ticket); cm->push_ticket = nullptr; }
        if (cm->mail_err) { (void)hipHostFree(cm->mail_err); cm->mail_err = nullptr; }
        cm->mail_finegrained = false;
        return rc;
    };
    hipError_t e = hipExtMallocWithFlags((void **)&cm->mail, sizeof(MailBox), hipDeviceMallocFinegrained);
    cm->mail_finegrained = e == hipSuccess;
    if (e != hipSuccess) { (void)hipGetLastError(); cm->mail = nullptr; e = hipMalloc((void **)&cm->mail, sizeof(MailBox)); }   // system-scope atomics still reach memory
    if (e != hipSuccess) { cm->mail = nullptr; return undo(mik_fail(ctx, MIK_ERR_NOMEM, "mailbox transport: %s", hipGetErrorString(e))); }
    if ((e = hipMemset(cm->mail, 0, sizeof(MailBox))) != hipSuccess ||
        (e = hipMalloc((void **)&cm->peers_dev, sizeof(MailBox *) * MIK_MAIL_MAXP)) != hipSuccess ||
        (e = hipMalloc((void **)&cm->push_ticket, sizeof(unsigned) * 80)) != hipSuccess || (e = hipMemset(cm->push_ticket, 0, sizeof(unsigned) * 80)) != hipSuccess ||
        (e = hipHostMalloc((void **)&cm->mail_err, sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
        return undo(mik_fail(ctx, MIK_ERR_HIP, "mailbox transport: %s", hipGetErrorString(e)));
    *cm->mail_err = 0;
    cm->peers.assign((size_t)cm->nranks, nullptr);
    cm->peers[(size_t)cm->rank] = cm->mail;
    if ((e = hipMemcpy(cm->peers_dev, cm->peers.data(), sizeof(MailBox *) * cm->peers.size(), hipMemcpyHostToDevice)) != hipSuccess)
        return undo(mik_fail(ctx, MIK_ERR_HIP, "mailbox transport: %s", hipGetErrorString(e)));
    const char *ms = getenv("MIK_MAILBOX_TIMEOUT_MS");
    const double msv = ms && atof(ms) > 0 ? atof(ms) : 10000.0;
    cm->timeout_ticks = (unsigned long long)(msv * 1e5);               // wall_clock64 runs at 100 MHz
    cm->mail_ready = cm->nranks == 1;                                  // a world of one has nobody to connect to
    return MIK_OK;
}

static int mailbox_check(mik_comm *cm, const char *who)
{
    if (cm && cm->mail_err && *cm->mail_err) {
        *cm->mail_err = 0;
        return mik_fail(cm->ctx, MIK_ERR_HIP, "%s: a mailbox wait timed out (a peer rank stopped, or the ranks enqueued different exchange sequences)", who);
    }
    return MIK_OK;
}

extern "C" int mik_comm_mailbox_export(mik_comm *cm, void *handle64)
{
    if (!cm || !handle64) return MIK_ERR_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    MIK_TRY(mailbox_alloc(cm));
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof(h));
    if (cm->nranks > 1) MIK_HIP(cm->ctx, hipIpcGetMemHandle(&h, cm->mail));
    memcpy(handle64, &h, 64);
    return MIK_OK;
}

extern "C" int mik_comm_mailbox_connect(mik_comm *cm, const void *handles)
{
    if (!cm || (cm->nranks > 1 && !handles)) return MIK_ERR_INVALID;
    MIK_TRY(mailbox_alloc(cm));
    mik_ctx *ctx = cm->ctx;
    (void)hipSetDevice(ctx->device);
    for (int q = 0; q < cm->nranks; ++q) {
        if (q == cm->rank || cm->peers[(size_t)q]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const unsigned char *)handles + 64 * (size_t)q, 64);
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return mik_fail(ctx, MIK_ERR_HIP, "mik_comm_mailbox_connect: hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(e));
        cm->ipc_open.push_back(p);
        cm->peers[(size_t)q] = (MailBox *)p;
    }
    MIK_HIP(ctx, hipMemcpy(cm->peers_dev, cm->peers.data(), sizeof(MailBox *) * cm->peers.size(), hipMemcpyHostToDevice));
    cm->mail_ready = true;
    return MIK_OK;
}

// ---------------------------------------------------------------------------------------------
// mik_plink: the device-driven links of ONE row partition (halo plan + landing buffer + peer mappings) on a communicator's mailboxes.
// The row-partitioned CG iterable builds one from its halo plan (mik_cgd_ghost_export / mik_cgd_connect_ghosts); the row-partitioned
// GMRES iterable takes one in mik_partition.link and then needs no host callbacks (csrc/mik_krylov.hip).
// ---------------------------------------------------------------------------------------------
struct mik_plink {
    mik_comm *cm = nullptr;
    int dtype = MIK_F64;
    size_t es = 8;
    int64_t n_ghost = 0;
    std::vector<mik_cgd::HaloSeg> recv, send;   // offsets into the ghost region / into the packed send buffer, in elements
    void *land = nullptr;                // fine-grained device memory: 2 (exchange parity) x n_ghost elements; the PEERS write it, this rank copies it out
    size_t land_stride = 0;              // bytes between the two parities
    bool land_finegrained = false;
    std::vector<unsigned char *> send_dst;      // per send segment: parity 0 of its place in the receiver's landing buffer, as mapped here
    std::vector<size_t> send_stride;            // ... and the receiver's parity stride
    bool connected = false;
    bool symmetric = true;           // send peers == receive peers (mik_plink_create)
};

static void plink_orphan(mik_plink *pl) { pl->cm = nullptr; pl->connected = false; }

static size_t plink_stride(int64_t n_ghost, size_t es) { return ((size_t)std::max<int64_t>(n_ghost, 1) * es + 255) / 256 * 256; }

extern "C" int mik_plink_destroy(mik_plink *pl)
{
    if (!pl) return MIK_OK;
    if (pl->cm && pl->cm->ctx) { (void)hipSetDevice(pl->cm->ctx->device); (void)hipStreamSynchronize(pl->cm->ctx->stream); if (pl->cm->side) (void)hipStreamSynchronize(pl->cm->side); }
    if (pl->cm) pl->cm->links.erase(std::remove(pl->cm->links.begin(), pl->cm->links.end(), pl), pl->cm->links.end());
    if (pl->land) (void)hipFree(pl->land);
    delete pl;
    return MIK_OK;
}

extern "C" int mik_plink_create(mik_comm *cm, int dtype, int64_t n_ghost, int n_recv, const int *recv_peer, const int64_t *recv_off, const int64_t *recv_cnt,
                                int n_send, const int *send_peer, const int64_t *send_off, const int64_t *send_cnt, mik_plink **out)
{
    if (!cm || !out || n_ghost < 0 || n_recv < 0 || n_send < 0 || (dtype != MIK_F64 && dtype != MIK_F32) ||
        (n_recv && (!recv_peer || !recv_off || !recv_cnt)) || (n_send && (!send_peer || !send_off || !send_cnt)))
        return MIK_ERR_INVALID;
    *out = nullptr;
    mik_ctx *ctx = cm->ctx;
    if (n_recv > PushSegs::MAX || n_send > PushSegs::MAX) return mik_fail(ctx, MIK_ERR_NOTIMPL, "mik_plink_create: more than %d halo segments per direction", PushSegs::MAX);
    MIK_TRY(mailbox_alloc(cm));
    mik_plink *pl = new (std::nothrow) mik_plink();
    if (!pl) return mik_fail(ctx, MIK_ERR_NOMEM, "mik_plink_create: host allocation failed");
    pl->cm = cm; pl->dtype = dtype; pl->es = mik_dtype_size(dtype); pl->n_ghost = n_ghost;
    for (int i = 0; i < n_recv; ++i) {
        if (recv_peer[i] < 0 || recv_peer[i] >= cm->nranks || recv_off[i] < 0 || recv_cnt[i] < 0 || recv_off[i] + recv_cnt[i] > n_ghost) {
            delete pl;
            return mik_fail(ctx, MIK_ERR_INVALID, "mik_plink_create: receive segment %d out of range", i);
        }
        pl->recv.push_back({recv_peer[i], recv_off[i], recv_cnt[i]});
    }
    for (int i = 0; i < n_send; ++i) {
        if (send_peer[i] < 0 || send_peer[i] >= cm->nranks || send_off[i] < 0 || send_cnt[i] < 0) {
            delete pl;
            return mik_fail(ctx, MIK_ERR_INVALID, "mik_plink_create: send segment %d out of range", i);
        }
        pl->send.push_back({send_peer[i], send_off[i], send_cnt[i]});
    }
    (void)hipSetDevice(ctx->device);
    pl->land_stride = plink_stride(n_ghost, pl->es);
    hipError_t e = hipExtMallocWithFlags(&pl->land, 2 * pl->land_stride, hipDeviceMallocFinegrained);
    pl->land_finegrained = e == hipSuccess;
    if (e != hipSuccess) { (void)hipGetLastError(); pl->land = nullptr; e = hipMalloc(&pl->land, 2 * pl->land_stride); }   // (read with system-scope loads either way)
    if (e != hipSuccess || (e = hipMemset(pl->land, 0, 2 * pl->land_stride)) != hipSuccess) {
        (void)hipGetLastError();
        mik_plink_destroy(pl);
        return mik_fail(ctx, MIK_ERR_NOMEM, "mik_plink_create: landing buffer (%zu bytes): %s", 2 * pl->land_stride, hipGetErrorString(e));
    }
    pl->connected = pl->send.empty() && pl->recv.empty();
    {   // Does every peer this rank sends to also send to this rank?  Then a rank cannot run two exchanges ahead of a neighbour by itself (its
        // own landing of exchange n + 1 waits for that neighbour's push, which follows the neighbour's landing of exchange n), and the two-parity
        // landing buffer needs no consumed-acknowledgement.  An asymmetric plan (structurally unsymmetric operators) is safe only because
        // cg! / gmres! put a sum over ALL ranks between two products; mik_plink_exchange, which has no such sum, refuses it.
        std::vector<int> sp, rp;
        for (auto &sg : pl->send) sp.push_back(sg.peer);
        for (auto &sg : pl->recv) rp.push_back(sg.peer);
        std::sort(sp.begin(), sp.end()); sp.erase(std::unique(sp.begin(), sp.end()), sp.end());
        std::sort(rp.begin(), rp.end()); rp.erase(std::unique(rp.begin(), rp.end()), rp.end());
        pl->symmetric = sp == rp;
    }
    cm->links.push_back(pl);
    *out = pl;
    return MIK_OK;
}

// 64-byte HIP IPC handle of this rank's landing buffer (what its neighbours map to push their halo segments)
extern "C" int mik_plink_export(mik_plink *pl, void *handle64)
{
    if (!pl || !pl->cm || !handle64) return MIK_ERR_INVALID;
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof(h));
    (void)hipSetDevice(pl->cm->ctx->device);
    if (pl->cm->nranks > 1) MIK_HIP(pl->cm->ctx, hipIpcGetMemHandle(&h, pl->land));
    memcpy(handle64, &h, 64);
    return MIK_OK;
}

// handles: per RANK the landing-buffer handle of its link (mik_plink_export there; ignored for this rank itself); ghost_counts: per rank the
// n_ghost of its link (the parity stride of its landing buffer follows from it); dst_elem: per SEND segment, the element of the receiver's
// GHOST REGION at which the segment lands (the offset of the matching receive segment there).
extern "C" int mik_plink_connect(mik_plink *pl, const void *handles, const int64_t *ghost_counts, const int64_t *dst_elem)
{
    if (!pl || !pl->cm) return MIK_ERR_INVALID;
    mik_comm *cm = pl->cm;
    mik_ctx *ctx = cm->ctx;
    if (!pl->send.empty() && (!dst_elem || !ghost_counts)) return MIK_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    pl->send_dst.clear();
    pl->send_stride.clear();
    for (size_t i = 0; i < pl->send.size(); ++i) {
        const int q = pl->send[i].peer;
        if (dst_elem[i] < 0 || dst_elem[i] + pl->send[i].cnt > ghost_counts[q])
            return mik_fail(ctx, MIK_ERR_INVALID, "mik_plink_connect: send segment %zu does not fit rank %d's ghost region", i, q);
        unsigned char *base = nullptr;
        if (q == cm->rank) base = (unsigned char *)pl->land;
        else {
            if (!handles) return MIK_ERR_INVALID;
            const unsigned char *hq = (const unsigned char *)handles + 64 * (size_t)q;
            void *mapped = nullptr;
            for (const mik_comm::GhostMap &g : cm->ghost_maps)
                if (g.rank == q && memcmp(g.handle, hq, 64) == 0) { mapped = g.base; break; }
            if (!mapped) {
                hipIpcMemHandle_t h;
                memcpy(&h, hq, 64);
                hipError_t e = hipIpcOpenMemHandle(&mapped, h, hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) return mik_fail(ctx, MIK_ERR_HIP, "mik_plink_connect: hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(e));
                cm->ipc_open.push_back(mapped);
                mik_comm::GhostMap g;
                g.rank = q; memcpy(g.handle, hq, 64); g.base = mapped;
                cm->ghost_maps.push_back(g);
            }
            base = (unsigned char *)mapped;
        }
        pl->send_dst.push_back(base + pl->es * (size_t)dst_elem[i]);
        pl->send_stride.push_back(plink_stride(ghost_counts[q], pl->es));
    }
    pl->connected = true;
    return MIK_OK;
}

extern "C" int mik_plink_info(const mik_plink *pl, int *connected, int *finegrained, int64_t *n_ghost)
{
    if (!pl) return MIK_ERR_INVALID;
    if (connected) *connected = pl->connected ? 1 : 0;
    if (finegrained) *finegrained = pl->land_finegrained ? 1 : 0;
    if (n_ghost) *n_ghost = pl->n_ghost;
    return MIK_OK;
}

// exchange no. `no` of the link: the packed send buffer -> the neighbours' landing buffers (half no & 1) + their flags, on stream `s`
static int plink_push(mik_plink *pl, const void *send_buf, unsigned long long no, hipStream_t s)
{
    mik_comm *cm = pl->cm;
    mik_ctx *ctx = cm->ctx;
    PushSegs sg{};
    int64_t total = 0;
    sg.n = (int)pl->send.size();
    if (sg.n == 0) return MIK_OK;
    for (int i = 0; i < sg.n; ++i) {
        sg.dst[i] = pl->send_dst[(size_t)i] + (no & 1ull) * pl->send_stride[(size_t)i];
        sg.off[i] = pl->send[(size_t)i].off; sg.cnt[i] = pl->send[(size_t)i].cnt; sg.peer[i] = pl->send[(size_t)i].peer;
        total += pl->send[(size_t)i].cnt;
    }
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(128, (total * (int64_t)pl->es / 64 + MIK_BLOCK - 1) / MIK_BLOCK));
    if (pl->dtype == MIK_F64)
        hipLaunchKernelGGL((k_halo_push<double>), dim3(grid), dim3(MIK_BLOCK), 0, s, (const double *)send_buf, sg, (MailBox *const *)cm->peers_dev, cm->rank, no, cm->push_ticket);
    else
        hipLaunchKernelGGL((k_halo_push<float>), dim3(grid), dim3(MIK_BLOCK), 0, s, (const float *)send_buf, sg, (MailBox *const *)cm->peers_dev, cm->rank, no, cm->push_ticket);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// ... and on the receiving side: wait for the senders' flags of exchange `no`, copy its half of the landing buffer into `ghost`
static int plink_land(mik_plink *pl, void *ghost, unsigned long long no, hipStream_t s)
{
    mik_comm *cm = pl->cm;
    mik_ctx *ctx = cm->ctx;
    WaitPeers wp{};
    wp.n = (int)pl->recv.size();
    if (wp.n == 0) return MIK_OK;
    for (int i = 0; i < wp.n; ++i) wp.peer[i] = pl->recv[(size_t)i].peer;
    const unsigned char *src = (const unsigned char *)pl->land + (no & 1ull) * pl->land_stride;
    const long long cnt = (long long)pl->n_ghost;
    const int grid = (int)std::max<long long>(1, std::min<long long>(64, (cnt + 8 * MIK_BLOCK - 1) / (8 * MIK_BLOCK)));
    if (pl->dtype == MIK_F64)
        hipLaunchKernelGGL((k_halo_land<unsigned long long>), dim3(grid), dim3(MIK_BLOCK), 0, s, (const MailBox *)cm->mail, wp, no, cm->timeout_ticks, cm->mail_err,
                           (const unsigned long long *)src, (unsigned long long *)ghost, cnt);
    else
        hipLaunchKernelGGL((k_halo_land<unsigned>), dim3(grid), dim3(MIK_BLOCK), 0, s, (const MailBox *)cm->mail, wp, no, cm->timeout_ticks, cm->mail_err,
                           (const unsigned *)src, (unsigned *)ghost, cnt);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// ---- what csrc/mik_krylov.hip calls for a row-partitioned GMRES with a link (declared in mik_iter.h) -------------------------------
bool plink_ready(const mik_plink *pl) { return pl && pl->cm && pl->connected && pl->cm->mail_ready; }
const mik_ctx *plink_ctx(const mik_plink *pl) { return pl->cm ? pl->cm->ctx : nullptr; }
int plink_rank(const mik_plink *pl) { return pl->cm ? pl->cm->rank : -1; }
int plink_nranks(const mik_plink *pl) { return pl->cm ? pl->cm->nranks : 0; }

PlinkMail plink_mail(const mik_plink *pl)
{
    const mik_comm *cm = pl->cm;
    return PlinkMail{(MailBox *const *)cm->peers_dev, cm->nranks, cm->rank, cm->timeout_ticks, cm->mail_err};
}
unsigned long long plink_next_vec_tag(mik_plink *pl) { return ++pl->cm->vseq; }

int plink_check(mik_plink *pl, const char *who) { return pl->cm ? mailbox_check(pl->cm, who) : MIK_ERR_INVALID; }

// the halo of one SpMV, entirely on the compute stream (behind the pack kernel): push, land
int plink_halo(mik_plink *pl, const void *send_buf, void *ghost)
{
    if (!pl->cm) return MIK_ERR_INVALID;          // the communicator was destroyed before the iterable that holds this link
    mik_comm *cm = pl->cm;
    if (pl->send.empty() && pl->recv.empty()) return MIK_OK;
    const unsigned long long no = ++cm->halo_no;
    MIK_TRY(plink_push(pl, send_buf, no, cm->ctx->stream));
    return plink_land(pl, ghost, no, cm->ctx->stream);
}

// One halo exchange through the link, blocking: the mik_halo_fn of a host that drives a row-partitioned operator itself, and what a
// transport self-test times (bench.py --gpus N).  Bounded like every mailbox wait.
extern "C" int mik_plink_exchange(mik_plink *pl, const void *send_buf, void *ghost)
{
    if (!pl || !pl->cm) return MIK_ERR_INVALID;
    mik_comm *cm = pl->cm;
    if (!pl->connected || !cm->mail_ready) return mik_fail(cm->ctx, MIK_ERR_INVALID, "mik_plink_exchange: the link (or its communicator's mailbox) is not connected");
    if ((!pl->send.empty() && !send_buf) || (!pl->recv.empty() && !ghost)) return MIK_ERR_INVALID;
    if (!pl->symmetric)
        return mik_fail(cm->ctx, MIK_ERR_NOTIMPL, "mik_plink_exchange: this rank's send and receive peers differ; back-to-back exchanges of such a plan need a sum over all "
                                                  "ranks between them (what cg! / gmres! have) -- the landing buffers carry no consumed-acknowledgement");
    (void)hipSetDevice(cm->ctx->device);
    MIK_TRY(plink_halo(pl, send_buf, ghost));
    MIK_HIP(cm->ctx, mik_wait(cm->ctx));
    return mailbox_check(cm, "mik_plink_exchange");
}

// level 2 of this rank's `nseg` segment sums + the sum over the ranks in rank order, one launch; mode 0: out[0] = sum, 1: out[0] = sqrt(sum), out[1] = 1 / out[0]
int plink_fin_sum(mik_plink *pl, const void *partials, int64_t nseg, void *out_dev, int mode)
{
    if (!pl->cm) return MIK_ERR_INVALID;          // the communicator was destroyed before the iterable that holds this link
    mik_comm *cm = pl->cm;
    mik_ctx *ctx = cm->ctx;
    const unsigned long long seq = ++cm->mseq[2];
    if (pl->dtype == MIK_F64)
        hipLaunchKernelGGL((k_fin_sum_mail<double>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const double *)partials, nseg, (double *)out_dev, mode,
                           (MailBox *const *)cm->peers_dev, cm->nranks, cm->rank, seq, (double *)cm->scratch, cm->timeout_ticks, cm->mail_err);
    else
        hipLaunchKernelGGL((k_fin_sum_mail<float>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const float *)partials, nseg, (float *)out_dev, mode,
                           (MailBox *const *)cm->peers_dev, cm->nranks, cm->rank, seq, (float *)cm->scratch, cm->timeout_ticks, cm->mail_err);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// Modified Gram-Schmidt over a link as the LAUNCH-LEAN chain (orthogonalize_enqueue's form for n <= 1024 segments, csrc/mik_krylov.hip): every
// pass finalises the previous pass's reduction itself AND exchanges it (k_map_pro<..., MailSum>), k + 2 launches per Arnoldi column instead of
// 2 k + 3.  Leaves h in hd[0, k), nrm in hd[k], 1 / nrm in hd[k + 1] (NaN / 1 outside the safe range: w unscaled, the caller recomputes).
// Every workgroup of a pass spins on the mailbox, so ranks that SHARE a GPU must all fit on it: the caller keeps this form to <= 256 segments.
template <typename T>
static int plink_mgs_lean_t(mik_plink *pl, int64_t n, int k, const T *V, int64_t ldv, T *w, T *hd, T *part, bool vec, bool vecw, int hints)
{
    mik_comm *cm = pl->cm;
    mik_ctx *ctx = cm->ctx;
    const int m = (int)mik_nseg<T>(n);
    T *P2[2] = {part, part + 1024};
    auto xch = [&]() { return MailSum{(MailBox *const *)cm->peers_dev, cm->nranks, cm->rank, ++cm->mseq[2], cm->timeout_ticks, cm->mail_err}; };
    if (k > 0) {
        OpDot<T> d0{V, w};
        MIK_TRY((launch_map<T>(ctx, n, d0, vec, P2[0], nullptr)));
        for (int i = 0; i + 1 < k; ++i) {
            OpMgsPass<T, false> op{w, V + (int64_t)i * ldv, V + (int64_t)(i + 1) * ldv, coef_val<T>(T(0)), hints};
            MIK_TRY((launch_map_pro<T, 1>(ctx, n, op, vec, P2[(i + 1) & 1], (const T *)P2[i & 1], m, hd + i, xch())));
        }
        OpMgsPass<T, true> last{w, V + (int64_t)(k - 1) * ldv, nullptr, coef_val<T>(T(0)), hints};
        MIK_TRY((launch_map_pro<T, 1>(ctx, n, last, vec, P2[k & 1], (const T *)P2[(k - 1) & 1], m, hd + k - 1, xch())));
    } else {
        OpDot<T> dn{w, w};
        MIK_TRY((launch_map<T>(ctx, n, dn, vecw, P2[0], nullptr)));
    }
    OpScal<T> sc{w, coef_val<T>(T(0))};                                // w .*= inv(norm(w))  src/orthogonalize.jl:75-76
    return launch_map_pro<T, 2>(ctx, n, sc, vecw, (T *)nullptr, (const T *)P2[k & 1], m, hd + k, xch());
}

int plink_mgs_lean(mik_plink *pl, int64_t n, int k, const void *V, int64_t ldv, void *w, void *hd, void *partials, bool vec, bool vecw, int hints)
{
    if (!pl->cm) return MIK_ERR_INVALID;
    return pl->dtype == MIK_F64 ? plink_mgs_lean_t<double>(pl, n, k, (const double *)V, ldv, (double *)w, (double *)hd, (double *)partials, vec, vecw, hints)
                                : plink_mgs_lean_t<float>(pl, n, k, (const float *)V, ldv, (float *)w, (float *)hd, (float *)partials, vec, vecw, hints);
}

// vals_dev[0 .. count): this rank's partial sums -> the sums over the ranks in rank order, in place, one one-wave launch
int plink_sum_vec(mik_plink *pl, void *vals_dev, int count)
{
    if (!pl->cm) return MIK_ERR_INVALID;          // the communicator was destroyed before the iterable that holds this link
    mik_comm *cm = pl->cm;
    mik_ctx *ctx = cm->ctx;
    if (count <= 0) return MIK_OK;
    const unsigned long long seq0 = cm->vseq + 1;
    cm->vseq += (unsigned long long)((count + MIK_MAIL_VEC - 1) / MIK_MAIL_VEC);
    if (pl->dtype == MIK_F64)
        hipLaunchKernelGGL((k_mail_sum_vec<double>), dim3(1), dim3(64), 0, ctx->stream, (MailBox *const *)cm->peers_dev, cm->nranks, cm->rank, seq0, (double *)vals_dev, count,
                           cm->timeout_ticks, cm->mail_err);
    else
        hipLaunchKernelGGL((k_mail_sum_vec<float>), dim3(1), dim3(64), 0, ctx->stream, (MailBox *const *)cm->peers_dev, cm->nranks, cm->rank, seq0, (float *)vals_dev, count,
                           cm->timeout_ticks, cm->mail_err);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// all_dev[rank] (written by the kernel before on the stream) -> all_dev[0 .. P) on every rank
int plink_gather(mik_plink *pl, void *all_dev)
{
    if (!pl->cm) return MIK_ERR_INVALID;          // the communicator was destroyed before the iterable that holds this link
    mik_comm *cm = pl->cm;
    mik_ctx *ctx = cm->ctx;
    const unsigned long long seq = ++cm->mseq[2];
    if (pl->dtype == MIK_F64)
        hipLaunchKernelGGL((k_mail_gather<double>), dim3(1), dim3(64), 0, ctx->stream, (MailBox *const *)cm->peers_dev, cm->nranks, cm->rank, 2, seq, (double *)all_dev,
                           cm->timeout_ticks, cm->mail_err);
    else
        hipLaunchKernelGGL((k_mail_gather<float>), dim3(1), dim3(64), 0, ctx->stream, (MailBox *const *)cm->peers_dev, cm->nranks, cm->rank, 2, seq, (float *)all_dev,
                           cm->timeout_ticks, cm->mail_err);
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}

// ---- the row-partitioned CG iterable on the same links ---------------------------------------------------------------------------
// After mik_cgd_set_halo_plan + mik_cgd_set_comm: allocate this rank's landing buffer and hand out its IPC handle ...
extern "C" int mik_cgd_ghost_export(mik_cgd *it, void *handle64)
{
    if (!it || !it->comm || !handle64) return MIK_ERR_INVALID;
    mik_ctx *ctx = it->base.ctx;
    if (!it->link) {
        std::vector<int> rp, sp;
        std::vector<int64_t> ro, rc, so, sc;
        for (const auto &g : it->recv) { rp.push_back(g.peer); ro.push_back(g.off); rc.push_back(g.cnt); }
        for (const auto &g : it->send) { sp.push_back(g.peer); so.push_back(g.off); sc.push_back(g.cnt); }
        MIK_TRY(mik_plink_create(it->comm, it->base.dtype, it->n_ext - it->base.n, (int)rp.size(), rp.data(), ro.data(), rc.data(), (int)sp.size(), sp.data(), so.data(),
                                 sc.data(), &it->link));
    }
    (void)ctx;
    return mik_plink_export(it->link, handle64);
}

// ... and, once every rank has exported, connect (collective; arguments as mik_plink_connect)
extern "C" int mik_cgd_connect_ghosts(mik_cgd *it, const void *handles, const int64_t *ghost_counts, const int64_t *dst_elem)
{
    if (!it || !it->comm) return MIK_ERR_INVALID;
    if (!it->link) { unsigned char tmp[64]; MIK_TRY(mik_cgd_ghost_export(it, tmp)); }
    MIK_TRY(mik_plink_connect(it->link, handles, ghost_counts, dst_elem));
    it->ghosts = true;
    return MIK_OK;
}

extern "C" int mik_comm_mailbox_info(const mik_comm *cm, int *ready, int *finegrained)
{
    if (!cm) return MIK_ERR_INVALID;
    if (ready) *ready = cm->mail_ready ? 1 : 0;
    if (finegrained) *finegrained = cm->mail && cm->mail_finegrained ? 1 : 0;   // recorded when it was allocated (the plain hipMalloc fall-back reports 0)
    return MIK_OK;
}

// The halo of u (or of x during init).  Three steps, split so that the compute-stream work that overlaps the exchange can be
// enqueued BEFORE the host spends its tens of microseconds inside RCCL:
//   halo_mark  "the send buffer is packed" on the compute stream;
//   halo_issue the transfer on the side stream, ordered behind the mark;
//   halo_end   the compute stream waits for the ghost region.
// With a mailbox the two orderings are FLAGS: a one-thread kernel stores the exchange number behind the pack kernel, a one-thread
// kernel on the side stream waits for it; the side stream (or, for pushed halos, the sender's push kernel) stores it in the
// receiver's mailbox and a one-wave kernel on the compute stream waits there.  hipEventRecord / hipStreamWaitEvent each cost the
// compute stream a ~6 us hole between two kernels (profiles/r03_dist_selfhalo_timeline.txt); a 1-thread launch costs ~2.
// MIK_KNOB_TRANSPORT, bit 0: events.  Every waiting kernel is submitted after the kernel that satisfies it, so streams that share a
// hardware queue cannot deadlock.
static bool halo_flags(const mik_cgd *it) { return it->comm && it->comm->mail && (it->base.ctx->tuning[MIK_KNOB_TRANSPORT] & 1) == 0; }
static bool halo_p2p(const mik_cgd *it) { return it->comm && it->comm->mail_ready && it->ghosts && it->link; }
static bool halo_any(const mik_cgd *it)
{
    return it->comm && (it->comm->nccl || halo_p2p(it)) && !(it->recv.empty() && it->send.empty());
}

static int halo_mark(mik_cgd *it)
{
    mik_comm *cm = it->comm;
    if (!halo_any(it)) return MIK_OK;
    mik_ctx *ctx = it->base.ctx;
    cm->halo_no += 1;
    if (halo_flags(it)) {
        hipLaunchKernelGGL(k_mail_mark, dim3(1), dim3(1), 0, ctx->stream, &cm->mail->packed_seq, cm->halo_no);
        MIK_LAUNCH_CHECK(ctx);
    } else {
        MIK_HIP(ctx, hipEventRecord(cm->ev_packed, ctx->stream));
    }
    return MIK_OK;
}

static int halo_issue(mik_cgd *it, bool *pending)
{
    *pending = false;
    mik_comm *cm = it->comm;
    mik_ctx *ctx = it->base.ctx;
    if (!halo_any(it)) return MIK_OK;
    const bool flags = halo_flags(it);
    if (flags) {
        hipLaunchKernelGGL(k_mail_wait_flag, dim3(1), dim3(1), 0, cm->side, (const unsigned long long *)&cm->mail->packed_seq, cm->halo_no, cm->timeout_ticks, cm->mail_err);
        MIK_LAUNCH_CHECK(ctx);
    } else {
        MIK_HIP(ctx, hipStreamWaitEvent(cm->side, cm->ev_packed, 0));
    }
    const size_t es = mik_dtype_size(it->base.dtype);
    if (halo_p2p(it)) {
        MIK_TRY(plink_push(it->link, it->send_buf, cm->halo_no, cm->side));
    } else {
        Rccl *R = rccl();
        const int nt = it->base.dtype == MIK_F64 ? NCCL_F64 : NCCL_F32;
        unsigned char *ghost = (unsigned char *)it->u_ext + es * (size_t)it->base.n;
        MIK_NCCL(ctx, R->GroupStart());
        for (const auto &sg : it->recv) MIK_NCCL(ctx, R->Recv(ghost + es * (size_t)sg.off, (size_t)sg.cnt, nt, sg.peer, cm->nccl, cm->side));
        for (const auto &sg : it->send) MIK_NCCL(ctx, R->Send((const unsigned char *)it->send_buf + es * (size_t)sg.off, (size_t)sg.cnt, nt, sg.peer, cm->nccl, cm->side));
        MIK_NCCL(ctx, R->GroupEnd());
        if (flags) {      // "every receive of exchange no. halo_no has landed", under this rank's own name
            hipLaunchKernelGGL(k_mail_mark, dim3(1), dim3(1), 0, cm->side, &cm->mail->halo_seq[cm->rank], cm->halo_no);
            MIK_LAUNCH_CHECK(ctx);
        } else {
            MIK_HIP(ctx, hipEventRecord(cm->ev_halo, cm->side));
        }
    }
    *pending = true;
    return MIK_OK;
}

static int halo_begin(mik_cgd *it, bool *pending)
{
    MIK_TRY(halo_mark(it));
    return halo_issue(it, pending);
}

static int halo_end(mik_cgd *it, bool pending)
{
    if (!pending) return MIK_OK;
    mik_comm *cm = it->comm;
    mik_ctx *ctx = it->base.ctx;
    if (halo_p2p(it)) {
        // wait for the neighbours' flags and copy the landed halo into the ghost tail of u_ext (k_halo_land)
        MIK_TRY(plink_land(it->link, (unsigned char *)it->u_ext + mik_dtype_size(it->base.dtype) * (size_t)it->base.n, cm->halo_no, ctx->stream));
    } else if (halo_flags(it)) {
        WaitPeers wp{};
        wp.n = 1; wp.peer[0] = cm->rank;
        hipLaunchKernelGGL(k_halo_wait, dim3(1), dim3(64), 0, ctx->stream, (const MailBox *)cm->mail, wp, cm->halo_no, cm->timeout_ticks, cm->mail_err);
        MIK_LAUNCH_CHECK(ctx);
    } else {
        MIK_HIP(ctx, hipStreamWaitEvent(ctx->stream, cm->ev_halo, 0));
    }
    return MIK_OK;
}

// slot [rank] of `all` -> every rank's `all` (one scalar per rank), on the compute stream.  kind: the mailbox lane (0 dot, 1 |r|^2, 2 other)
static int gather_scalar(mik_cgd *it, void *all, int kind)
{
    mik_comm *cm = it->comm;
    if (!cm) return MIK_OK;
    mik_ctx *ctx = it->base.ctx;
    // MIK_KNOB_TRANSPORT, bit 1: the scalars over RCCL although a mailbox is connected; bit 2: through the mailbox even in a world of one
    if (cm->mail_ready && (ctx->tuning[MIK_KNOB_TRANSPORT] & 2) == 0 && (it->nranks > 1 || (ctx->tuning[MIK_KNOB_TRANSPORT] & 4) != 0)) {
        const unsigned long long seq = ++cm->mseq[kind];
        if (it->base.dtype == MIK_F64)
            hipLaunchKernelGGL((k_mail_gather<double>), dim3(1), dim3(64), 0, ctx->stream, (MailBox *const *)cm->peers_dev, it->nranks, it->rank, kind, seq, (double *)all,
                               cm->timeout_ticks, cm->mail_err);
        else
            hipLaunchKernelGGL((k_mail_gather<float>), dim3(1), dim3(64), 0, ctx->stream, (MailBox *const *)cm->peers_dev, it->nranks, it->rank, kind, seq, (float *)all,
                               cm->timeout_ticks, cm->mail_err);
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    if (!cm->nccl) return MIK_OK;
    Rccl *R = rccl();
    const size_t es = mik_dtype_size(it->base.dtype);
    MIK_NCCL(ctx, R->AllGather((const unsigned char *)all + es * (size_t)it->rank, all, 1, it->base.dtype == MIK_F64 ? NCCL_F64 : NCCL_F32, cm->nccl, ctx->stream));
    return MIK_OK;
}

// ---- the over-/underflow-safe norm of r across the partition (include/mik.h "Norms") ---------------------------------------
// Every rank saw the same out-of-range total, so every rank comes here.  Stage 0: local max |r_i| into rr_all[rank]; [gather];
// stage 1: amax = max over the ranks (0 / Inf / NaN are the norm as they are), s = 2^-exponent(amax) -- the SAME exact power of
// two on every rank -- then the local tree sum of (r_i s)^2 into rr_all[rank]; [gather]; stage 2: the rank sums added in rank
// order, norm = sqrt(t') / s.  The oracle's safe_nrm_ with a partition (oracle/orc_impl.inc) is this arithmetic.
template <typename T> static int cgd_read_slots(mik_cgd *it, std::vector<T> &v)
{
    mik_ctx *ctx = it->base.ctx;
    v.resize((size_t)it->nranks);
    MIK_HIP(ctx, hipMemcpyAsync(v.data(), it->rr_all, sizeof(T) * v.size(), hipMemcpyDeviceToHost, ctx->stream));
    MIK_HIP(ctx, mik_wait(ctx));
    return MIK_OK;
}

// returns *direct = true when the norm is known after stage 1 (no second pass; identical decision on every rank)
template <typename T> static int cgd_norm_stage1(mik_cgd *it, bool *direct)
{
    std::vector<T> v;
    MIK_TRY(cgd_read_slots<T>(it, v));
    T amax = T(0);
    for (T a : v) amax = (a > amax || a != a) ? a : amax;
    *direct = amax == T(0) || amax != amax || amax > std::numeric_limits<T>::max();
    if (*direct) { it->norm_res = (double)amax; return MIK_OK; }
    int e;
    (void)std::frexp((double)amax, &e);
    e = std::max(-NrmRange<T>::EC, std::min(NrmRange<T>::EC, e));
    it->norm_scale = std::ldexp(1.0, -e);
    return mik_cgd_phase(it, 22, 0);
}

template <typename T> static int cgd_norm_stage2(mik_cgd *it)
{
    std::vector<T> v;
    MIK_TRY(cgd_read_slots<T>(it, v));
    T t2 = v[0];
    for (size_t q = 1; q < v.size(); ++q) t2 = t2 + v[q];
    const T sinv = (T)(1.0 / it->norm_scale);              // exact: a power of two
    it->norm_res = (double)((T)std::sqrt(t2) * sinv);
    return MIK_OK;
}

static int cgd_norm_stage1_any(mik_cgd *it, bool *direct) { return it->base.dtype == MIK_F64 ? cgd_norm_stage1<double>(it, direct) : cgd_norm_stage1<float>(it, direct); }
static int cgd_norm_stage2_any(mik_cgd *it) { return it->base.dtype == MIK_F64 ? cgd_norm_stage2<double>(it) : cgd_norm_stage2<float>(it); }

static int cgd_require_transport(mik_cgd *it, const char *who)
{
    if (it->nranks > 1 && (!it->comm || !(it->comm->nccl || it->comm->mail_ready)))
        return mik_fail(it->base.ctx, MIK_ERR_INVALID, "%s: %d ranks need a communicator (mik_cgd_set_comm) with RCCL or a connected mailbox, or the in-process group calls", who, it->nranks);
    if (it->nranks > 1 && !it->comm->nccl && !(it->recv.empty() && it->send.empty()) && !it->ghosts)
        return mik_fail(it->base.ctx, MIK_ERR_INVALID, "%s: a communicator without RCCL needs the peers' ghost regions (mik_cgd_connect_ghosts)", who);
    return MIK_OK;
}

static int comm_scaled_norm(mik_cgd *it)
{
    bool direct = false;
    MIK_TRY(mik_cgd_phase(it, 20, 0));
    MIK_TRY(gather_scalar(it, it->rr_all, 2));
    MIK_TRY(cgd_norm_stage1_any(it, &direct));
    if (direct) return MIK_OK;
    MIK_TRY(gather_scalar(it, it->rr_all, 2));
    return cgd_norm_stage2_any(it);
}

// cg_iterator! (src/cg.jl:120-155) over the partition: phases 10, [halo of x], 11, [gather |r|^2], 12, then one wait.
extern "C" int mik_cgd_init(mik_cgd *it, double *residual, double *tol)
{
    if (!it) return MIK_ERR_INVALID;
    MIK_TRY(cgd_require_transport(it, "mik_cgd_init"));
    MIK_TRY(mik_cgd_phase(it, 10, 0));
    bool pending = false;
    if (!it->initially_zero) MIK_TRY(halo_begin(it, &pending));
    MIK_TRY(halo_end(it, pending));
    MIK_TRY(mik_cgd_phase(it, 11, 0));
    MIK_TRY(gather_scalar(it, it->rr_all, 2));
    MIK_TRY(mik_cgd_phase(it, 12, 0));
    int done = 0;
    int64_t steps = 0;
    CgMirror m;
    MIK_TRY(cgd_wait_raw(it, &m));
    if (m.range) {                                                  // norm(r) outside the safe range on every rank alike: common scale
        MIK_TRY(comm_scaled_norm(it));
        MIK_TRY(mik_cgd_phase(it, 24, 0));
        MIK_TRY(cgd_wait_raw(it, &m));
    }
    MIK_TRY(mailbox_check(it->comm, "mik_cgd_init"));
    MIK_TRY(cgd_collect(it, m, residual, tol, &done, nullptr, 0, &steps));
    it->initialised = true;
    return MIK_OK;
}

// one iterate() (src/cg.jl:43-66) of this rank, enqueued without any host synchronisation, in two halves: the HEAD writes
// only u (with its halo), c and this rank's dot slot -- all of its inputs are final once the previous tail has run -- so the
// head of the step after a call is enqueued before the host waits (as in the single-GPU path, cg_enqueue_head); every rank
// takes the same decision (same iteration counts, identical stopping scalars), so the RCCL call sequences stay aligned.
// the step's two scalar exchanges inside the finalising kernels: a connected mailbox, more than one rank (or MIK_KNOB_TRANSPORT bit 2),
// not switched to the one-wave gather launches (bit 3) or to RCCL (bit 1)
static bool mail_fused(const mik_cgd *it)
{
    const mik_comm *cm = it->comm;
    const int k = it->base.ctx->tuning[MIK_KNOB_TRANSPORT];
    return cm && cm->mail_ready && (k & 2) == 0 && (k & 8) == 0 && (it->nranks > 1 || (k & 4) != 0);
}

template <typename T> static int mail_fin_dot(mik_cgd *it)
{
    mik_comm *cm = it->comm;
    mik_cg &bs = it->base;
    const unsigned long long seq = ++cm->mseq[0];
    hipLaunchKernelGGL((k_cgd_fin_dot_mail<T>), dim3(MIK_FIN_WGS), dim3(64), 0, bs.ctx->stream, (const T *)bs.seg_spmv, (int64_t)mik_spmv_nwg(bs.n), (FinScratch<T> *)bs.fin,
                       (CgDev<T> *)bs.dev, (T *)it->dot_all, (MailBox *const *)cm->peers_dev, it->nranks, it->rank, seq, cm->timeout_ticks, cm->mail_err);
    MIK_LAUNCH_CHECK(bs.ctx);
    return MIK_OK;
}

template <typename T> static int mail_fin_rr(mik_cgd *it, int64_t iteration)
{
    mik_comm *cm = it->comm;
    mik_cg &bs = it->base;
    if (it->hist_total >= bs.hist_cap) return mik_fail(bs.ctx, MIK_ERR_INVALID, "mik_cgd_iterate_many: more than %lld steps enqueued without a wait", (long long)bs.hist_cap);
    it->hist_total += 1;
    bs.seq += 1;
    const unsigned long long seq = ++cm->mseq[1];
    hipLaunchKernelGGL((k_cgd_fin_rr_mail<T>), dim3(MIK_FIN_WGS), dim3(64), 0, bs.ctx->stream, (const T *)bs.seg_vec, mik_nseg<T>(bs.n), (FinScratch<T> *)bs.fin,
                       (CgDev<T> *)bs.dev, (T *)it->rr_all, (MailBox *const *)cm->peers_dev, it->nranks, it->rank, seq, cm->timeout_ticks, cm->mail_err, (T *)bs.hist,
                       (long long)(iteration + 1), (long long)bs.maxiter, bs.mirror, bs.seq, (int)(it->hist_total - 1), bs.fuse_x ? 1 : 0);
    MIK_LAUNCH_CHECK(bs.ctx);
    return MIK_OK;
}

// the local dot(u, c) of the head and its sum over the ranks: phase `with_fin` (SpMV + finaliser into the rank's slot) and a gather, or
// phase `without` and the finaliser that exchanges by itself
static int head_spmv_dot(mik_cgd *it, int with_fin, int without, int64_t iteration)
{
    if (mail_fused(it)) {
        MIK_TRY(mik_cgd_phase(it, without, iteration));
        return it->base.dtype == MIK_F64 ? mail_fin_dot<double>(it) : mail_fin_dot<float>(it);
    }
    MIK_TRY(mik_cgd_phase(it, with_fin, iteration));
    return gather_scalar(it, it->dot_all, 0);
}

static int cgd_enqueue_head(mik_cgd *it, int64_t iteration)
{
    const bool early = it->n_early > 0 && halo_any(it);
    bool pending = false;
    if (early && halo_flags(it)) {
        // Flags instead of events (a mailbox exists): the boundary rows are updated and packed first, a one-thread launch publishes
        // "packed", the halo travels on the side stream
        // underneath the bulk of the sweep over u (95 us at 16.7 M rows against ~20-30 us for the two 2 MB planes), ONE one-wave
        // launch waits for it, and the SpMV is ONE launch over all row-blocks -- no interior / boundary split: every launch costs the
        // compute stream ~5 us whatever it does (profiles/r04_dist_selfhalo_timeline_*.txt), and the split bought overlap that the
        // sweep already provides.
        MIK_TRY(mik_cgd_phase(it, it->early_merged ? 9 : 7, iteration));   // u on the rows the neighbours need; pack
        MIK_TRY(halo_mark(it));                                     // "packed": a one-thread launch (the pack kernel publishing the flag itself --
                                                                    // write-through stores, a two-level ticket of its 2,048 workgroups -- cost it 8 us
                                                                    // more than this launch costs: 15.8 us against 8.1 + 5.7, round 4)
        MIK_TRY(mik_cgd_phase(it, 8, iteration));                   // the bulk of the sweep over u is on the compute stream ...
        MIK_TRY(halo_issue(it, &pending));                          // ... before the host enters RCCL
        MIK_TRY(halo_end(it, pending));
        return head_spmv_dot(it, 1, 14, iteration);                 // c = A u, local dot(u, c), summed over the ranks
    }
    if (early) {
        MIK_TRY(mik_cgd_phase(it, it->early_merged ? 9 : 7, iteration));   // u on the rows the neighbours need; pack
        MIK_TRY(halo_mark(it));
        MIK_TRY(mik_cgd_phase(it, 8, iteration));                   // the bulk of the sweep over u ...
        if (it->int_end > it->int_begin) {
            MIK_TRY(mik_cgd_phase(it, 4, iteration));               // ... and the interior row-blocks are on the compute stream
            MIK_TRY(halo_issue(it, &pending));                 // before the host enters RCCL: the halo travels underneath them
            MIK_TRY(halo_end(it, pending));
            return head_spmv_dot(it, 5, 13, iteration);             // boundary row-blocks + local dot(u, c), summed over the ranks
        }
        MIK_TRY(halo_issue(it, &pending));
    } else {
        MIK_TRY(mik_cgd_phase(it, 0, iteration));                   // u = r + beta u; pack the halo
        MIK_TRY(halo_begin(it, &pending));
    }
    if (pending && it->int_end > it->int_begin) {
        MIK_TRY(mik_cgd_phase(it, 4, iteration));                   // interior row-blocks while the halo is in flight
        MIK_TRY(halo_end(it, pending));
        return head_spmv_dot(it, 5, 13, iteration);                 // boundary row-blocks + local dot(u, c), summed over the ranks
    }
    MIK_TRY(halo_end(it, pending));
    return head_spmv_dot(it, 1, 14, iteration);
}

static int cgd_enqueue_tail(mik_cgd *it, int64_t iteration)
{
    if (mail_fused(it)) {
        MIK_TRY(mik_cgd_phase(it, 16, iteration));                  // x, r update with the alpha k_cgd_fin_dot_mail stored; local |r|^2 partials
        return it->base.dtype == MIK_F64 ? mail_fin_rr<double>(it, iteration) : mail_fin_rr<float>(it, iteration);   // ... summed over the ranks; residual, stopping test
    }
    MIK_TRY(mik_cgd_phase(it, 2, iteration));                       // alpha; x, r update; local |r|^2
    MIK_TRY(gather_scalar(it, it->rr_all, 1));
    return mik_cgd_phase(it, 3, iteration);                         // residual, beta, stopping test
}

// Up to max_steps iterate() calls of this rank with ONE host wait (every rank makes the same call; the stopping test
// runs on the device from identical scalars, so all ranks execute the same number of steps).
extern "C" int mik_cgd_iterate_many(mik_cgd *it, int64_t iteration, int64_t max_steps, double *residuals, int64_t *steps_done)
{
    if (!it || !steps_done || iteration < 0) return MIK_ERR_INVALID;
    *steps_done = 0;
    MIK_TRY(cgd_require_transport(it, "mik_cgd_iterate_many"));
    if (!it->initialised) return mik_fail(it->base.ctx, MIK_ERR_INVALID, "mik_cgd_iterate_many: call mik_cgd_init first");
    mik_cg &bs = it->base;
    if (max_steps <= 0 || iteration >= bs.maxiter || bs.residual <= bs.tol) return MIK_OK;      // done(it, iteration), src/cg.jl:36
    max_steps = std::min<int64_t>(std::min<int64_t>(max_steps, bs.maxiter - iteration), bs.hist_cap);
    const bool ahead_ok = bs.ctx->tuning[MIK_KNOB_NO_LOOKAHEAD] == 0 && iteration + max_steps < bs.maxiter;      // MIK_KNOB_NO_LOOKAHEAD: nothing ahead of the host
    bs.fuse_x = (bs.ctx->tuning[MIK_KNOB_CG_STEP] & 1) == 0;                              // x .+= alpha .* u rides on the next sweep over u (as in mik_cg_*)
    CgMirror m;
    for (int64_t j0 = 0;;) {
        for (int64_t j = j0; j < max_steps; ++j) {
            if (!bs.head_ahead) MIK_TRY(cgd_enqueue_head(it, iteration + j));
            bs.head_ahead = false;
            MIK_TRY(cgd_enqueue_tail(it, iteration + j));
        }
        if (ahead_ok) MIK_TRY(cgd_enqueue_head(it, iteration + max_steps));
        else if (bs.fuse_x) MIK_TRY(mik_cgd_phase(it, 6, iteration + max_steps - 1));   // no sweep over u follows: apply the last x update now
        MIK_TRY(cgd_wait_raw(it, &m));
        if (!m.range) { bs.head_ahead = ahead_ok && !m.done; break; }  // stopped: the head ahead was a no-op on every rank
        // Step m.nhist of this call updated x and r, but |r|^2 summed over the ranks left the range in which sqrt(sum of squares)
        // is safe: every rank froze its batch on the same total; finish that step with the norm rescaled across the ranks and
        // go on (the single-GPU iterable does the same on its own, cg_iterate_many_impl).
        bs.head_ahead = false;
        // The frozen step's x .+= alpha .* u has been applied by now whatever came behind it: by the head of the next (no-op) step of
        // the batch, by the head enqueued ahead (ahead_ok) or by the flush of phase 6 -- but only a TAIL clears the pending flag, and
        // none follows the head ahead when the frozen step was the last of the batch.  Left set, the fresh head of the next call
        // would add the same alpha u to x a second time: clear it behind everything that is enqueued.
        if (bs.fuse_x) MIK_TRY(mik_cgd_phase(it, 25, 0));
        MIK_TRY(comm_scaled_norm(it));
        it->norm_fix_index = (int)m.nhist;
        it->norm_it_next = iteration + m.nhist + 1;
        MIK_TRY(mik_cgd_phase(it, 23, 0));
        MIK_TRY(cgd_wait_raw(it, &m));
        j0 = m.nhist;
        if (m.done || j0 >= max_steps) break;
    }
    int done = 0;
    MIK_TRY(mailbox_check(it->comm, "mik_cgd_iterate_many"));
    return cgd_collect(it, m, nullptr, nullptr, &done, residuals, max_steps, steps_done);
}

// ---------------------------------------------------------------------------------------------
// in-process group: one host thread, P ranks (one ctx each; same or different devices)
// ---------------------------------------------------------------------------------------------
namespace {
struct GroupEvents {
    std::vector<hipEvent_t> packed, halo, dot, rr;
    std::vector<hipStream_t> side;
};

struct GroupState {
    int P = 0;
    std::vector<mik_cgd *> its;
    GroupEvents ev;
    ~GroupState()
    {
        for (auto *v : {&ev.packed, &ev.halo, &ev.dot, &ev.rr})
            for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e);
        for (hipStream_t s : ev.side) if (s) (void)hipStreamDestroy(s);
    }
    bool peer_access = true;         // hipDeviceEnablePeerAccess succeeded between every pair of devices
};

// groups are keyed by their rank-0 handle and live until mik_cgd_group_release (or process exit)
std::vector<GroupState *> g_groups;

GroupState *find_group(mik_cgd **its, int P)
{
    for (GroupState *g : g_groups)
        if (g->P == P && std::equal(g->its.begin(), g->its.end(), its)) return g;
    return nullptr;
}
}  // namespace

static int group_check(mik_cgd **its, int P, const char *who)
{
    if (!its || P < 1) return MIK_ERR_INVALID;
    for (int p = 0; p < P; ++p) {
        if (!its[p]) return MIK_ERR_INVALID;
        if (its[p]->nranks != P || its[p]->rank != p) return mik_fail(its[p]->base.ctx, MIK_ERR_MISMATCH, "%s: handle %d is rank %d of %d", who, p, its[p]->rank, its[p]->nranks);
        if (its[p]->base.dtype != its[0]->base.dtype) return mik_fail(its[p]->base.ctx, MIK_ERR_MISMATCH, "%s: mixed dtypes", who);
    }
    return MIK_OK;
}

// a copy between two ranks of the group: same device -> device-to-device, else an explicit peer copy (works with and without peer access)
static inline hipError_t group_copy(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t s)
{
    return dst_dev == src_dev ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) : hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s);
}

static int group_get(mik_cgd **its, int P, GroupState **out)
{
    GroupState *g = find_group(its, P);
    if (!g) {
        g = new (std::nothrow) GroupState();
        if (!g) return MIK_ERR_NOMEM;
        g->P = P;
        g->its.assign(its, its + P);
        for (auto *v : {&g->ev.packed, &g->ev.halo, &g->ev.dot, &g->ev.rr}) v->assign((size_t)P, nullptr);
        g->ev.side.assign((size_t)P, nullptr);
        for (int p = 0; p < P; ++p) {
            mik_ctx *ctx = its[p]->base.ctx;
            MIK_HIP(ctx, hipSetDevice(ctx->device));
            for (int q = 0; q < P; ++q)          // peer access for copies between different devices (xGMI P2P)
                if (its[q]->base.ctx->device != ctx->device) {
                    // (not fatal: the copies below are hipMemcpyPeerAsync, which the runtime stages through the host where two devices
                    // cannot address each other -- the group is the transport of last resort and must come up on any topology)
                    hipError_t e = hipDeviceEnablePeerAccess(its[q]->base.ctx->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) g->peer_access = false;
                    (void)hipGetLastError();
                }
            for (auto *v : {&g->ev.packed, &g->ev.halo, &g->ev.dot, &g->ev.rr}) MIK_HIP(ctx, hipEventCreateWithFlags(&(*v)[p], hipEventDisableTiming));
            MIK_HIP(ctx, hipStreamCreateWithFlags(&g->ev.side[p], hipStreamNonBlocking));
        }
        g_groups.push_back(g);
    }
    *out = g;
    return MIK_OK;
}

// called by mik_cgd_destroy: a group must not outlive one of its handles
void mik_cgd_group_forget(mik_cgd *it)
{
    for (size_t i = 0; i < g_groups.size();) {
        GroupState *g = g_groups[i];
        if (std::find(g->its.begin(), g->its.end(), it) != g->its.end()) {
            for (int p = 0; p < g->P; ++p) (void)hipStreamSynchronize(g->ev.side[p]);
            g_groups.erase(g_groups.begin() + (long)i);
            delete g;
        } else ++i;
    }
}

extern "C" int mik_cgd_group_release(mik_cgd **its, int P)
{
    GroupState *g = its ? find_group(its, P) : nullptr;
    if (!g) return MIK_OK;
    for (int p = 0; p < P; ++p) { (void)hipSetDevice(its[p]->base.ctx->device); (void)hipStreamSynchronize(its[p]->base.ctx->stream); (void)hipStreamSynchronize(g->ev.side[p]); }
    g_groups.erase(std::find(g_groups.begin(), g_groups.end(), g));
    delete g;
    return MIK_OK;
}

// halo of every rank: rank p's ghost segments are copied from the owners' packed send buffers on p's side stream,
// after the owner's pack kernel (ev.packed[peer]) and after p's own previous SpMV (ev.packed[p])
static int group_halo_begin(GroupState *g, std::vector<char> &pending)
{
    const int P = g->P;
    pending.assign((size_t)P, 0);
    for (int p = 0; p < P; ++p) {
        mik_ctx *ctx = g->its[p]->base.ctx;
        MIK_HIP(ctx, hipSetDevice(ctx->device));
        MIK_HIP(ctx, hipEventRecord(g->ev.packed[p], ctx->stream));
    }
    for (int p = 0; p < P; ++p) {
        mik_cgd *it = g->its[p];
        if (it->recv.empty()) continue;
        mik_ctx *ctx = it->base.ctx;
        const size_t es = mik_dtype_size(it->base.dtype);
        MIK_HIP(ctx, hipSetDevice(ctx->device));
        MIK_HIP(ctx, hipStreamWaitEvent(g->ev.side[p], g->ev.packed[p], 0));
        unsigned char *ghost = (unsigned char *)it->u_ext + es * (size_t)it->base.n;
        for (const auto &sg : it->recv) {
            mik_cgd *src = g->its[sg.peer];
            const mik_cgd::HaloSeg *match = nullptr;
            for (const auto &ss : src->send) if (ss.peer == p) { match = &ss; break; }
            if (!match || match->cnt != sg.cnt)
                return mik_fail(ctx, MIK_ERR_MISMATCH, "halo plans disagree: rank %d expects %lld entries from rank %d", p, (long long)sg.cnt, sg.peer);
            MIK_HIP(ctx, hipStreamWaitEvent(g->ev.side[p], g->ev.packed[sg.peer], 0));
            MIK_HIP(ctx, group_copy(ghost + es * (size_t)sg.off, ctx->device, (const unsigned char *)src->send_buf + es * (size_t)match->off, src->base.ctx->device,
                                    es * (size_t)sg.cnt, g->ev.side[p]));
        }
        MIK_HIP(ctx, hipEventRecord(g->ev.halo[p], g->ev.side[p]));
        pending[(size_t)p] = 1;
    }
    return MIK_OK;
}

// slot [q] of rank q's array -> slot [q] of every other rank's array (`which`: 0 = dot_all, 1 = rr_all)
static int group_gather_scalar(GroupState *g, int which)
{
    const int P = g->P;
    if (P == 1) return MIK_OK;
    std::vector<hipEvent_t> &ev = which == 0 ? g->ev.dot : g->ev.rr;
    for (int p = 0; p < P; ++p) {
        mik_ctx *ctx = g->its[p]->base.ctx;
        MIK_HIP(ctx, hipSetDevice(ctx->device));
        MIK_HIP(ctx, hipEventRecord(ev[p], ctx->stream));
    }
    for (int p = 0; p < P; ++p) {
        mik_cgd *it = g->its[p];
        mik_ctx *ctx = it->base.ctx;
        const size_t es = mik_dtype_size(it->base.dtype);
        MIK_HIP(ctx, hipSetDevice(ctx->device));
        unsigned char *mine = (unsigned char *)(which == 0 ? it->dot_all : it->rr_all);
        for (int q = 0; q < P; ++q) {
            if (q == p) continue;
            const unsigned char *theirs = (const unsigned char *)(which == 0 ? g->its[q]->dot_all : g->its[q]->rr_all);
            MIK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev[q], 0));
            MIK_HIP(ctx, group_copy(mine + es * (size_t)q, ctx->device, theirs + es * (size_t)q, g->its[q]->base.ctx->device, es, ctx->stream));
        }
    }
    return MIK_OK;
}

static int group_all(GroupState *g, int phase, int64_t iteration)
{
    for (int p = 0; p < g->P; ++p) {
        MIK_HIP(g->its[p]->base.ctx, hipSetDevice(g->its[p]->base.ctx->device));
        MIK_TRY(mik_cgd_phase(g->its[p], phase, iteration));
    }
    return MIK_OK;
}

static int group_halo_end(GroupState *g, const std::vector<char> &pending, int p)
{
    if (pending[(size_t)p]) MIK_HIP(g->its[p]->base.ctx, hipStreamWaitEvent(g->its[p]->base.ctx->stream, g->ev.halo[p], 0));
    return MIK_OK;
}

static int group_wait_raw(GroupState *g, std::vector<CgMirror> &ms)
{
    for (int p = 0; p < g->P; ++p) {
        MIK_HIP(g->its[p]->base.ctx, hipSetDevice(g->its[p]->base.ctx->device));
        MIK_TRY(cgd_wait_raw(g->its[p], &ms[(size_t)p]));
        if (ms[(size_t)p].range != ms[0].range)
            return mik_fail(g->its[p]->base.ctx, MIK_ERR_MISMATCH, "row-partitioned cg: rank %d disagrees with rank 0 on the range of |r|^2", p);
    }
    return MIK_OK;
}

// the scaled norm of r over the group's ranks (cgd_norm_stage*): every rank repeats the host arithmetic on identical slots
static int group_scaled_norm(GroupState *g)
{
    const int P = g->P;
    MIK_TRY(group_all(g, 20, 0));
    MIK_TRY(group_gather_scalar(g, 1));
    bool direct = false;
    for (int p = 0; p < P; ++p) {
        bool dp = false;
        MIK_HIP(g->its[p]->base.ctx, hipSetDevice(g->its[p]->base.ctx->device));
        MIK_TRY(cgd_norm_stage1_any(g->its[p], &dp));
        if (p == 0) direct = dp;
        else if (dp != direct) return mik_fail(g->its[p]->base.ctx, MIK_ERR_MISMATCH, "row-partitioned cg: rank %d disagrees with rank 0 on max |r|", p);
    }
    if (direct) return MIK_OK;
    MIK_TRY(group_gather_scalar(g, 1));
    for (int p = 0; p < P; ++p) {
        MIK_HIP(g->its[p]->base.ctx, hipSetDevice(g->its[p]->base.ctx->device));
        MIK_TRY(cgd_norm_stage2_any(g->its[p]));
    }
    return MIK_OK;
}

extern "C" int mik_cgd_group_init(mik_cgd **its, int P, double *residual, double *tol)
{
    MIK_TRY(group_check(its, P, "mik_cgd_group_init"));
    GroupState *g = nullptr;
    MIK_TRY(group_get(its, P, &g));
    std::vector<char> pending;
    MIK_TRY(group_all(g, 10, 0));
    if (!its[0]->initially_zero) {
        MIK_TRY(group_halo_begin(g, pending));
        for (int p = 0; p < P; ++p) MIK_TRY(group_halo_end(g, pending, p));
    }
    MIK_TRY(group_all(g, 11, 0));
    MIK_TRY(group_gather_scalar(g, 1));
    MIK_TRY(group_all(g, 12, 0));
    std::vector<CgMirror> ms((size_t)P);
    MIK_TRY(group_wait_raw(g, ms));
    if (ms[0].range) {
        MIK_TRY(group_scaled_norm(g));
        MIK_TRY(group_all(g, 24, 0));
        MIK_TRY(group_wait_raw(g, ms));
    }
    for (int p = 0; p < P; ++p) {
        double res = 0, tl = 0;
        int done = 0;
        int64_t steps = 0;
        MIK_HIP(its[p]->base.ctx, hipSetDevice(its[p]->base.ctx->device));
        MIK_TRY(cgd_collect(its[p], ms[(size_t)p], &res, &tl, &done, nullptr, 0, &steps));
        its[p]->initialised = true;
        if (p == 0) { if (residual) *residual = res; if (tol) *tol = tl; }
        else if (res != its[0]->base.residual || tl != its[0]->base.tol)
            return mik_fail(its[p]->base.ctx, MIK_ERR_MISMATCH, "mik_cgd_group_init: rank %d disagrees with rank 0 on the initial residual", p);
    }
    return MIK_OK;
}

extern "C" int mik_cgd_group_iterate_many(mik_cgd **its, int P, int64_t iteration, int64_t max_steps, double *residuals, int64_t *steps_done)
{
    MIK_TRY(group_check(its, P, "mik_cgd_group_iterate_many"));
    if (!steps_done || iteration < 0) return MIK_ERR_INVALID;
    *steps_done = 0;
    GroupState *g = nullptr;
    MIK_TRY(group_get(its, P, &g));
    mik_cg &b0 = its[0]->base;
    for (int p = 0; p < P; ++p) if (!its[p]->initialised) return mik_fail(b0.ctx, MIK_ERR_INVALID, "mik_cgd_group_iterate_many: call mik_cgd_group_init first");
    if (max_steps <= 0 || iteration >= b0.maxiter || b0.residual <= b0.tol) return MIK_OK;
    max_steps = std::min<int64_t>(std::min<int64_t>(max_steps, b0.maxiter - iteration), b0.hist_cap);
    bool split = true, early = true;
    for (int p = 0; p < P; ++p) {
        split = split && its[p]->int_end > its[p]->int_begin;
        early = early && its[p]->n_early > 0;
        its[p]->base.fuse_x = (its[p]->base.ctx->tuning[MIK_KNOB_CG_STEP] & 1) == 0;                // as mik_cgd_iterate_many: x .+= alpha .* u rides on the next sweep over u
    }
    early = early && split;
    std::vector<char> pending;
    std::vector<CgMirror> ms((size_t)P);
    for (int64_t j0 = 0;;) {
        for (int64_t j = j0; j < max_steps; ++j) {
            const int64_t itn = iteration + j;
            if (early) {                                                // the step structure of cgd_enqueue_head, with peer copies for RCCL
                for (int p = 0; p < P; ++p) {
                    MIK_HIP(its[p]->base.ctx, hipSetDevice(its[p]->base.ctx->device));
                    MIK_TRY(mik_cgd_phase(its[p], its[p]->early_merged ? 9 : 7, itn));
                }
                MIK_TRY(group_halo_begin(g, pending));
                MIK_TRY(group_all(g, 8, itn));
            } else {
                MIK_TRY(group_all(g, 0, itn));
                MIK_TRY(group_halo_begin(g, pending));
            }
            if (split) {
                MIK_TRY(group_all(g, 4, itn));
                for (int p = 0; p < P; ++p) MIK_TRY(group_halo_end(g, pending, p));
                MIK_TRY(group_all(g, 5, itn));
            } else {
                for (int p = 0; p < P; ++p) MIK_TRY(group_halo_end(g, pending, p));
                MIK_TRY(group_all(g, 1, itn));
            }
            MIK_TRY(group_gather_scalar(g, 0));
            MIK_TRY(group_all(g, 2, itn));
            MIK_TRY(group_gather_scalar(g, 1));
            MIK_TRY(group_all(g, 3, itn));
        }
        if (its[0]->base.fuse_x) MIK_TRY(group_all(g, 6, iteration + max_steps - 1));     // nothing runs ahead here: apply the last x update now
        MIK_TRY(group_wait_raw(g, ms));
        if (!ms[0].range) break;
        // a frozen step (|r|^2 outside the safe range on every rank alike): scaled norm over the ranks, close the step, go on
        MIK_TRY(group_scaled_norm(g));
        for (int p = 0; p < P; ++p) {
            its[p]->norm_fix_index = (int)ms[(size_t)p].nhist;
            its[p]->norm_it_next = iteration + ms[(size_t)p].nhist + 1;
        }
        MIK_TRY(group_all(g, 23, 0));
        MIK_TRY(group_wait_raw(g, ms));
        j0 = ms[0].nhist;
        if (ms[0].done || j0 >= max_steps) break;
    }
    std::vector<double> h0((size_t)max_steps), hp((size_t)max_steps);
    int64_t n0 = 0;
    for (int p = 0; p < P; ++p) {
        int done = 0;
        int64_t steps = 0;
        MIK_HIP(its[p]->base.ctx, hipSetDevice(its[p]->base.ctx->device));
        MIK_TRY(cgd_collect(its[p], ms[(size_t)p], nullptr, nullptr, &done, p == 0 ? h0.data() : hp.data(), max_steps, &steps));
        if (p == 0) n0 = steps;
        else if (steps != n0 || !std::equal(h0.begin(), h0.begin() + n0, hp.begin()))
            return mik_fail(its[p]->base.ctx, MIK_ERR_MISMATCH, "mik_cgd_group_iterate_many: rank %d disagrees with rank 0 on the residual history", p);
    }
    if (residuals) std::copy(h0.begin(), h0.begin() + n0, residuals);
    *steps_done = n0;
    return MIK_OK;
}
