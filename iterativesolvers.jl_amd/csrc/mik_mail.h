// mik_mail.h -- device side of the peer-mapped mailbox (include/mik.h "Transport 3"; host side and protocol: csrc/mik_comm.hip).
// Shared by mik_comm.hip (the row-partitioned CG step, the links) and mik_krylov.hip (the single-launch Gram-Schmidt of a row-partitioned GMRES,
// whose grid-wide sums are exchanged between the ranks INSIDE the launch).
#pragma once
#include "mik_internal.h"

constexpr int MIK_MAIL_MAXP = 64;
constexpr int MIK_MAIL_KINDS = 3;        // 0: dot(u, c)   1: |r|^2   2: every other gather (initial residual, the scaled-norm stages)
// a scalar in flight: two 8-byte words, each {low 32 bits of the sequence number, half of the value's bits} -- every word is ONE atomic
// store, so the two need no ordering between them (no release fence, i.e. no L2 write-back, in the finalising kernels): the reader
// takes the value once BOTH words carry the sequence number it waits for
struct MailSlot { unsigned long long w0, w1; };
constexpr int MIK_MAIL_VEC = 64;         // scalars of one vector exchange (the k projections of a CGS / DGKS column of the row-partitioned GMRES)
struct MailBox {
    MailSlot slot[MIK_MAIL_KINDS][2][MIK_MAIL_MAXP];    // [kind][seq & 1][sender]
    unsigned long long halo_seq[MIK_MAIL_MAXP];          // [sender]: its halo of exchange no. halo_seq[sender] has landed in this rank's landing buffer
    unsigned long long packed_seq;                       // this rank: the send buffer of exchange no. packed_seq is packed (the side stream waits for it)
    MailSlot vec[2][MIK_MAIL_MAXP][MIK_MAIL_VEC];        // [seq & 1][sender][j]: element j of a vector in flight (lane j of the one-wave exchange serves it)
};


#ifdef __HIPCC__
template <typename T> static __device__ __forceinline__ unsigned long long mail_bits(T v)
{
    if (sizeof(T) == 8) { double d = (double)v; return __builtin_bit_cast(unsigned long long, d); }
    float f = (float)v;
    return (unsigned long long)__builtin_bit_cast(unsigned, f);
}
template <typename T> static __device__ __forceinline__ T mail_value(unsigned long long b)
{
    if (sizeof(T) == 8) return (T)__builtin_bit_cast(double, b);
    return (T)__builtin_bit_cast(float, (unsigned)b);
}

// wait until *p >= want; false after `ticks` of the wall clock.  Relaxed system-scope loads: what the flag guards is read by the NEXT
// kernel on the stream, whose start is the acquire (an acquire per poll would invalidate this XCD's L2 again and again)
static __device__ __forceinline__ bool mail_wait(const unsigned long long *p, unsigned long long want, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= want) return true;
        if ((spins & 255u) == 255u && wall_clock64() - t0 > ticks) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// value -> slot [kind][seq & 1][rank] of every peer; then the P slots of this rank's own mailbox -> all[0 .. P): lane q serves peer q
template <typename T>
static __device__ __forceinline__ T mail_exchange(MailBox *const *__restrict__ peers, int P, int rank, int kind, unsigned long long seq, T mine,
                                           T *__restrict__ all, unsigned long long ticks, unsigned *__restrict__ err)
{
    const int q = threadIdx.x & 63;
    if (q >= P) return T(0);
    const unsigned long long tag = (seq & 0xFFFFFFFFull) << 32, b = mail_bits<T>(mine);
    MailSlot *dst = &peers[q]->slot[kind][seq & 1ull][rank];
    __hip_atomic_store(&dst->w0, tag | (b & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&dst->w1, tag | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const MailSlot *src = &peers[rank]->slot[kind][seq & 1ull][q];
    unsigned long long a0 = 0, a1 = 0;
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
        a0 = __hip_atomic_load(&src->w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        a1 = __hip_atomic_load(&src->w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((a0 >> 32 << 32) == tag && (a1 >> 32 << 32) == tag) break;
        if ((spins & 255u) == 255u && wall_clock64() - t0 > ticks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    const T v = mail_value<T>((a0 & 0xFFFFFFFFull) | (a1 << 32));
    all[q] = v;
    return v;
}

// mail_exchange in two halves: post this rank's value to ONE peer; collect ONE rank's value from this rank's own mailbox
template <typename T> static __device__ __forceinline__ void mail_post(MailBox *peer, int kind, unsigned long long seq, int rank, T mine)
{
    const unsigned long long tag = (seq & 0xFFFFFFFFull) << 32, b = mail_bits<T>(mine);
    MailSlot *dst = &peer->slot[kind][seq & 1ull][rank];
    __hip_atomic_store(&dst->w0, tag | (b & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&dst->w1, tag | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <typename T> static __device__ __forceinline__ T mail_collect(const MailBox *mine, int kind, unsigned long long seq, int q, unsigned long long ticks, unsigned *__restrict__ err)
{
    const unsigned long long tag = (seq & 0xFFFFFFFFull) << 32;
    const MailSlot *src = &mine->slot[kind][seq & 1ull][q];
    unsigned long long a0 = 0, a1 = 0;
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
        a0 = __hip_atomic_load(&src->w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        a1 = __hip_atomic_load(&src->w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((a0 >> 32 << 32) == tag && (a1 >> 32 << 32) == tag) break;
        if ((spins & 255u) == 255u && wall_clock64() - t0 > ticks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    return mail_value<T>((a0 & 0xFFFFFFFFull) | (a1 << 32));
}

// the P values of a wave (lane q: rank q's) added in rank order, in every lane
template <typename T> static __device__ __forceinline__ T mail_rank_sum(T v, int P)
{
    T s = __shfl(v, 0);
    for (int p = 1; p < P; ++p) s = s + __shfl(v, p);
    return s;
}

// The exchange functor of k_map_pro (csrc/mik_kernels.h) for a row partition: EVERY workgroup of the sweep turns the local total it has just
// finalised into the sum over the ranks -- workgroup 0 posts the total to every peer, the first wave of every workgroup collects the P totals
// from this rank's own mailbox (they stay there until the exchange after next: a peer posts that one only after it has seen this rank's NEXT
// total, which this rank posts from its NEXT launch, i.e. after every workgroup of this launch has finished) and adds them in rank order.
// One launch per Gram-Schmidt pass instead of sweep + finalise-and-exchange.
struct MailSum {
    MailBox *const *peers; int P, rank; unsigned long long seq, ticks; unsigned *err;
    template <typename T> __device__ __forceinline__ T operator()(T cf) const
    {
        __shared__ T glob;
        if (threadIdx.x < 64) {
            const int q = threadIdx.x;
            T v = T(0);
            if (q < P) {
                if (blockIdx.x == 0) mail_post<T>(peers[q], 2, seq, rank, cf);
                v = mail_collect<T>(peers[rank], 2, seq, q, ticks, err);
            }
            const T sum = mail_rank_sum(v, P);
            if (threadIdx.x == 0) glob = sum;
        }
        __syncthreads();
        return glob;
    }
};


// lane j's value -> element j of slot vec[seq & 1][rank] of every peer; then, per lane, the P values of element j added in rank order
template <typename T>
static __device__ __forceinline__ T mail_exchange_vec(MailBox *const *__restrict__ peers, int P, int rank, unsigned long long seq, T mine, int count,
                                               unsigned long long ticks, unsigned *__restrict__ err)
{
    const int j = threadIdx.x & 63;
    if (j >= count) return T(0);
    const unsigned long long tag = (seq & 0xFFFFFFFFull) << 32, b = mail_bits<T>(mine);
    for (int q = 0; q < P; ++q) {
        MailSlot *dst = &peers[q]->vec[seq & 1ull][rank][j];
        __hip_atomic_store(&dst->w0, tag | (b & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&dst->w1, tag | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    T sum = T(0);
    const unsigned long long t0 = wall_clock64();
    for (int q = 0; q < P; ++q) {
        const MailSlot *src = &peers[rank]->vec[seq & 1ull][q][j];
        unsigned long long a0 = 0, a1 = 0;
        for (unsigned spins = 0;; ++spins) {
            a0 = __hip_atomic_load(&src->w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            a1 = __hip_atomic_load(&src->w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((a0 >> 32 << 32) == tag && (a1 >> 32 << 32) == tag) break;
            if ((spins & 255u) == 255u && wall_clock64() - t0 > ticks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        const T v = mail_value<T>((a0 & 0xFFFFFFFFull) | (a1 << 32));
        sum = q == 0 ? v : sum + v;                       // rank order: ((v_0 + v_1) + v_2) + ...
    }
    return sum;
}


// The exchange functor of the SINGLE-LAUNCH Gram-Schmidt (k_mgs_fused, csrc/mik_kernels.h) over a row partition: pass `pass` of launch `tag` uses
// element `pass` of the vector slots vec[tag & 1][sender] -- one slot per pass, because no kernel boundary separates the passes of a launch (a fast
// rank may post pass i + 2 while a slow workgroup of a peer is still collecting pass i); two launches alternate between the parities, and launch
// tag + 2 cannot start anywhere before every rank has finished launch tag (it needs all of launch tag + 1's totals).  Workgroup 0 posts this rank's
// total to every peer; the first wave of EVERY workgroup collects the P totals from this rank's mailbox and adds them in rank order.
struct MailSumPass {
    MailBox *const *peers; int P, rank; unsigned long long tag, ticks; unsigned *err;
    template <typename T> __device__ __forceinline__ T pass(T cf, int pass_no, bool poster) const
    {
        __shared__ T glob;
        if (threadIdx.x < 64) {
            const int q = threadIdx.x;
            const unsigned long long tg = (tag & 0xFFFFFFFFull) << 32;
            T v = T(0);
            if (q < P) {
                if (poster) {
                    const unsigned long long b = mail_bits<T>(cf);
                    MailSlot *dst = &peers[q]->vec[tag & 1ull][rank][pass_no];
                    __hip_atomic_store(&dst->w0, tg | (b & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(&dst->w1, tg | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                const MailSlot *src = &peers[rank]->vec[tag & 1ull][q][pass_no];
                unsigned long long a0 = 0, a1 = 0;
                const unsigned long long t0 = wall_clock64();
                for (unsigned spins = 0;; ++spins) {
                    a0 = __hip_atomic_load(&src->w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    a1 = __hip_atomic_load(&src->w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((a0 >> 32 << 32) == tg && (a1 >> 32 << 32) == tg) break;
                    if ((spins & 255u) == 255u && wall_clock64() - t0 > ticks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                v = mail_value<T>((a0 & 0xFFFFFFFFull) | (a1 << 32));
            }
            const T sum = mail_rank_sum(v, P);
            if (threadIdx.x == 0) glob = sum;
        }
        __syncthreads();
        const T out = glob;
        __syncthreads();                   // (glob is reused by the next pass)
        return out;
    }
};
#endif  // __HIPCC__

// what a kernel outside mik_comm.hip needs of a link's communicator (csrc/mik_comm.hip: plink_mail)
struct PlinkMail { MailBox *const *peers; int P, rank; unsigned long long ticks; unsigned *err; };
struct mik_plink;
PlinkMail plink_mail(const mik_plink *pl);
unsigned long long plink_next_vec_tag(mik_plink *pl);       // the tag of the next vector / per-pass exchange on the link's communicator
