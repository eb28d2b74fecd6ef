// mik_mgs_res.h -- orthogonalize_and_normalize!(V, w, h, ModifiedGramSchmidt()) (src/orthogonalize.jl:67-79) as ONE launch at sizes
// where the Krylov basis lives in HBM: the "resident w" form.
//
// The multi-launch chain (k_map<OpMgsPass>) moves 4 n words per pass: it reads w, v_i, v_{i+1} and writes w, because the coefficient
// of pass i + 1 needs the w of pass i completed everywhere.  Here one workgroup per compute unit stays on the device for the whole
// column and KEEPS ITS PART OF w ON THE CHIP between the passes: the first RR rounds of its segments in registers, the next RL rounds
// in LDS (128 KB), only what does not fit is streamed.  A pass then reads v_i and v_{i+1} (2 n words) plus 2 (1 - f) n for the
// non-resident share f of w: 2.25 n at 256^3 fp64 on 256 CUs (f = 0.875) instead of 4 n.  The hand-off between passes is the slot
// mechanism of k_mgs_fused (csrc/mik_kernels.h): every workgroup publishes the sums of its segments, every workgroup evaluates the
// fixed level-2 tree over ALL segment sums itself.
//
// Same arithmetic, same bits: a reduction segment is MIK_BLOCK * W * L consecutive elements summed by 256 virtual threads exactly as
// everywhere else (thread t: its W elements of load 0, then of load 1; wave tree; the 4 wave sums left to right); NT / 256 segments
// are in flight per round, each on its own group of four waves.  Level 2: the 1024-virtual-thread shape of level2_sum.
#pragma once
#include <type_traits>

#include "mik_kernels.h"

#ifndef MIK_MGS_RES_RR
#define MIK_MGS_RES_RR 17        // rounds of w a thread keeps in registers (4 doubles / 8 floats each = 8 registers): 136 of the 256 registers of a thread of a 512-thread workgroup (18 and more spill)
#endif
#ifndef MIK_MGS_RES_DEPTH
#define MIK_MGS_RES_DEPTH 2      // rounds of the column streams in flight ahead of the arithmetic (register rounds)
#endif
// The form pays while at least 0.6 of w stays on the chip: measured per inner iteration of gmres!(30), chain -> resident, at a resident share of
// 0.81 (256^3) 1,620 -> 1,215 us, 0.67 (272^3) 1,717 -> 1,482, 0.57 (288^3) 2,055 -> 2,015, 0.50 (300^3) 2,282 -> 2,399, 0.41 (320^3) 2,855 -> 3,068
// (scripts/micro/mgs_resident_big.py): the streamed rounds run at less memory-level parallelism than the chain's sweeps.  Beyond: the chain.
#define MIK_MGS_RES_MAX_S(rr, rl) ((((rr) + (rl)) * 2 * 5) / 3)      /* segments per workgroup (2 per round) with (rr + rl) / rounds >= 0.6 */
#ifndef MIK_MGS_RES_RL
#define MIK_MGS_RES_RL 9         // rounds in LDS: 9 x 2 segments x 8 KB = 144 KB of the 160 KB
#endif

// level 2 over `ns` slots by a workgroup of NT threads (NT / 64 waves): real thread (wave w, lane l) plays the virtual threads
// (w + NW j) * 64 + l; a virtual thread adds its slots vt, vt + 1024, ... in ascending order from +0; wave tree per 64; the 16 wave
// sums left to right -- level2_sum's shape, whatever ns.  Slots are fetched in batches of 8 independent loads and re-polled until
// none shows the "not yet written" pattern (bounded).
template <typename T, int NT>
__device__ __forceinline__ T mgs_grid_sum_wide(const T *__restrict__ slots, int ns, T *lds16, int *err)
{
    using U = typename MgsBits<T>::U;
    constexpr int NW = NT / 64, J = 16 / NW;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const U *sp = reinterpret_cast<const U *>(slots);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int vt = (w + NW * j) * 64 + lane;
        T v = T(0);
        for (int q0 = vt; q0 < ns; q0 += 8 * MIK_FIN_THREADS) {
            U bits[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int q = q0 + c * MIK_FIN_THREADS;
                bits[c] = q < ns ? __hip_atomic_load(sp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : U(0);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int q = q0 + c * MIK_FIN_THREADS;
                if (q < ns) {
                    // bounded, and at most ONE wait runs to its bound: once any thread of the workgroup has given up (a workgroup of the launch never got
                    // its compute unit: GPU shared with other work) nobody spins any more -- the launch drains in well under a second with err set, and
                    // the host redoes the column with the chain
                    for (int spin = 0; bits[c] == MgsBits<T>::EMPTY && spin < (1 << 18) && !*(volatile int *)err; ++spin)
                        bits[c] = __hip_atomic_load(sp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (bits[c] == MgsBits<T>::EMPTY) *err = 1;      // timed out: never hang the device
                    T val;
                    __builtin_memcpy(&val, &bits[c], sizeof(T));
                    v = v + val;
                }
            }
        }
        v = wave_tree(v);
        if (lane == 0) lds16[w + NW * j] = v;
    }
    __syncthreads();
    T tot = lds16[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) tot = tot + lds16[q];
    __syncthreads();
    return tot;
}

template <int I, int N, typename F> __device__ __forceinline__ void mgs_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        mgs_static_for<I + 1, N>(f);
    }
}

// NT threads = NT / 256 segments per round; S segments per workgroup (consecutive), rounds = ceil(S / SPR); rounds [0, RR) of w in
// registers, [RR, RR + RL) in LDS, the rest streamed.  P: slot rows [2][kmax + 1][stride], stride >= nseg.
template <typename T, int NT, int RR, int RL>
__device__ __forceinline__ void mgs_resident_body(int64_t n, int k, const T *__restrict__ V, int64_t ldv, T *__restrict__ w, T *__restrict__ P, int kmax,
                                                  int stride, int nseg, int S, int parity, MgsMirror *mirror, unsigned long long seq,
                                                  typename VT<T>::vec *wl, T *lds16, T (*segw)[4], int *s_err_p)
{
    using U = typename MgsBits<T>::U;
    using vec = typename VT<T>::vec;
    constexpr int W = VT<T>::W, L = MIK_RED_L, SPR = NT / MIK_BLOCK;
    constexpr int64_t SEG = (int64_t)MIK_BLOCK * W * L;
    static_assert(NT % MIK_BLOCK == 0 && 16 % (NT / 64) == 0, "whole segments per round, whole virtual waves per real wave");
    int &s_err = *s_err_p;
    // (wave-uniform values are told to the compiler as such: every element address is then a scalar base + ONE per-thread offset)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), sub = wave >> 2, tt = t & (MIK_BLOCK - 1), wv = wave & 3, s = (int)blockIdx.x;
    const int rounds = (S + SPR - 1) / SPR;
    if (t == 0) s_err = 0;
    T *cur = P + (size_t)parity * (size_t)(kmax + 1) * stride;
    T *oth = P + (size_t)(parity ^ 1) * (size_t)(kmax + 1) * stride;
    for (int q = t; q < (kmax + 1) * S; q += NT) {                    // re-arm this workgroup's slots of the other buffer
        const int g = s * S + q % S;
        if (g < nseg) __hip_atomic_store(reinterpret_cast<U *>(oth + (size_t)(q / S) * stride) + g, MgsBits<T>::EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // element index of (round r, load l) for this thread; a segment beyond this workgroup's range or the vector's end contributes nothing
    auto seg_of = [&](int r) { return r * SPR + sub; };
    // Addressing: every stream (w, a column) is read through a buffer descriptor over THIS workgroup's S segments of it; an access is
    // descriptor + one per-thread byte offset (the same vector register for every access of the kernel) + a scalar byte offset for (round, load).
    // Bytes beyond the vector's end or beyond the workgroup's range are outside the descriptor: loads return 0, stores are dropped
    // (16-byte accesses are checked dword by dword, so an odd n loses nothing and writes nothing beyond its last element).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int64_t chunk0 = (int64_t)s * S * SEG;                      // first element of this workgroup
    const int64_t left = n - chunk0;
    const int nbytes = (int)((left < (int64_t)S * SEG ? (left > 0 ? left : 0) : (int64_t)S * SEG) * (int64_t)sizeof(T));
    auto rsrc = [&](const T *p) { return __builtin_amdgcn_make_buffer_rsrc((void *)(p + chunk0), (short)0, nbytes, (int)0x00020000); };
    const unsigned tbyte = (unsigned)(W * tt) * (unsigned)sizeof(T);
    auto boff = [&](int r, int l) -> int { return (int)(((int64_t)seg_of(r) * SEG + (int64_t)l * MIK_BLOCK * W) * (int64_t)sizeof(T)); };
    // element index of the first of this thread's W elements of (round r, load l), for the bounds tests of the arithmetic
    auto idx = [&](int r, int l) -> int64_t {
        const int q = seg_of(r);
        return (q < S) ? chunk0 + (int64_t)q * SEG + (int64_t)l * MIK_BLOCK * W + (int64_t)(W * tt) : (int64_t)n;
    };
    auto ld = [&](__amdgpu_buffer_rsrc_t rs, int r, int l, T(&d)[W]) {
        // (the hardware's range check covers the vector offset, not the scalar one: the whole offset travels as vector offset)
        vec v = __builtin_bit_cast(vec, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)tbyte + boff(r, l), 0, 0));
#pragma unroll
        for (int e = 0; e < W; ++e) d[e] = el<T>(v, e);
    };
    auto st = [&](__amdgpu_buffer_rsrc_t rs, int r, int l, const T(&d)[W]) {
        vec v;
#pragma unroll
        for (int e = 0; e < W; ++e) el<T>(v, e) = d[e];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)tbyte + boff(r, l), 0, 0);
    };
    auto lds_at = [&](int r, int l) -> vec & { return wl[(((r - RR) * SPR + sub) * L + l) * MIK_BLOCK + tt]; };
    auto seg_done = [&](int r, T acc) {                               // wave tree of a segment's 256 partial sums; its 4 wave sums are added by publish()
        const T ws = wave_tree(acc);
        const int q = seg_of(r);
        if (lane == 0 && q < S) segw[q][wv] = ws;
    };
    auto publish = [&](int pass) {
        __syncthreads();
        if (t < S && s * S + t < nseg) {
            T tot = segw[t][0];
            tot = tot + segw[t][1]; tot = tot + segw[t][2]; tot = tot + segw[t][3];
            __hip_atomic_store(reinterpret_cast<U *>(cur + (size_t)pass * stride) + s * S + t, mgs_slot_bits<T>(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    };

    T wr[RR > 0 ? RR : 1][L][W];
    // One sweep over this workgroup's segments.  first: w comes on chip (a = w) and nothing is subtracted; otherwise a = v_i and
    // w .-= h .* a (src/orthogonalize.jl:72).  Then the products for the next coefficient: with b = v_{i+1} (:71), or w .* w when there is no b (:75).
    // Per 16-byte group: all W elements updated, then their products added in element order (the order of OpMgsPass::compute_vec and of k_mgs_fused).
    // The register rounds run as a software pipeline D rounds deep (staging registers sa / sb; sched_barrier keeps the compiler from hoisting every load
    // of the unrolled sweep to the top, which would spill): 2 D L 16-byte loads per thread in flight.
    constexpr int D = MIK_MGS_RES_DEPTH;
    const __amdgpu_buffer_rsrc_t wrs = rsrc(w);
    // geth() delivers the coefficient of this pass (the grid-wide sum of the previous one) AFTER the first column loads have been issued: they do
    // not depend on it, and travel while the workgroups hand their segment sums to each other.
    auto sweep = [&](auto First, auto HasB, const T *__restrict__ pa, const T *__restrict__ pb, auto geth) {
        constexpr bool first = decltype(First)::value, hasb = decltype(HasB)::value;       // (compile-time: no selects between the two sources in the sweep)
        const __amdgpu_buffer_rsrc_t ars = rsrc(pa), brs = rsrc(hasb ? pb : pa);
        T sa[D][L][W], sb[D][L][W];
        auto issue = [&](auto Rc) {
            constexpr int r = decltype(Rc)::value, slot = r % D;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                if (first) ld(ars, r, l, wr[r][l]);             // w lands where it stays
                else ld(ars, r, l, sa[slot][l]);
                if (hasb) ld(brs, r, l, sb[slot][l]);
            }
        };
        mgs_static_for<0, (D < RR ? D : RR)>(issue);
        const T h = geth();
        mgs_static_for<0, RR>([&](auto Rc) {
            constexpr int r = decltype(Rc)::value, slot = r % D;
            T acc = T(0);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int64_t i = idx(r, l);
                if (!first) {
#pragma unroll
                    for (int e = 0; e < W; ++e)
                        if (i + e < n) { T tq = h * sa[slot][l][e]; wr[r][l][e] = wr[r][l][e] - tq; }
                }
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i + e < n) { T p = (hasb ? sb[slot][l][e] : wr[r][l][e]) * wr[r][l][e]; acc = acc + p; }
            }
            seg_done(r, acc);
            if constexpr (r + D < RR) issue(std::integral_constant<int, r + D>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        // the rounds that live in LDS or are streamed: the column data of round r + 1 is requested before round r is worked on
        T na[L][W], nb[L][W];
        auto fetch = [&](int r) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                ld(ars, r, l, na[l]);
                if (hasb) ld(brs, r, l, nb[l]);
            }
        };
        if (RR < rounds) fetch(RR);
#pragma unroll 1
        for (int r = RR; r < rounds; ++r) {
            const bool in_lds = r < RR + RL;
            T ca[L][W], cb[L][W];
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int e = 0; e < W; ++e) { ca[l][e] = na[l][e]; if (hasb) cb[l][e] = nb[l][e]; }
            if (r + 1 < rounds) fetch(r + 1);
            T acc = T(0);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int64_t i = idx(r, l);
                T ww[W];
                if (first) {
#pragma unroll
                    for (int e = 0; e < W; ++e) ww[e] = ca[l][e];
                } else {
                    if (in_lds) {
                        vec v = lds_at(r, l);
#pragma unroll
                        for (int e = 0; e < W; ++e) ww[e] = el<T>(v, e);
                    } else ld(wrs, r, l, ww);
#pragma unroll
                    for (int e = 0; e < W; ++e)
                        if (i + e < n) { T tq = h * ca[l][e]; ww[e] = ww[e] - tq; }
                }
#pragma unroll
                for (int e = 0; e < W; ++e)
                    if (i + e < n) { T p = (hasb ? cb[l][e] : ww[e]) * ww[e]; acc = acc + p; }
                if (in_lds) {
                    vec v;
#pragma unroll
                    for (int e = 0; e < W; ++e) el<T>(v, e) = ww[e];
                    lds_at(r, l) = v;
                } else if (!first) st(wrs, r, l, ww);
            }
            seg_done(r, acc);
        }
    };
    // ---- pass "0": dot(v_1, w) (src/orthogonalize.jl:71, i = 1) or, without columns, norm(w)^2 ------------------------------------------
    using Yes = std::true_type;
    using No = std::false_type;
    T *hout = reinterpret_cast<T *>(mirror + 1);
    // the tail: norm, scale, every element (back) to memory                                              src/orthogonalize.jl:75-76
    auto finish = [&]() {
        const T ss = mgs_grid_sum_wide<T, NT>(cur + (size_t)k * stride, nseg, lds16, &s_err);
        T nrm = mik_sqrt(ss);
        const bool ok = mik_nrm_in_range(ss);          // outside the safe range: leave w unscaled, the host rescales
        const T inv = ok ? T(1) / nrm : T(1);
        if (!ok) nrm = __builtin_nan("");
        mgs_static_for<0, RR>([&](auto Rc) {
            constexpr int r = decltype(Rc)::value;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                T o[W];
#pragma unroll
                for (int e = 0; e < W; ++e) o[e] = wr[r][l][e] * inv;
                st(wrs, r, l, o);
            }
        });
#pragma unroll 1
        for (int r = RR; r < rounds; ++r)
#pragma unroll
            for (int l = 0; l < L; ++l) {
                T o[W];
                if (r < RR + RL) {
                    vec v = lds_at(r, l);
#pragma unroll
                    for (int e = 0; e < W; ++e) o[e] = el<T>(v, e) * inv;
                } else {
                    ld(wrs, r, l, o);
#pragma unroll
                    for (int e = 0; e < W; ++e) o[e] = o[e] * inv;
                }
                st(wrs, r, l, o);
            }
        if (t == 0 && s_err) __hip_atomic_store(&mirror->err, s_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (s == 0 && t == 0) {
            hout[k] = nrm;
            __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    // Straight-line control flow around the register-resident part of w (no branch whose two sides both rewrite it: the register allocator
    // would otherwise hold two copies at the merge): a column-free call; else first sweep, k - 1 middle passes in ONE loop body, the last pass.
    if (k == 0) {
        sweep(Yes{}, No{}, w, (const T *)nullptr, [] { return T(0); });        // norm(w)^2
        publish(0);
        finish();
        return;
    }
    sweep(Yes{}, Yes{}, w, V, [] { return T(0); });             // dot(v_1, w)                                    :71 (i = 1)
    publish(0);
#pragma unroll 1
    for (int c = 0; c + 1 < k; ++c) {
        sweep(No{}, Yes{}, V + (int64_t)c * ldv, V + (int64_t)(c + 1) * ldv, [&] {     // w .-= h v_i; dot(v_{i+1}, w)      :72, :71
            const T h = mgs_grid_sum_wide<T, NT>(cur + (size_t)c * stride, nseg, lds16, &s_err);
            if (s == 0 && t == 0) hout[c] = h;
            return h;
        });
        publish(c + 1);
    }
    {
        const int c = k - 1;
        sweep(No{}, No{}, V + (int64_t)c * ldv, (const T *)nullptr, [&] {              // w .-= h v_k; norm(w)^2            :72, :75
            const T h = mgs_grid_sum_wide<T, NT>(cur + (size_t)c * stride, nseg, lds16, &s_err);
            if (s == 0 && t == 0) hout[c] = h;
            return h;
        });
        publish(k);
    }
    finish();
}

template <typename T, int NT, int RR, int RL>
__global__ __launch_bounds__(NT, 1) void k_mgs_resident(int64_t n, int k, const T *__restrict__ V, int64_t ldv, T *__restrict__ w, T *__restrict__ P, int kmax,
                                                        int stride, int nseg, int S, int parity, MgsMirror *mirror, unsigned long long seq)
{
    using vec = typename VT<T>::vec;
    constexpr int L = MIK_RED_L, SPR = NT / MIK_BLOCK;
    __shared__ vec wl[(RL > 0 ? RL : 1) * SPR * L * MIK_BLOCK];       // LDS-resident rounds: [round][sub][load][thread of the segment]
    __shared__ T lds16[16];
    __shared__ T segw[128][4];                                        // wave sums of this workgroup's segments (S <= 128)
    __shared__ int s_err;
    mgs_resident_body<T, NT, RR, RL>(n, k, V, ldv, w, P, kmax, stride, nseg, S, parity, mirror, seq, wl, lds16, segw, &s_err);
}
