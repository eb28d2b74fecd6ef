// mik_internal.h -- shared host structures and device helpers of libmik (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mik.h"
#include "../../include/mik_dev.h"

// ---------------------------------------------------------------------------------------------
// compile-time shape of the level-1 reduction tree (exported through mik_reduce_shape)
// ---------------------------------------------------------------------------------------------
constexpr int MIK_BLOCK = 256;       // threads per workgroup = 4 wave-64
constexpr int MIK_RED_L = 2;         // 16-byte loads per thread per segment
constexpr int MIK_FIN_THREADS = 1024;
constexpr int MIK_SPMV_TILE = 2048;  // nnz staged in LDS per row-block pass
constexpr int MIK_SPMV_G = 1;        // row-blocks per SpMV workgroup (= L of the fused-dot tree); >1 measured slower
constexpr int MIK_LONG_ROW = 256;      // rows with more entries go to the wave-per-row kernel (wave-shaped row sum)
constexpr int MIK_XCD_MAP = 8;       // the block -> XCD maps below (xcd_remap, spmv_block_map strips, the XCD-local Gram-Schmidt) are written for the
                                     // round-robin dispatch over the 8 XCDs of an unpartitioned MI355X; a context whose device reports another
                                     // count (CPX / partitioned modes) runs the identity map and the device-wide forms (mik_xcd_maps)

template <typename T> struct VT;
template <> struct VT<double> { static constexpr int W = 2; using vec = double2; };
template <> struct VT<float>  { static constexpr int W = 4; using vec = float4;  };

// ---------------------------------------------------------------------------------------------
// host structures
// ---------------------------------------------------------------------------------------------
struct mik_ctx {
    int device = 0;
    // the machine, queried in mik_ctx_create (hipDeviceGetAttribute); nothing on a selection path assumes 256 CUs / 8 XCDs (mik_cus / mik_xcds below)
    int cu_count = 0, xcd_count = 0, wave_size = 0;
    int64_t lds_per_cu = 0, l2_bytes = 0, hbm_bytes = 0;
    char arch[64] = {0};
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    void *partials = nullptr;        // level-1 segment sums (device)
    size_t partials_bytes = 0;
    void *coef = nullptr;            // small device array of coefficients / scalar results
    void *coef_host = nullptr;       // pinned host mirror
    hipEvent_t wait_event = nullptr; // for mik_wait
    void *pub = nullptr;             // pinned, device-mapped, coherent: PUB_BYTES of scalar results + the ticket of the last publication (read_scalars)
    unsigned long long pub_seq = 0;
    static constexpr size_t PUB_BYTES = 4096;
    const void *spmv_ep_w = nullptr, *spmv_ep_c = nullptr;   // the next fused-dot SpMV launch stores y = A x + (*c) w and sums x .* y (set and cleared by mik_minres_step)
    const void *spmv_ep_z = nullptr;                         // ... and sums z .* y instead of x .* y (dot(r_shadow, A u) of BiCGStab(l): set and cleared by mik_bicgstab_step)
    int tuning[MIK_KNOB_COUNT] = {0};   // development knobs of THIS context (include/mik_dev.h): a copy of the defaults at creation, mik_ctx_set_tuning
    static constexpr size_t COEF_BYTES = 8192;      // [0, 4096): coefficient blocks of the callers; the tail: scratch of mik_safe_norm_slow
    static constexpr size_t COEF_SAFE_SLOT = 4096;  // byte offset of that scratch
    // whole-iteration handles (mik_bicgstab / mik_minres) still alive: mik_ctx_destroy frees them, so that a host finalizer which
    // finds its context already closed (and must then skip mik_*_destroy) leaks nothing
    std::vector<std::pair<void *, int (*)(void *)>> owned;
};

// The machine shape the selection paths plan for: what the device reported, unless the development knob MIK_KNOB_MACHINE
// (compute units | XCDs << 16) overrides it for a test.
static inline int mik_cus(const mik_ctx *ctx) { const int v = ctx->tuning[MIK_KNOB_MACHINE] & 0xFFFF; return v > 0 ? v : std::max(1, ctx->cu_count); }
static inline int mik_xcds(const mik_ctx *ctx) { const int v = (ctx->tuning[MIK_KNOB_MACHINE] >> 16) & 0xFF; return v > 0 ? v : std::max(1, ctx->xcd_count); }
static inline bool mik_xcd_maps(const mik_ctx *ctx) { return mik_xcds(ctx) == MIK_XCD_MAP; }
// grid cap of the grid-stride sweeps: 32 workgroups (8 resident x 4 rounds) per compute unit; results never depend on it (segments are fixed)
static inline int mik_max_grid(const mik_ctx *ctx) { return mik_cus(ctx) * 32; }
// workgroups of a launch whose workgroups wait for each other (single-launch Gram-Schmidt, mailbox spins): one per compute unit
static inline int mik_resident_cap(const mik_ctx *ctx) { return mik_cus(ctx); }
// "strips" workgroup map of a banded operator (spmv_block_map): bandwidth bw in rows, nb 256-row blocks; 0 = identity
static inline int mik_strip_for(const mik_ctx *ctx, int64_t bw, int64_t nb)
{
    if (!mik_xcd_maps(ctx)) return 0;
    const int64_t P = ((bw + MIK_BLOCK - 1) / MIK_BLOCK + MIK_XCD_MAP - 1) / MIK_XCD_MAP * MIK_XCD_MAP;
    return (P >= MIK_XCD_MAP && P <= nb / 4) ? (int)P : 0;
}

int mik_mgs_resident_max_s();          // segments per workgroup up to which Modified Gram-Schmidt runs in the resident-w form (csrc/mik_mgs_res.h, mik_krylov.hip)

struct mik_csr {
    mik_ctx *ctx = nullptr;
    int dtype = MIK_F64;
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int *rowptr = nullptr;           // device, n_rows + 1
    int *col = nullptr;              // device, nnz (+ padding)
    void *val = nullptr;             // device, nnz (+ padding)
    int max_row_nnz = 0;
    int strip = 0;                   // workgroup map for banded operators (spmv_block_map), 0 = identity
    int force_layout = -1;           // mik_csr_set_layout: 0 = run on the CSR arrays, -1 = automatic
    int max_rowblock_nnz = 0;        // largest nnz of any 256-row block (short part)
    int n_long = 0;                  // rows longer than MIK_LONG_ROW, stored behind the short part
    int *xwin_lo = nullptr;          // device, one per 256-row block: first column of the block's window of x (k_spmv_rowblock XWIN); NULL = no windows
    int xwin_span = 0;               // elements of x per window (a multiple of 1 KiB)
    int *long_win = nullptr;         // device, one per long-row workgroup (4 virtual rows): first element of its window of x in LDS, or -1 (spmv_long_window)
    int long_lw = 0;                 // elements of x per such window (whole 1-KiB pieces)
    bool long_spread = false;        // most long-row workgroups have a window: they are spread over the merged launch instead of leading it
    unsigned char *rperm = nullptr;  // device, n_rows rounded up to whole 256-row blocks: thread t of a row-block sums row rperm[r0 + t] (k_spmv_rowblock RPERM); NULL = row t
    int *long_rows = nullptr;        // device: [n_long] targets, [n_long] start offsets, [n_long] lengths of the virtual rows, then the
                                     // segment tables: [n_seg] cut-row index, [n_cut] row, [n_cut] first segment, [n_cut] segments, [n_cut] tickets
    int n_seg = 0, n_cut = 0;        // rows longer than MIK_LONG_SEG are cut into n_seg segments (csrc/mik_spmv.h)
    void *seg_sum = nullptr;         // device: one partial per segment
    unsigned char *is_long = nullptr;   // device: n_rows flags (only when n_long > 0)
    // slice-constant values with up to 32 offsets per slice + one mask word per row (k_spmv_sdiaw, csrc/mik_sell.h)
    void *sdiaw_pats = nullptr;      // device, sdiaw_npat SdiawPattern<T>
    int *sdiaw_pat_id = nullptr;     // device, nb
    unsigned *sdiaw_mask = nullptr;  // device, n_rows
    void *sdiaw_uz = nullptr;        // device, one uint4 {all even, any even, all odd, any odd} per 128-row chunk (k_sdiaw_chunk_bits)
    int sdiaw_npat = 0, sdiaw_koff = 0, sdiaw_pat_bytes = 0;
    // jagged slices (csrc/mik_jds.h): operators with long near-uniform rows (finite elements)
    int *jds_ptr = nullptr;          // device, slices + 1: first group of every 64-row slice
    unsigned short *jds_len = nullptr;   // device, n_rows: entries of every row (MIK_JDS_LONG: a split-off long row)
    int *jds_col = nullptr;          // device, groups * W columns
    void *jds_val = nullptr;         // device, groups * W values
    int64_t jds_groups = 0;          // groups of W = 16 B / sizeof(T) entries
    int64_t jds_short_nnz = 0;       // entries held by the slices (the rest: split-off long rows)
    // sliced-ELL with per-slice offsets + per-row masks (k_spmv_sdia): every slice uses <= 8 distinct offsets
    int *sdia_ptr = nullptr;         // device, nb + 1
    int *sdia_off = nullptr;         // device, nb * 8
    int *sdia_tri = nullptr;         // device, nb: first slot of an (o-1, o, o+1) run, or -1
    unsigned char *sdia_mask = nullptr;   // device, n_rows
    void *sdia_val = nullptr;        // device, slot-major inside a slice (absent when the slice-constant form below exists)
    int64_t sdia_entries = 0;
    // slice-constant slot values (k_spmv_sdiac): distinct slice patterns {ns, tri, off[8], val[8]} + one pattern index per slice
    void *sdia_pats = nullptr;       // device, sdia_npat patterns (SdiaPattern<T>), or NULL
    int *sdia_pat_id = nullptr;      // device, nb
    int sdia_npat = 0;
    bool sdia_buf_ok = false;        // k_spmv_sdiab applies: finite pattern values, 32-bit row / slot byte offsets
    int sdia_koff = 0;               // - (most negative slot offset) of the operator, >= 0
    void *sdia_recs = nullptr;       // device, nb SdiaSliceRec (k_spmv_sdiab): scalar offsets + pattern shape per slice
    int sdia_cls = 0;                // (slots, centre slot) class k_spmv_sdiab runs specialised (mik_sdiab_cls_*), 0 = none
};

// norm(x) from t = sum of x_i^2: sqrt(t) whenever t lies inside [LO, HI] -- then no square that matters has
// underflowed and nothing has overflowed.  Outside (0, denormal range, Inf, NaN) the norm is recomputed with
// scaling (mik_safe_norm_slow); include/mik.h "Norms".  oracle/mik_oracle.c carries the same constants.
template <typename T> struct NrmRange;
template <> struct NrmRange<double> { static constexpr double LO = 0x1p-900, HI = 0x1p+900; static constexpr int EC = 1022; };
template <> struct NrmRange<float>  { static constexpr float LO = 0x1p-70f, HI = 0x1p+100f; static constexpr int EC = 126; };
template <typename T> __host__ __device__ static inline bool mik_nrm_in_range(T t) { return t >= NrmRange<T>::LO && t <= NrmRange<T>::HI; }

// Over-/underflow-safe norm of a device n-vector (amax pass, exact power-of-two scaling, the same fixed-shape
// tree over the scaled squares); *out is a host scalar.  Uses ctx->partials and the tail of ctx->coef; synchronises.
template <typename T> int mik_safe_norm_slow(mik_ctx *ctx, int64_t n, const T *x, T *out);

extern thread_local std::string g_mik_create_error;
extern int g_mik_tuning[MIK_KNOB_COUNT];  // DEFAULTS of the development knobs (what a new context starts with; mik_set_tuning also writes every live context)

int mik_fail(mik_ctx *ctx, int code, const char *fmt, ...);
// operator upload (mik_core.hip / mik_upload.hip)
size_t mik_sdia_pattern_bytes(size_t es);
int mik_sdiac_finish(mik_ctx *ctx, mik_csr *A, const std::vector<unsigned char> &desc, int64_t nb, size_t es, int64_t slots);
int mik_upload_device(mik_ctx *ctx, mik_csr *A, int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *ptr, const int64_t *idx,
                      const void *val, int index_base, int is_csc);

int mik_ensure_partials(mik_ctx *ctx, size_t bytes);
// device builders of the wide slice-constant layout and of the jagged slices from A's device CSR arrays (mik_upload.hip)
int mik_build_sdiaw_device(mik_ctx *ctx, mik_csr *A);
int mik_build_jds_device(mik_ctx *ctx, mik_csr *A);
bool mik_spmv_has_epilogue(const mik_csr *A);   // the active SpMV kernel of A takes the y = A x + c w epilogue (ctx->spmv_ep_*)
// The window rule of the product-tile kernel (k_spmv_rowblock XWIN), shared by the host builder (csr_build_xwin, mik_core.hip) and the
// device builder (mik_build_xwin_device, mik_upload.hip) so that they cannot disagree.  Input: per 256-row block the smallest / largest
// column its short rows reference and their entry count (cnt <= 0: no entries).  A block's window starts at its first column aligned
// DOWN to 16 bytes (the LDS-DMA source must be 16-byte aligned); all windows share one span = the widest qualifying block's, rounded up
// to whole 1-KiB pieces; a block that spans more than 32 KB of x is marked -1 (it gathers from memory).  A window that would leave x at
// the end of the vector slides down to the last aligned start that keeps it inside x -- when n_cols is not a multiple of 16 bytes that
// start leaves the last n_cols % W columns OUTSIDE the window, and the kernel has no memory fall-back inside a windowed block
// (an out-of-window column would be clamped to the window's last element): such a block is marked -1 too.
// Returns false when no window table should be built.
static inline bool mik_xwin_plan(int64_t nb, const int *mn, const int *mx, const int *cnt, size_t es, int64_t n_cols, int64_t total_entries,
                                 std::vector<int> &lo, int *span_out)
{
    const int W = (int)(16 / es), XP = (int)(1024 / es);
    const int64_t cap = 32768 / (int64_t)es - W;
    lo.assign((size_t)nb, 0);
    int64_t need = 0, inside = 0;                               // widest qualifying block; entries of the qualifying blocks
    for (int64_t b = 0; b < nb; ++b) {
        if (cnt[b] <= 0) continue;
        lo[(size_t)b] = mn[b] & ~(W - 1);
        const int64_t nd = (int64_t)mx[b] + 1 - lo[(size_t)b];
        if (nd > cap) { lo[(size_t)b] = -1; continue; }         // this block gathers from memory
        need = std::max(need, nd);
        inside += cnt[b];
    }
    if (need <= 0 || 4 * inside < 3 * total_entries) return false;
    const int64_t span = (need + W + XP - 1) / XP * XP;
    if (span + W > n_cols) return false;
    for (int64_t b = 0; b < nb; ++b) {
        if (cnt[b] <= 0 || lo[(size_t)b] < 0 || (int64_t)lo[(size_t)b] + span <= n_cols) continue;
        const int64_t slid = (n_cols - span) & ~(int64_t)(W - 1);
        lo[(size_t)b] = slid + span > (int64_t)mx[b] ? (int)slid : -1;
    }
    *span_out = (int)span;
    return true;
}
int mik_build_xwin_device(mik_ctx *ctx, mik_csr *A);
int mik_build_rperm_host(mik_ctx *ctx, mik_csr *A, const int *rowptr_host);   // rows of a block sorted by length over its threads (k_spmv_rowblock RPERM)   // after the jagged slices: windows of x for the product-tile kernel (k_spmv_rowblock XWIN)

#define MIK_HIP(ctx, call)                                                                    \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return mik_fail((ctx), e_ == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP,    \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,   \
                            __LINE__);                                                        \
    } while (0)

#define MIK_LAUNCH_CHECK(ctx)                                                                 \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess)                                                                 \
            return mik_fail((ctx), MIK_ERR_HIP, "kernel launch failed: %s (%s:%d)",           \
                            hipGetErrorString(e_), __FILE__, __LINE__);                       \
    } while (0)

static inline bool mik_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t mik_dtype_size(int dtype) { return dtype == MIK_F64 ? 8 : 4; }

// number of level-1 segments of an n-vector for dtype T
// number of SpMV workgroups (= fused-dot segment sums) for n rows
static inline int64_t mik_spmv_nwg(int64_t n)
{
    const int64_t rows = (int64_t)MIK_BLOCK * MIK_SPMV_G;
    return (n + rows - 1) / rows;
}

template <typename T> static inline int64_t mik_nseg(int64_t n)
{
    const int64_t seg = (int64_t)MIK_BLOCK * VT<T>::W * MIK_RED_L;
    return (n + seg - 1) / seg;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libmik's device code is written for gfx950 (MI355X) only: v_permlane32_swap / v_permlane16_swap, wave-wide DPP shifts, LDS-DMA and sc1 hand-offs have no fallback"
#endif

// A coefficient that is either an immediate or read from device memory (written by a finalise
// kernel earlier on the stream), optionally negated/never contracted.
template <typename T> struct Coef {
    const T *ptr;
    T val;
    __device__ __forceinline__ T get() const { return ptr ? *ptr : val; }
};
template <typename T> static inline Coef<T> coef_val(T v) { return Coef<T>{nullptr, v}; }
template <typename T> static inline Coef<T> coef_ptr(const T *p) { return Coef<T>{p, T(0)}; }

// The value `OFF` lanes further down the wave (OFF = 32, 16: gfx950 lane-swap instructions; 8, 4, 2, 1: DPP shifts inside a
// 16-lane row) for the lanes a shuffle-down tree reads at that level (lane l < OFF; other lanes: unspecified).  All of it
// runs on the vector ALU; __shfl_down goes through the LDS crossbar (ds_bpermute) and costs a wave ~12 dependent LDS
// round trips per tree.
template <int OFF> __device__ __forceinline__ unsigned lane_down_u32(unsigned v)
{
    if (OFF == 32) return __builtin_amdgcn_permlane32_swap(v, v, false, false)[1];   // lanes 0..31 <- lanes 32..63
    if (OFF == 16) return __builtin_amdgcn_permlane16_swap(v, v, false, false)[1];   // rows 0, 2 <- rows 1, 3
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + OFF, 0xf, 0xf, true);   // row_shl:OFF
}
template <int OFF> __device__ __forceinline__ double lane_down(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = lane_down_u32<OFF>((unsigned)b), hi = lane_down_u32<OFF>((unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int OFF> __device__ __forceinline__ float lane_down(float v)
{
    return __builtin_bit_cast(float, lane_down_u32<OFF>(__builtin_bit_cast(unsigned, v)));
}

// The value of the neighbouring lane across the WHOLE wave (DPP wave_shr:1 / wave_shl:1, GFX9 only): lane l receives lane
// l - 1 (from_below) or lane l + 1 (from_above); the lane without a neighbour (0 / 63) receives 0 bits.
template <bool BELOW> __device__ __forceinline__ unsigned lane_next_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, BELOW ? 0x138 : 0x130, 0xf, 0xf, true);
}
template <bool BELOW> __device__ __forceinline__ double lane_next(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = lane_next_u32<BELOW>((unsigned)b), hi = lane_next_u32<BELOW>((unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <bool BELOW> __device__ __forceinline__ float lane_next(float v)
{
    return __builtin_bit_cast(float, lane_next_u32<BELOW>(__builtin_bit_cast(unsigned, v)));
}

// wave-64 shuffle-down tree, offsets 32..1; the value in lane 0 is the tree sum
template <typename T> __device__ __forceinline__ T wave_tree(T v)
{
    v = v + lane_down<32>(v);
    v = v + lane_down<16>(v);
    v = v + lane_down<8>(v);
    v = v + lane_down<4>(v);
    v = v + lane_down<2>(v);
    v = v + lane_down<1>(v);
    return v;
}

// 256-thread block: wave tree, then the 4 wave sums left to right.  Result valid in thread 0.
template <typename T> __device__ __forceinline__ T block_tree_256(T v, T *lds4)
{
    v = wave_tree(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds4[w] = v;
    __syncthreads();
    T tot = T(0);
    if (threadIdx.x == 0) {
        tot = lds4[0];
        tot = tot + lds4[1];
        tot = tot + lds4[2];
        tot = tot + lds4[3];
    }
    __syncthreads();
    return tot;
}

// 1024-thread block: wave tree, then the 16 wave sums left to right.  Result valid in thread 0.
template <typename T> __device__ __forceinline__ T block_tree_1024(T v, T *lds16)
{
    v = wave_tree(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds16[w] = v;
    __syncthreads();
    T tot = T(0);
    if (threadIdx.x == 0) {
        tot = lds16[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tot = tot + lds16[i];
    }
    return tot;
}

// level 2: sum of m segment sums with the fixed 1024-thread shape; result valid in thread 0
template <typename T> __device__ __forceinline__ T level2_sum(const T *__restrict__ S, int64_t m, T *lds16)
{
    // Same ascending-j order as `for j: acc += S[j]`, but 32 (then 8) independent loads are issued before the
    // dependent add chain consumes them (a single workgroup is latency-bound, not bandwidth-bound).
    T acc = T(0);
    int64_t j = threadIdx.x;
    for (; j + 31 * (int64_t)MIK_FIN_THREADS < m; j += 32 * (int64_t)MIK_FIN_THREADS) {
        T v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = S[j + q * (int64_t)MIK_FIN_THREADS];
#pragma unroll
        for (int q = 0; q < 32; ++q) acc = acc + v[q];
    }
    for (; j + 7 * (int64_t)MIK_FIN_THREADS < m; j += 8 * (int64_t)MIK_FIN_THREADS) {
        T v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = S[j + q * (int64_t)MIK_FIN_THREADS];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = acc + v[q];
    }
    for (; j < m; j += MIK_FIN_THREADS) acc = acc + S[j];
    return block_tree_1024(acc, lds16);
}

// level 2 evaluated by a 256-thread workgroup for m <= 1024 segment sums: bit-identical to level2_sum
// (virtual thread vt holds 0 + S[vt]; wave tree per 64 virtual threads; 16 wave sums left to right).
// Every thread returns the total.  Lets a consumer kernel finalise its producer's reduction itself
// instead of waiting for a separate single-workgroup launch.
template <typename T> __device__ __forceinline__ T block_level2_256(const T *__restrict__ S, int m, T *lds16)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int vt = (w + 4 * j) * 64 + lane;
        T v = T(0);
        if (vt < m) v = v + S[vt];
        v = wave_tree(v);
        if (lane == 0) lds16[w + 4 * j] = v;
    }
    __syncthreads();
    T tot = lds16[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) tot = tot + lds16[q];
    __syncthreads();
    return tot;
}

// correctly rounded square root (IEEE): the product path needs sqrt(rr) to match the host's.
// NB: HIP's __fsqrt_rn is __ocml_native_sqrt_f32 (NOT correctly rounded) unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined; sqrtf()/sqrt() lower to the correctly rounded OCML
// routines (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).
__device__ __forceinline__ double mik_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float mik_sqrt(float x) { return sqrtf(x); }

// block -> row-block map that hands every XCD (block b runs on XCD b % 8) one contiguous range
__device__ __forceinline__ int xcd_remap(int b, int nb)
{
    const int xcd = b & 7, idx = b >> 3;
    const int q = nb >> 3, rem = nb & 7;
    return xcd * q + (xcd < rem ? xcd : rem) + idx;
}

// Wait for everything enqueued on the ctx stream.  hipStreamSynchronize parks the host thread when the queue
// is not about to drain and takes hundreds of microseconds to come back -- more than the kernels of one solver
// iteration -- so the per-iteration scalar reads spin on an event instead (development knob MIK_KNOB_HOST_WAIT bit 1: plain synchronize).
static inline hipError_t mik_wait(mik_ctx *ctx)
{
    if ((ctx->tuning[MIK_KNOB_HOST_WAIT] & 2) || !ctx->wait_event) return hipStreamSynchronize(ctx->stream);
    hipError_t e = hipEventRecord(ctx->wait_event, ctx->stream);
    if (e != hipSuccess) return e;
    while ((e = hipEventQuery(ctx->wait_event)) == hipErrorNotReady) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return e;
}

#define MIK_TRY(expr)                 \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != MIK_OK) return rc_; \
    } while (0)


// `count` values of the device array `dev`, as soon as everything enqueued before has run.  A one-wave kernel copies them into the
// context's pinned mailbox and then stores a ticket (system scope); the host spins on the ticket -- no copy engine, no event in
// the path (hipMemcpyAsync + event spin, MIK_KNOB_HOST_WAIT bit 0, takes several microseconds longer per read, and the solvers that
// are driven from the host -- MINRES, BiCGStab(l), the generic GMRES path -- read 2 ... 5 scalars per iteration).
template <typename T>
__global__ __launch_bounds__(64) void k_publish(const T *__restrict__ src, int count, T *__restrict__ dst, unsigned long long *ticket, unsigned long long seq)
{
    for (int i = (int)threadIdx.x; i < count; i += 64) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(ticket, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T> static inline int mik_read_scalars(mik_ctx *ctx, const T *dev, int count, T *host_out)
{
    if ((ctx->tuning[MIK_KNOB_HOST_WAIT] & 1) || !ctx->pub || sizeof(T) * (size_t)count > mik_ctx::PUB_BYTES) {
        MIK_HIP(ctx, hipMemcpyAsync(ctx->coef_host, dev, sizeof(T) * count, hipMemcpyDeviceToHost, ctx->stream));
        MIK_HIP(ctx, mik_wait(ctx));
        memcpy(host_out, ctx->coef_host, sizeof(T) * count);
        return MIK_OK;
    }
    unsigned long long *ticket = reinterpret_cast<unsigned long long *>((unsigned char *)ctx->pub + mik_ctx::PUB_BYTES);
    const unsigned long long want = ++ctx->pub_seq;
    hipLaunchKernelGGL((k_publish<T>), dim3(1), dim3(64), 0, ctx->stream, dev, count, (T *)ctx->pub, ticket, want);
    MIK_LAUNCH_CHECK(ctx);
    for (unsigned long long spins = 0;; ++spins) {
        if (__atomic_load_n(ticket, __ATOMIC_ACQUIRE) == want) break;
        if ((spins & 0xFFFFF) == 0xFFFFF) {                  // every ~1M polls: has the stream died, or finished without publishing?
            const hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) {
                if (__atomic_load_n(ticket, __ATOMIC_ACQUIRE) == want) break;
                return mik_fail(ctx, MIK_ERR_HIP, "scalar read: stream idle but ticket %llu was never published", want);
            }
            if (e != hipErrorNotReady) return mik_fail(ctx, MIK_ERR_HIP, "scalar read: %s", hipGetErrorString(e));
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    memcpy(host_out, ctx->pub, sizeof(T) * count);
    return MIK_OK;
}


#endif  // __HIPCC__
