// mik_core.hip -- context, device memory, CSR upload, and the L1 operator/vector entry points
// (mul!, dot, norm, broadcast forms) of include/mik.h.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <limits>
#include <new>

#include "mik_kernels.h"
#include "mik_spmv.h"
#include <algorithm>
#include <mutex>

#include "mik_sell.h"
#include "mik_jds.h"
#include <map>
#include <string>
#include <unordered_map>

thread_local std::string g_mik_create_error;
int g_mik_tuning[MIK_KNOB_COUNT] = {0};
// every live context: mik_set_tuning (the process-wide development setter) writes the defaults AND all of them; launches only ever read
// their own context's table
static std::vector<mik_ctx *> g_mik_contexts;
static std::mutex g_mik_contexts_mu;

int mik_fail(mik_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_mik_create_error = buf;
    return code;
}

int mik_ensure_partials(mik_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->partials_bytes) return MIK_OK;
    size_t want = std::max(bytes, (size_t)1 << 20);
    if (ctx->partials) {
        MIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MIK_HIP(ctx, hipFree(ctx->partials));
        ctx->partials = nullptr;
        ctx->partials_bytes = 0;
    }
    MIK_HIP(ctx, hipMalloc(&ctx->partials, want));
    ctx->partials_bytes = want;
    return MIK_OK;
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// library / context
// ---------------------------------------------------------------------------------------------
extern "C" int mik_abi_version(void) { return MIK_ABI_VERSION; }

extern "C" int mik_device_count(int *count)
{
    if (!count) return MIK_ERR_INVALID;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return mik_fail(nullptr, MIK_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = c;
    return MIK_OK;
}

extern "C" int mik_ctx_create(int device, mik_ctx **out)
{
    if (!out) return mik_fail(nullptr, MIK_ERR_INVALID, "mik_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return mik_fail(nullptr, MIK_ERR_HIP, "mik_ctx_create: no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count)
        return mik_fail(nullptr, MIK_ERR_INVALID, "mik_ctx_create: device %d out of range [0, %d)", device, count);
    mik_ctx *ctx = new (std::nothrow) mik_ctx();
    if (!ctx) return mik_fail(nullptr, MIK_ERR_NOMEM, "mik_ctx_create: host allocation failed");
    ctx->device = device;
    auto bail = [&](hipError_t err, const char *what) {
        int rc = mik_fail(nullptr, MIK_ERR_HIP, "mik_ctx_create: %s: %s", what, hipGetErrorString(err));
        delete ctx;
        return rc;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    ctx->stream = ctx->own_stream;
    {   // the machine: compute units, XCDs, LDS, L2 -- every "one workgroup per CU" cap and XCD map is derived from this (mik_ctx_info)
        int v = 0;
        if ((e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return bail(e, "hipDeviceGetAttribute(MultiprocessorCount)");
        ctx->cu_count = v;
        v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, device) != hipSuccess || v < 1) { (void)hipGetLastError(); v = 1; }   // (a runtime without the attribute: one XCD = identity maps, device-wide forms)
        ctx->xcd_count = v;
        v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeWarpSize, device) == hipSuccess) ctx->wave_size = v; else (void)hipGetLastError();
        v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) == hipSuccess) ctx->lds_per_cu = v; else (void)hipGetLastError();
        v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeL2CacheSize, device) == hipSuccess) ctx->l2_bytes = v; else (void)hipGetLastError();
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
            ctx->hbm_bytes = (int64_t)prop.totalGlobalMem;
            snprintf(ctx->arch, sizeof(ctx->arch), "%s", prop.gcnArchName);
        } else (void)hipGetLastError();
        if (ctx->wave_size != 0 && ctx->wave_size != 64) {
            int rc = mik_fail(nullptr, MIK_ERR_HIP, "mik_ctx_create: device %d runs %d-wide wavefronts; libmik's kernels are written for wave-64 (gfx950)", device, ctx->wave_size);
            delete ctx;
            return rc;
        }
    }
    if ((e = hipMalloc(&ctx->coef, mik_ctx::COEF_BYTES)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipHostMalloc(&ctx->coef_host, mik_ctx::COEF_BYTES, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
    if ((e = hipEventCreateWithFlags(&ctx->wait_event, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipHostMalloc(&ctx->pub, mik_ctx::PUB_BYTES + 64, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail(e, "hipHostMalloc");
    memset(ctx->pub, 0, mik_ctx::PUB_BYTES + 64);
    {
        std::lock_guard<std::mutex> lk(g_mik_contexts_mu);
        memcpy(ctx->tuning, g_mik_tuning, sizeof(ctx->tuning));
        g_mik_contexts.push_back(ctx);
    }
    *out = ctx;
    return MIK_OK;
}

extern "C" int mik_ctx_info(const mik_ctx *ctx, mik_device_info *out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    out->device = ctx->device;
    out->compute_units = ctx->cu_count;
    out->xcds = ctx->xcd_count;
    out->wavefront_size = ctx->wave_size;
    out->lds_bytes_per_cu = ctx->lds_per_cu;
    out->l2_bytes = ctx->l2_bytes;
    out->hbm_bytes = ctx->hbm_bytes;
    snprintf(out->arch, sizeof(out->arch), "%s", ctx->arch);
    // what the selection paths derive from it (development knob MIK_KNOB_MACHINE applied)
    out->planned_compute_units = mik_cus(ctx);
    out->planned_xcds = mik_xcds(ctx);
    out->xcd_maps = mik_xcd_maps(ctx) ? 1 : 0;
    out->resident_workgroup_cap = mik_resident_cap(ctx);
    out->gs_single_launch_max_segments = 8 * mik_resident_cap(ctx);
    out->gs_xcd_local_max_workgroups = mik_xcd_maps(ctx) ? 4 * (mik_cus(ctx) / mik_xcds(ctx)) : 0;
    out->sweep_grid_cap = mik_max_grid(ctx);
    out->mgs_resident_max_segments = ctx->lds_per_cu >= 160 * 1024 ? mik_mgs_resident_max_s() * mik_resident_cap(ctx) : 0;
    return MIK_OK;
}

extern "C" int mik_ctx_destroy(mik_ctx *ctx)
{
    if (!ctx) return MIK_OK;
    {
        std::lock_guard<std::mutex> lk(g_mik_contexts_mu);
        g_mik_contexts.erase(std::remove(g_mik_contexts.begin(), g_mik_contexts.end(), ctx), g_mik_contexts.end());
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    {   // step handles the host never destroyed (their destroy functions unregister themselves: walk a copy)
        std::vector<std::pair<void *, int (*)(void *)>> owned;
        owned.swap(ctx->owned);
        for (auto &h : owned) (void)h.second(h.first);
    }
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->coef) (void)hipFree(ctx->coef);
    if (ctx->coef_host) (void)hipHostFree(ctx->coef_host);
    if (ctx->pub) (void)hipHostFree(ctx->pub);
    if (ctx->wait_event) (void)hipEventDestroy(ctx->wait_event);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return MIK_OK;
}

extern "C" int mik_ctx_set_stream(mik_ctx *ctx, void *hip_stream)
{
    if (!ctx) return MIK_ERR_INVALID;
    MIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return MIK_OK;
}

extern "C" int mik_ctx_synchronize(mik_ctx *ctx)
{
    if (!ctx) return MIK_ERR_INVALID;
    MIK_HIP(ctx, mik_wait(ctx));
    return MIK_OK;
}

extern "C" const char *mik_last_error(mik_ctx *ctx) { return ctx ? ctx->err.c_str() : g_mik_create_error.c_str(); }

extern "C" int mik_spmv_dot_shape(int *W, int *L)
{
    if (W) *W = 1;
    if (L) *L = MIK_SPMV_G;
    return MIK_OK;
}

extern "C" int mik_spmv_long_row(int *threshold)
{
    if (threshold) *threshold = MIK_LONG_ROW;
    return MIK_OK;
}

extern "C" int mik_spmv_long_segment(int *segment)
{
    if (segment) *segment = ((g_mik_tuning[MIK_KNOB_LONG_SEGMENT] > 0 ? g_mik_tuning[MIK_KNOB_LONG_SEGMENT] : MIK_LONG_SEG) + 3) & ~3;   // whole groups of MIK_LONG_G entries
    return MIK_OK;
}

extern "C" int mik_spmv_long_group(int *group)
{
    if (group) *group = MIK_LONG_G;
    return MIK_OK;
}

static inline bool spmv_csr_rowgather(const mik_csr *A);

extern "C" int mik_ctx_set_tuning(mik_ctx *ctx, int key, int value)
{
    if (!ctx || key < 0 || key >= MIK_KNOB_COUNT) return MIK_ERR_INVALID;
    ctx->tuning[key] = value;
    return MIK_OK;
}

extern "C" int mik_set_tuning(int key, int value)
{
    if (key < 0 || key >= MIK_KNOB_COUNT) return MIK_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_mik_contexts_mu);
    g_mik_tuning[key] = value;
    for (mik_ctx *c : g_mik_contexts) c->tuning[key] = value;
    return MIK_OK;
}

// Host-only (no device needed): the window rule of the product-tile kernel on caller-supplied block statistics, so that the rule
// itself is testable on a box without a GPU (tests/test_host_logic.py replays ADVICE r4's n = 9001 case).
extern "C" int mik_dev_xwin_plan(int64_t n_blocks, const int *first_col, const int *last_col, const int *entries, int elem_size, int64_t n_cols,
                                 int64_t total_entries, int *win_lo, int *span)
{
    if (n_blocks < 0 || !first_col || !last_col || !entries || !win_lo || !span || (elem_size != 4 && elem_size != 8)) return MIK_ERR_INVALID;
    std::vector<int> lo;
    int sp = 0;
    const bool built = mik_xwin_plan(n_blocks, first_col, last_col, entries, (size_t)elem_size, n_cols, total_entries, lo, &sp);
    *span = built ? sp : 0;
    for (int64_t b = 0; b < n_blocks; ++b) win_lo[b] = built ? lo[(size_t)b] : -1;
    return MIK_OK;
}

extern "C" int mik_reduce_shape(int dtype, int *W, int *L)
{
    if (dtype != MIK_F64 && dtype != MIK_F32) return MIK_ERR_INVALID;
    if (W) *W = dtype == MIK_F64 ? VT<double>::W : VT<float>::W;
    if (L) *L = MIK_RED_L;
    return MIK_OK;
}

// ---------------------------------------------------------------------------------------------
// device memory
// ---------------------------------------------------------------------------------------------
extern "C" int mik_malloc(mik_ctx *ctx, size_t bytes, void **dptr)
{
    if (!ctx || !dptr) return MIK_ERR_INVALID;
    *dptr = nullptr;
    MIK_HIP(ctx, hipSetDevice(ctx->device));
    MIK_HIP(ctx, hipMalloc(dptr, bytes ? bytes : 16));
    return MIK_OK;
}

extern "C" int mik_free(mik_ctx *ctx, void *dptr)
{
    if (!ctx) return MIK_ERR_INVALID;
    if (!dptr) return MIK_OK;
    MIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MIK_HIP(ctx, hipFree(dptr));
    return MIK_OK;
}

extern "C" int mik_memcpy_h2d(mik_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return MIK_ERR_INVALID;
    if (!bytes) return MIK_OK;
    MIK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    MIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MIK_OK;
}

extern "C" int mik_memcpy_d2h(mik_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return MIK_ERR_INVALID;
    if (!bytes) return MIK_OK;
    MIK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MIK_OK;
}

extern "C" int mik_copy(mik_ctx *ctx, int dtype, int64_t n, const void *x, void *y)
{
    if (!ctx || n < 0 || (n && (!x || !y))) return MIK_ERR_INVALID;
    if (n == 0 || x == y) return MIK_OK;
    MIK_HIP(ctx, hipMemcpyAsync(y, x, (size_t)n * mik_dtype_size(dtype), hipMemcpyDeviceToDevice, ctx->stream));
    return MIK_OK;
}

template <typename T> static int fill_impl(mik_ctx *ctx, int64_t n, const void *value, void *x)
{
    OpFill<T> op{(T *)x, *(const T *)value};
    return launch_map<T>(ctx, n, op, mik_aligned16(x), (T *)nullptr, nullptr);
}

extern "C" int mik_fill(mik_ctx *ctx, int dtype, int64_t n, const void *value, void *x)
{
    if (!ctx || n < 0 || !value || (n && !x)) return MIK_ERR_INVALID;
    return dtype == MIK_F64 ? fill_impl<double>(ctx, n, value, x) : fill_impl<float>(ctx, n, value, x);
}

// k_spmv_sdiab relies on two properties of raw buffer loads on this GPU: the scalar offset is added to the address, and a
// vector offset of all ones is out of range and reads 0.0 whatever the scalar offset.  Checked once per process on the
// device itself; if the probe disagrees the slice-constant layout keeps its flat-load kernel (k_spmv_sdiac).
__global__ void k_probe_buffer_range(const double *x, double *out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)x, (short)0, (int)0xFFFFFFF0u, (int)0x00020000);
    const int t = threadIdx.x;
    out[t] = buffer_gather<double>(rs, (t & 1) ? ~0u : (unsigned)t * 4u, 8);     // even lanes: x[t / 2 + 1]; odd lanes: out of range
    // stores beyond the descriptor's range must be dropped: out[8..15] are covered by a 4-element descriptor whose lanes 4..7 miss
    const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void *)(out + 8), (short)0, 4 * 8, (int)0x00020000);
    buffer_put<double>(ws, (unsigned)t * 8u, -1.0, 0);
}

static bool buffer_range_semantics_ok(mik_ctx *ctx)
{
    static int state = -1;                                                         // -1 unknown, 0 no, 1 yes
    if (state >= 0) return state == 1;
    double h[16], r[16], *d = nullptr;
    for (int i = 0; i < 16; ++i) h[i] = 1.0 + i;
    state = 0;
    if (hipMalloc((void **)&d, sizeof(h) + sizeof(r)) != hipSuccess) return false;
    if (hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) == hipSuccess && hipMemset(d + 16, 0, sizeof(r)) == hipSuccess) {
        hipLaunchKernelGGL(k_probe_buffer_range, dim3(1), dim3(8), 0, ctx->stream, d, d + 16);
        if (hipStreamSynchronize(ctx->stream) == hipSuccess && hipMemcpy(r, d + 16, sizeof(r), hipMemcpyDeviceToHost) == hipSuccess) {
            bool ok = true;
            for (int t = 0; t < 8; ++t) ok = ok && r[t] == ((t & 1) ? 0.0 : h[t / 2 + 1]) && r[8 + t] == (t < 4 ? -1.0 : 0.0);
            state = ok ? 1 : 0;
        }
    }
    (void)hipFree(d);
    return state == 1;
}

// Slice descriptions (one SdiaPattern-shaped record per slice, soff still empty) -> the pattern table of the slice-constant
// layout: equal descriptions are stored once, a slice keeps a pattern index.  Shared by the host builder below and the
// device-side upload (mik_upload.hip).  Also decides whether k_spmv_sdiab applies (finite values, 32-bit byte offsets,
// range-check semantics of the device) and which (slots, centre slot) class it runs specialised.
size_t mik_sdia_pattern_bytes(size_t es) { return 16 + 32 + 32 + 8 * es; }

int mik_sdiac_finish(mik_ctx *ctx, mik_csr *A, const std::vector<unsigned char> &desc, int64_t nb, size_t es, int64_t slots)
{
    hipError_t e;
    const int64_t n_rows = A->n_rows;
    const size_t psz = mik_sdia_pattern_bytes(es), voff0 = 80;
    std::map<std::string, int> index;
    std::vector<unsigned char> pats;
    std::vector<int> pid((size_t)nb, 0);
    std::vector<int64_t> slices_of;                            // slices per pattern
    for (int64_t b = 0; b < nb; ++b) {
        std::string key((const char *)&desc[(size_t)b * psz], psz);
        auto it = index.find(key);
        if (it == index.end()) {
            it = index.emplace(key, (int)index.size()).first;
            pats.insert(pats.end(), key.begin(), key.end());
            slices_of.push_back(0);
        }
        pid[(size_t)b] = it->second;
        ++slices_of[(size_t)it->second];
    }
    // k_spmv_sdiab (buffer loads; absent slots read 0.0 and add value * 0): needs finite values and 32-bit byte
    // offsets; the scalar offset of slot q is (off[q] + koff) * sizeof(T), koff = - the operator's smallest offset
    {
        int64_t omin = 0, omax = 0;
        bool finite = true;
        for (size_t ip = 0; ip < index.size(); ++ip) {
            int hdr[4], off[8];
            memcpy(hdr, &pats[ip * psz], 16);
            memcpy(off, &pats[ip * psz + 16], 32);
            for (int q = 0; q < hdr[0]; ++q) {
                omin = std::min<int64_t>(omin, off[q]);
                omax = std::max<int64_t>(omax, off[q]);
                double vq;
                if (es == 8) memcpy(&vq, &pats[ip * psz + voff0 + 8 * (size_t)q], 8);
                else { float f; memcpy(&f, &pats[ip * psz + voff0 + 4 * (size_t)q], 4); vq = f; }
                finite = finite && std::isfinite(vq);
            }
        }
        A->sdia_koff = (int)(-omin);
        A->sdia_buf_ok = finite && (uint64_t)n_rows * es <= 0xFFFFFFF0ull && (uint64_t)(omax - omin) * es < 0x7FFFFFF0ull &&
                         buffer_range_semantics_ok(ctx);
        // the (slots, centre slot) class most slices have, among the ones k_spmv_sdiab is specialised for
        int64_t best = 0;
        A->sdia_cls = 0;
        for (int c = 1; c < MIK_SDIAB_NCLS; ++c) {
            int64_t cnt = 0;
            for (size_t ip = 0; ip < index.size(); ++ip) {
                int hdr[4];
                memcpy(hdr, &pats[ip * psz], 16);
                if (hdr[0] == mik_sdiab_cls_ns(c) && hdr[2] == mik_sdiab_cls_cq(c)) cnt += slices_of[ip];
            }
            if (cnt > best) { best = cnt; A->sdia_cls = c; }
        }
        if (A->sdia_buf_ok)
            for (size_t ip = 0; ip < index.size(); ++ip) {
                int off[8], soff[8];
                memcpy(off, &pats[ip * psz + 16], 32);
                for (int q = 0; q < 8; ++q) soff[q] = (int)(((int64_t)off[q] + A->sdia_koff) * (int64_t)es);
                memcpy(&pats[ip * psz + 48], soff, 32);
            }
    }
    if ((e = hipMalloc((void **)&A->sdia_pat_id, sizeof(int) * (size_t)nb)) != hipSuccess ||
        (e = hipMalloc(&A->sdia_pats, pats.size())) != hipSuccess ||
        (e = hipMemcpy(A->sdia_pat_id, pid.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(A->sdia_pats, pats.data(), pats.size(), hipMemcpyHostToDevice)) != hipSuccess) {
        return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: slice-constant form: %s", hipGetErrorString(e));
    }
    A->sdia_npat = (int)index.size();
    A->sdia_entries = slots;
    if (A->sdia_buf_ok) {                                              // per-slice records of k_spmv_sdiab
        std::vector<SdiaSliceRec> recs((size_t)nb);
        for (int64_t b = 0; b < nb; ++b) {
            const unsigned char *pp = &pats[(size_t)pid[(size_t)b] * psz];
            SdiaSliceRec &r = recs[(size_t)b];
            memset(&r, 0, sizeof(r));
            int hdr[4];
            memcpy(hdr, pp, 16);
            memcpy(r.soff, pp + 48, 32);
            r.ns = hdr[0]; r.cq = hdr[2]; r.dfull = hdr[3]; r.pid = pid[(size_t)b];
        }
        if ((e = hipMalloc(&A->sdia_recs, sizeof(SdiaSliceRec) * (size_t)nb)) != hipSuccess ||
            (e = hipMemcpy(A->sdia_recs, recs.data(), sizeof(SdiaSliceRec) * (size_t)nb, hipMemcpyHostToDevice)) != hipSuccess)
            return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: slice records: %s", hipGetErrorString(e));
    }
    return MIK_OK;
}

// ---------------------------------------------------------------------------------------------
// sliced-ELL layout builders (host side of csrc/mik_sell.h)
// ---------------------------------------------------------------------------------------------
// Sliced-ELL with per-slice offsets and per-row masks (k_spmv_sdia): every 256-row slice uses at most 8
// distinct (column - row) offsets and the slot padding stays below 1/8 extra entries.
// Leaves A->sdia_* unset (and returns MIK_OK) when the operator does not qualify.
static int csr_build_sdia(mik_ctx *ctx, mik_csr *A, const std::vector<int> &rowptr, const std::vector<int> &col, const std::vector<unsigned char> &v,
        size_t es, int64_t n_rows, int64_t n_cols, int64_t nnz, int max_row)
{
    (void)max_row;
    hipError_t e;
    if (A->n_long == 0 && n_rows > 0 && nnz > 0 && n_cols > 0 && (ctx->tuning[MIK_KNOB_LAYOUTS] & 1) == 0 && (ctx->tuning[MIK_KNOB_LAYOUTS] & 4) == 0)
    {
        const int64_t nb = (n_rows + MIK_BLOCK - 1) / MIK_BLOCK;
        std::vector<int> dptr((size_t)nb + 1, 0), doff((size_t)nb * 8, 0), dtri((size_t)nb, -1);
        bool ok = true;
        int64_t slots = 0;
        for (int64_t b = 0; b < nb && ok; ++b) {
            // The slice's slot pattern: a common super-sequence of its rows' offset sequences (each row lists its
            // entries in the order they are summed -- ascending GLOBAL column, which for a rank's block with halo
            // columns is not ascending local offset), built by merging row after row.
            int offs8[8];
            int ns = 0;
            const int64_t rend = std::min<int64_t>((b + 1) * MIK_BLOCK, n_rows);
            for (int64_t r = b * MIK_BLOCK; r < rend && ok; ++r) {
                int p = 0;                                         // next admissible pattern position for this row
                for (int k2 = rowptr[r]; k2 < rowptr[r + 1]; ++k2) {
                    const int d = col[(size_t)k2] - (int)r;
                    int q = 0;
                    while (q < ns && offs8[q] != d) ++q;
                    if (q < ns) {
                        if (q < p) { ok = false; break; }          // two rows order the same offsets differently
                        p = q + 1;
                    } else {
                        if (ns == 8) { ok = false; break; }
                        for (int z = ns; z > p; --z) offs8[z] = offs8[z - 1];
                        offs8[p] = d;
                        ++ns;
                        ++p;
                    }
                }
            }
            for (int q = 0; q < ns; ++q) doff[(size_t)b * 8 + q] = offs8[q];
            for (int q = 0; q + 2 < ns; ++q)
                if (offs8[q + 1] == offs8[q] + 1 && offs8[q + 2] == offs8[q] + 2) { dtri[(size_t)b] = q; break; }
            slots += (int64_t)ns * MIK_BLOCK;
            if (slots >= INT32_MAX) ok = false;
            dptr[(size_t)b + 1] = (int)slots;
        }
        if (ok && slots <= nnz + nnz / 8 + 8 * MIK_BLOCK) {
            std::vector<unsigned char> dval, dmask;
            try {
                dval.assign((size_t)slots * es, 0);
                dmask.assign((size_t)n_rows, 0);
            } catch (const std::bad_alloc &) {
                return mik_fail(ctx, MIK_ERR_NOMEM, "mik_csr_create: host staging allocation failed (sliced-diagonal form)");
            }
            for (int64_t r = 0; r < n_rows && ok; ++r) {
                const int64_t b = r / MIK_BLOCK, t = r % MIK_BLOCK;
                const int ns = (dptr[(size_t)b + 1] - dptr[(size_t)b]) / MIK_BLOCK;
                const int *so = &doff[(size_t)b * 8];
                int q = 0, prevq = -1;
                for (int k2 = rowptr[r]; k2 < rowptr[r + 1]; ++k2) {
                    const int d = col[(size_t)k2] - (int)r;
                    while (q < ns && so[q] != d) ++q;             // columns ascend within a row, so do the slots
                    if (q >= ns || q <= prevq) { ok = false; break; }   // unsorted or duplicate column: keep the other layouts
                    const size_t dst = (size_t)dptr[(size_t)b] + (size_t)q * MIK_BLOCK + (size_t)t;
                    memcpy(&dval[dst * es], &v[(size_t)k2 * es], es);
                    dmask[(size_t)r] |= (unsigned char)(1u << q);
                    prevq = q;
                }
            }
            // slice-constant slots (k_spmv_sdiac): every row of a slice that has slot q carries the same value BITS there
            std::vector<unsigned char> cval;
            bool constant = ok && (ctx->tuning[MIK_KNOB_LAYOUTS] & 2) == 0;
            if (constant) {
                cval.assign((size_t)nb * 8 * es, 0);
                std::vector<unsigned char> seen((size_t)nb * 8, 0);
                for (int64_t r = 0; r < n_rows && constant; ++r) {
                    const int64_t b = r / MIK_BLOCK, t = r % MIK_BLOCK;
                    const int ns = (dptr[(size_t)b + 1] - dptr[(size_t)b]) / MIK_BLOCK;
                    for (int q = 0; q < ns; ++q) {
                        if (!((dmask[(size_t)r] >> q) & 1)) continue;
                        const unsigned char *src = &dval[((size_t)dptr[(size_t)b] + (size_t)q * MIK_BLOCK + (size_t)t) * es];
                        unsigned char *dst = &cval[((size_t)b * 8 + q) * es];
                        if (!seen[(size_t)b * 8 + q]) { memcpy(dst, src, es); seen[(size_t)b * 8 + q] = 1; }
                        else if (memcmp(dst, src, es) != 0) { constant = false; break; }
                    }
                }
            }
            if (ok && constant) {
                // per-slice descriptions {ns, tri, centre slot, diagonal-in-every-row, offsets, value bits} -> pattern table
                const size_t psz = mik_sdia_pattern_bytes(es);
                std::vector<unsigned char> desc((size_t)nb * psz, 0);
                for (int64_t b = 0; b < nb; ++b) {
                    const int ns = (dptr[(size_t)b + 1] - dptr[(size_t)b]) / MIK_BLOCK;
                    int cq = -1;
                    for (int q = 0; q < ns; ++q)
                        if (doff[(size_t)b * 8 + q] == 0) cq = q;
                    int dfull = cq >= 0;
                    for (int64_t r = b * MIK_BLOCK; dfull && r < std::min<int64_t>(n_rows, (b + 1) * MIK_BLOCK); ++r)
                        dfull = (dmask[(size_t)r] >> cq) & 1;
                    int hdr[4] = {ns, dtri[(size_t)b], cq, dfull};
                    memcpy(&desc[(size_t)b * psz], hdr, 16);
                    memcpy(&desc[(size_t)b * psz + 16], &doff[(size_t)b * 8], 32);
                    memcpy(&desc[(size_t)b * psz + 80], &cval[(size_t)b * 8 * es], 8 * es);
                }
                if ((e = hipMalloc((void **)&A->sdia_mask, (size_t)n_rows)) != hipSuccess ||
                    (e = hipMemcpy(A->sdia_mask, dmask.data(), (size_t)n_rows, hipMemcpyHostToDevice)) != hipSuccess)
                    return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: slice-constant form: %s", hipGetErrorString(e));
                const int rc = mik_sdiac_finish(ctx, A, desc, nb, es, slots);
                if (rc != MIK_OK) return rc;
            } else if (ok) {
                if ((e = hipMalloc((void **)&A->sdia_ptr, sizeof(int) * ((size_t)nb + 1))) != hipSuccess ||
                    (e = hipMalloc((void **)&A->sdia_off, sizeof(int) * (size_t)nb * 8)) != hipSuccess ||
                    (e = hipMalloc((void **)&A->sdia_tri, sizeof(int) * (size_t)nb)) != hipSuccess ||
                    (e = hipMemcpy(A->sdia_tri, dtri.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice)) != hipSuccess ||
                    (e = hipMalloc((void **)&A->sdia_mask, (size_t)n_rows)) != hipSuccess ||
                    (e = hipMalloc(&A->sdia_val, es * (size_t)std::max<int64_t>(slots, 1))) != hipSuccess ||
                    (e = hipMemcpy(A->sdia_ptr, dptr.data(), sizeof(int) * ((size_t)nb + 1), hipMemcpyHostToDevice)) != hipSuccess ||
                    (e = hipMemcpy(A->sdia_off, doff.data(), sizeof(int) * (size_t)nb * 8, hipMemcpyHostToDevice)) != hipSuccess ||
                    (e = hipMemcpy(A->sdia_mask, dmask.data(), (size_t)n_rows, hipMemcpyHostToDevice)) != hipSuccess ||
                    (slots && (e = hipMemcpy(A->sdia_val, dval.data(), es * (size_t)slots, hipMemcpyHostToDevice)) != hipSuccess)) {
                    return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: sliced-diagonal form: %s", hipGetErrorString(e));
                }
                A->sdia_entries = slots;
            }
        }
    }
    return MIK_OK;
}

// Pattern table of the wide slice-constant layout from the distinct slice patterns (sorted {offset, value bits} lists): scalar byte
// offsets, values and the item decomposition of k_spmv_sdiaw2.  MIK_ERR_NOTIMPL when the byte offsets do not fit.  Shared by the host
// builder below and the device builder of mik_upload.hip.
int mik_sdiaw_finish(mik_ctx *ctx, mik_csr *A, const std::vector<std::vector<std::pair<int, uint64_t>>> &pats, size_t es, int64_t n_cols)
{
    int64_t dmin = 0, dmax = 0;
    for (const auto &pt : pats) for (const auto &s2 : pt) { dmin = std::min<int64_t>(dmin, s2.first); dmax = std::max<int64_t>(dmax, s2.first); }
    const int64_t koff = -dmin;
    if ((uint64_t)(dmax + koff + 1) * es >= 0x7FFFFFF0ull || (uint64_t)(n_cols + koff) * es >= 0xFFFFFFF0ull) return MIK_ERR_NOTIMPL;
    const size_t ibytes = 16 + 3 * es, soff0 = 32, voff = soff0 + 128, ioff0 = voff + 32 * es;   // SdiawItem<T>; SdiawPattern<T>: val, items
    const size_t pbytes = ioff0 + 24 * ibytes;
    std::vector<unsigned char> pb(pats.size() * pbytes, 0);
    for (size_t i = 0; i < pats.size(); ++i) {
        unsigned char *o = &pb[i * pbytes];
        const int ns = (int)pats[i].size();
        memcpy(o, &ns, 4);
        for (int q = 0; q < ns; ++q) {
            const int so = (int)((pats[i][(size_t)q].first + koff) * (int64_t)es);
            memcpy(o + soff0 + 4 * q, &so, 4);
            memcpy(o + voff + es * q, &pats[i][(size_t)q].second, es);
        }
        // items of k_spmv_sdiaw2: runs (o - 1, o, o + 1) with o even; a lone even offset is such a run without its outer slots; the
        // list is padded to a multiple of 3 with items without slots.  An odd lone offset, or more than 24 items: the slice is
        // summed slot by slot (nitems = 0).
        struct Item { int off; unsigned b[3]; uint64_t v[3]; };
        std::vector<Item> items;
        bool okp = ns <= 31;                                  // bit 31 stays free: SdiawPattern::fullbits uses it as "never"
        for (int q = 0; q < ns && okp;) {
            const int d = pats[i][(size_t)q].first;
            if (items.size() == 24) { okp = false; break; }
            if (q + 2 < ns && pats[i][(size_t)q + 1].first == d + 1 && pats[i][(size_t)q + 2].first == d + 2 && ((d + 1) & 1) == 0) {
                items.push_back(Item{d + 1, {1u << q, 1u << (q + 1), 1u << (q + 2)},
                                     {pats[i][(size_t)q].second, pats[i][(size_t)q + 1].second, pats[i][(size_t)q + 2].second}});
                q += 3;
            } else if ((d & 1) == 0) {
                items.push_back(Item{d, {0u, 1u << q, 0u}, {0, pats[i][(size_t)q].second, 0}});
                q += 1;
            } else okp = false;
        }
        while (okp && items.size() % 3 != 0 && items.size() < 24) items.push_back(Item{0, {0u, 0u, 0u}, {0, 0, 0}});
        int nitems = (!okp || items.size() % 3 != 0) ? 0 : (int)items.size();
        memcpy(o + 4, &nitems, 4);
        unsigned exa = 0, exc = 0, fullbits = 0;
        bool all_full = nitems > 0;
        for (int k = 0; k < nitems; ++k) {
            unsigned char *it = o + ioff0 + (size_t)k * ibytes;
            const int offb = items[(size_t)k].off * (int)es;
            memcpy(it, &offb, 4);
            all_full = all_full && items[(size_t)k].b[0] && items[(size_t)k].b[1] && items[(size_t)k].b[2];
            fullbits |= items[(size_t)k].b[0] | items[(size_t)k].b[1] | items[(size_t)k].b[2];
            memcpy(it + 4, items[(size_t)k].b, 12);
            for (int c = 0; c < 3; ++c) memcpy(it + 16 + es * (size_t)c, &items[(size_t)k].v[c], es);
            exa |= items[(size_t)k].b[0]; exc |= items[(size_t)k].b[2];
        }
        if (!all_full) fullbits = 0x80000000u;
        memcpy(o + 8, &exa, 4);
        memcpy(o + 12, &exc, 4);
        memcpy(o + 16, &fullbits, 4);
    }
    hipError_t e;
    if ((e = hipMalloc(&A->sdiaw_pats, std::max<size_t>(pb.size(), 8))) != hipSuccess ||
        (e = hipMemcpy(A->sdiaw_pats, pb.data(), pb.size(), hipMemcpyHostToDevice)) != hipSuccess)
        return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: wide slice patterns: %s", hipGetErrorString(e));
    A->sdiaw_npat = (int)pats.size();
    A->sdiaw_koff = (int)koff;
    A->sdiaw_pat_bytes = (int)pbytes;
    return MIK_OK;
}

// {all, any} slot sets per 128-row chunk of the wide layout's row masks (k_sdiaw_chunk_bits): what lets k_spmv_sdiaw2 skip the
// per-lane slot test wherever a wave's rows agree.
static int sdiaw_chunk_bits(mik_ctx *ctx, mik_csr *A)
{
    if (!A->sdiaw_pats || A->sdiaw_uz) return MIK_OK;
    const int64_t nsl = (A->n_rows + MIK_BLOCK - 1) / MIK_BLOCK, nch = (nsl + 1) / 2 * 4;
    hipError_t e = hipMalloc(&A->sdiaw_uz, 16 * (size_t)nch);
    if (e != hipSuccess) return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: chunk bits: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_sdiaw_chunk_bits, dim3((unsigned)((nch + 3) / 4)), dim3(MIK_BLOCK), 0, ctx->stream, (int)A->n_rows, (int)nch, (int)nsl,
                       A->sdiaw_mask, A->sdiaw_pat_id, (const unsigned char *)A->sdiaw_pats, A->sdiaw_pat_bytes, (uint4 *)A->sdiaw_uz);
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create: chunk bits: %s", hipGetErrorString(e));
    return MIK_OK;
}

// Slice-constant values with up to 32 offsets per slice (k_spmv_sdiaw, csrc/mik_sell.h): 9 / 13 / 19 / 27-point constant-coefficient
// stencils.  Built on the host when the <= 8-offset forms do not apply: every 256-row slice uses <= 32 distinct (column - row)
// offsets, every offset carries ONE value (bit pattern) within its slice, every value is finite (an absent slot adds value * 0), and
// row / slot byte offsets fit the 32-bit fields of the buffer instruction.  The first slice that fails ends the attempt.
static int csr_build_sdiaw(mik_ctx *ctx, mik_csr *A, const std::vector<int> &rowptr, const std::vector<int> &col, const std::vector<unsigned char> &v,
                           size_t es, int64_t n_rows, int64_t n_cols, int64_t nnz)
{
    if (A->sdia_val || A->sdia_pats || A->n_long || n_rows <= 0 || nnz <= 0 || n_cols <= 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 1) != 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 4) != 0) return MIK_OK;
    if ((uint64_t)n_rows * es >= 0xFFFFFFF0ull) return MIK_OK;
    const int64_t nb = (n_rows + MIK_BLOCK - 1) / MIK_BLOCK;
    struct Slot { int d; uint64_t bits; };
    auto bits_of = [&](size_t k) { uint64_t b = 0; memcpy(&b, &v[k * es], es); return b; };
    auto finite = [&](uint64_t b) { return es == 8 ? ((b >> 52) & 0x7ff) != 0x7ff : ((b >> 23) & 0xff) != 0xff; };
    std::vector<unsigned> mask((size_t)n_rows, 0u);
    std::vector<int> pat_id((size_t)nb, 0);
    std::vector<std::vector<Slot>> pats;                    // distinct patterns (sorted slots)
    std::map<std::vector<uint64_t>, int> pat_of;            // key: ns, then (d, bits) pairs
    int64_t dmin = 0, dmax = 0;
    std::vector<Slot> sl;
    for (int64_t b = 0; b < nb; ++b) {
        sl.clear();
        const int64_t r0 = b * MIK_BLOCK, r1 = std::min<int64_t>(r0 + MIK_BLOCK, n_rows);
        for (int64_t r = r0; r < r1; ++r)
            for (int k = rowptr[(size_t)r]; k < rowptr[(size_t)r + 1]; ++k) {
                const int d = col[(size_t)k] - (int)r;
                const uint64_t bv = bits_of((size_t)k);
                size_t q = 0;
                while (q < sl.size() && sl[q].d != d) ++q;
                if (q == sl.size()) {
                    if (sl.size() == 32 || !finite(bv)) return MIK_OK;
                    sl.push_back(Slot{d, bv});
                } else if (sl[q].bits != bv) return MIK_OK;          // varying coefficients: not this layout
            }
        std::sort(sl.begin(), sl.end(), [](const Slot &a, const Slot &c) { return a.d < c.d; });
        for (int64_t r = r0; r < r1; ++r) {
            unsigned m = 0;
            size_t q = 0;
            for (int k = rowptr[(size_t)r]; k < rowptr[(size_t)r + 1]; ++k) {       // columns ascend within a row, and so do the slots
                const int d = col[(size_t)k] - (int)r;
                while (q < sl.size() && sl[q].d != d) ++q;
                if (q == sl.size() || (m >> q) & 1u) return MIK_OK;               // unsorted or duplicate columns: not this layout
                m |= 1u << q;
            }
            mask[(size_t)r] = m;
        }
        std::vector<uint64_t> key;
        key.push_back(sl.size());
        for (const Slot &s2 : sl) { key.push_back((uint64_t)(int64_t)s2.d); key.push_back(s2.bits); dmin = std::min<int64_t>(dmin, s2.d); dmax = std::max<int64_t>(dmax, s2.d); }
        auto itp = pat_of.find(key);
        if (itp == pat_of.end()) {
            if (pats.size() >= 65536) return MIK_OK;
            itp = pat_of.emplace(key, (int)pats.size()).first;
            pats.push_back(sl);
        }
        pat_id[(size_t)b] = itp->second;
    }
    std::vector<std::vector<std::pair<int, uint64_t>>> plist(pats.size());
    for (size_t i = 0; i < pats.size(); ++i) for (const Slot &s2 : pats[i]) plist[i].emplace_back(s2.d, s2.bits);
    const int rcf = mik_sdiaw_finish(ctx, A, plist, es, n_cols);
    if (rcf == MIK_ERR_NOTIMPL) return MIK_OK;               // offsets do not fit: not this layout
    if (rcf != MIK_OK) return rcf;
    hipError_t e;
    if ((e = hipMalloc((void **)&A->sdiaw_pat_id, sizeof(int) * (size_t)nb)) != hipSuccess ||
        (e = hipMalloc((void **)&A->sdiaw_mask, sizeof(unsigned) * (size_t)n_rows)) != hipSuccess ||
        (e = hipMemcpy(A->sdiaw_pat_id, pat_id.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(A->sdiaw_mask, mask.data(), sizeof(unsigned) * (size_t)n_rows, hipMemcpyHostToDevice)) != hipSuccess)
        return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: wide slice patterns: %s", hipGetErrorString(e));
    return MIK_OK;
}

// Rows of a 256-row block over its threads by length (k_spmv_rowblock RPERM, csrc/mik_spmv.h): the row with rank p among the block's
// rows (longest first, ties in row order) goes to thread ((p / 64 + block) % 4) * 64 + p % 64 -- the 64 longest rows share a wave, and
// that wave is a different one (a different SIMD) from block to block.  Built for the operators the product tile runs (irregular row
// lengths: split-off long rows or rows beyond 32 entries); MIK_KNOB_LAYOUTS bit 5: never.
int mik_build_rperm_host(mik_ctx *ctx, mik_csr *A, const int *rowptr)
{
    const int64_t n = A->n_rows;
    if (A->sdia_val || A->sdia_pats || A->sdiaw_pats || A->jds_val || n <= 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 32) != 0) return MIK_OK;
    if (A->n_long == 0 && A->max_row_nnz <= 32) return MIK_OK;
    const int64_t nb = (n + MIK_BLOCK - 1) / MIK_BLOCK;
    std::vector<unsigned char> perm((size_t)(nb * MIK_BLOCK));
    int ord[MIK_BLOCK], len[MIK_BLOCK];
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t r0 = b * MIK_BLOCK;
        for (int t = 0; t < MIK_BLOCK; ++t) { ord[t] = t; len[t] = r0 + t < n ? rowptr[r0 + t + 1] - rowptr[r0 + t] : -1; }
        std::stable_sort(ord, ord + MIK_BLOCK, [&](int a, int c) { return len[a] > len[c]; });
        for (int p = 0; p < MIK_BLOCK; ++p) perm[(size_t)(r0 + (((p >> 6) + (int)(b & 3)) & 3) * 64 + (p & 63))] = (unsigned char)ord[p];
    }
    hipError_t e;
    (void)hipSetDevice(ctx->device);
    if ((e = hipMalloc((void **)&A->rperm, perm.size())) != hipSuccess ||
        (e = hipMemcpy(A->rperm, perm.data(), perm.size(), hipMemcpyHostToDevice)) != hipSuccess)
        return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: row permutation: %s", hipGetErrorString(e));
    return MIK_OK;
}

// Windows of x for the product-tile kernel (k_spmv_rowblock XWIN, csrc/mik_spmv.h): for every 256-row block the first column its
// SHORT rows reference, aligned down to 16 bytes; one common span = the widest block's, rounded up to whole 1-KiB LDS-DMA pieces.
// Built for irregular operators (split-off long rows or rows beyond 32 entries: the uniform short-row operators stay with the
// LDS-DMA tile of k_spmv_rowgather) when the blocks that span at most 32 KB of x hold three quarters of the entries and x is at least
// one span long; a wider block (rows that wrap around the matrix) is marked -1 and gathers from memory; a window that would leave x at
// the end of the vector slides down.  MIK_KNOB_LAYOUTS bit 5 = never (read here), bit 6 = not used at launch.
static int csr_build_xwin(mik_ctx *ctx, mik_csr *A, const std::vector<int> &rowptr, const std::vector<int> &col, size_t es, int64_t n_rows, int64_t n_cols,
                          int max_row)
{
    if (A->sdia_val || A->sdia_pats || A->sdiaw_pats || A->jds_val || n_rows <= 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 32) != 0) return MIK_OK;
    if (A->n_long == 0 && max_row <= 32) return MIK_OK;
    const int64_t nb = (n_rows + MIK_BLOCK - 1) / MIK_BLOCK;
    std::vector<int> mn((size_t)nb, INT32_MAX), mx((size_t)nb, -1), cnt((size_t)nb, 0), lo;
    for (int64_t b = 0; b < nb; ++b) {
        const int ka = rowptr[(size_t)(b * MIK_BLOCK)], kb = rowptr[(size_t)std::min<int64_t>((b + 1) * MIK_BLOCK, n_rows)];
        cnt[(size_t)b] = kb - ka;
        for (int k = ka; k < kb; ++k) { mn[(size_t)b] = std::min(mn[(size_t)b], col[(size_t)k]); mx[(size_t)b] = std::max(mx[(size_t)b], col[(size_t)k]); }
    }
    int span = 0;
    if (!mik_xwin_plan(nb, mn.data(), mx.data(), cnt.data(), es, n_cols, (int64_t)rowptr[(size_t)n_rows], lo, &span)) return MIK_OK;
    hipError_t e;
    (void)hipSetDevice(ctx->device);
    if ((e = hipMalloc((void **)&A->xwin_lo, sizeof(int) * (size_t)nb)) != hipSuccess ||
        (e = hipMemcpy(A->xwin_lo, lo.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice)) != hipSuccess)
        return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: window table: %s", hipGetErrorString(e));
    A->xwin_span = span;
    return MIK_OK;
}

// Jagged slices (csrc/mik_jds.h) from the SHORT part of the CSR arrays (split-off long rows have no entries there; `is_long`
// marks them).  Built when no structured layout applies, every wave's lanes stay busy in the natural row order (wave
// iterations within 25 % of the ideal: near-uniform rows) and either the rows are long (more than 32 entries somewhere) or the
// group padding costs < 10 % of the CSR bytes.  Short rows with padding stay with the LDS-DMA tile, uneven rows with the
// product tile (mik_spmv.h).
static int csr_build_jds(mik_ctx *ctx, mik_csr *A, const std::vector<int> &rowptr, const std::vector<int> &col, const std::vector<unsigned char> &v,
                         size_t es, int64_t n_rows, const unsigned char *is_long)
{
    if (A->sdia_val || A->sdia_pats || A->sdiaw_pats || n_rows <= 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 1) != 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 8) != 0) return MIK_OK;   // development knob 28: 1 = never, 2 = whenever possible
    const int W = (int)(16 / es);
    const int64_t short_nnz = rowptr[(size_t)n_rows];
    if (short_nnz <= 0) return MIK_OK;
    auto glen = [&](int64_t r) { return (rowptr[(size_t)r + 1] - rowptr[(size_t)r] + W - 1) / W; };
    const int64_t nsl = (n_rows + 63) / 64;
    int64_t groups = 0, iters = 0;                              // iters: every slice runs as long as its longest row
    int maxlen = 0;
    for (int64_t s = 0; s < nsl; ++s) {
        int m = 0;
        for (int64_t r = s * 64; r < std::min<int64_t>(s * 64 + 64, n_rows); ++r) {
            groups += glen(r);
            m = std::max(m, glen(r));
            maxlen = std::max(maxlen, rowptr[(size_t)r + 1] - rowptr[(size_t)r]);
        }
        iters += m;
    }
    if (maxlen >= MIK_JDS_LONG || groups * W >= INT32_MAX) return MIK_OK;
    const int64_t jds_bytes = groups * W * (int64_t)(es + 4) + 2 * n_rows, csr_bytes = short_nnz * (int64_t)(es + 4) + 4 * n_rows;
    if ((ctx->tuning[MIK_KNOB_LAYOUTS] & 16) == 0 && !(iters * 64 * 4 <= groups * 5 && (maxlen > 32 || jds_bytes * 10 <= csr_bytes * 11))) return MIK_OK;
    std::vector<int> jptr, jcol;
    std::vector<unsigned short> jlen;
    std::vector<unsigned char> jval;
    try {
        jptr.assign((size_t)nsl + 8, 0);
        jlen.assign((size_t)n_rows, 0);
        jcol.assign((size_t)groups * W, 0);
        jval.assign((size_t)groups * W * es, 0);
    } catch (const std::bad_alloc &) {
        return mik_fail(ctx, MIK_ERR_NOMEM, "mik_csr_create: host staging allocation failed (jagged slices)");
    }
    int64_t cur = 0;
    for (int64_t s = 0; s < nsl; ++s) {
        jptr[(size_t)s] = (int)cur;
        const int64_t p0 = s * 64, p1 = std::min<int64_t>(p0 + 64, n_rows);
        int m = 0;
        for (int64_t r = p0; r < p1; ++r) m = std::max(m, glen(r));
        for (int g = 0; g < m; ++g)
            for (int64_t r = p0; r < p1; ++r) {
                const int k0 = rowptr[(size_t)r], len = rowptr[(size_t)r + 1] - k0;
                if (g * W >= len) continue;
                for (int e = 0; e < W; ++e) {
                    const int j = g * W + e;
                    const size_t dst = (size_t)cur * W + e;
                    if (j < len) {
                        jcol[dst] = col[(size_t)k0 + j];
                        memcpy(&jval[dst * es], &v[((size_t)k0 + j) * es], es);
                    } else {
                        jcol[dst] = col[(size_t)k0 + len - 1];           // padding: the row's last real column, value 0, never added
                    }
                }
                ++cur;
            }
    }
    for (size_t q = (size_t)nsl; q < jptr.size(); ++q) jptr[q] = (int)cur;     // the waves of the last workgroup beyond the last slice
    for (int64_t r = 0; r < n_rows; ++r)
        jlen[(size_t)r] = (is_long && is_long[r]) ? (unsigned short)MIK_JDS_LONG : (unsigned short)(rowptr[(size_t)r + 1] - rowptr[(size_t)r]);
    hipError_t e;
    const size_t pad = 64 * MIK_JDS_U * (size_t)W;         // lanes without a group read (and gather through) the tail: zero columns
    if ((e = hipMalloc((void **)&A->jds_ptr, sizeof(int) * jptr.size())) != hipSuccess ||
        (e = hipMalloc((void **)&A->jds_len, sizeof(unsigned short) * (size_t)n_rows)) != hipSuccess ||
        (e = hipMalloc((void **)&A->jds_col, sizeof(int) * ((size_t)groups * W + pad))) != hipSuccess ||
        (e = hipMalloc(&A->jds_val, es * ((size_t)groups * W + pad))) != hipSuccess ||
        (e = hipMemcpy(A->jds_ptr, jptr.data(), sizeof(int) * jptr.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(A->jds_len, jlen.data(), sizeof(unsigned short) * (size_t)n_rows, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemset(A->jds_col + (size_t)groups * W, 0, sizeof(int) * pad)) != hipSuccess ||
        (e = hipMemset((unsigned char *)A->jds_val + es * (size_t)groups * W, 0, es * pad)) != hipSuccess ||
        (e = hipMemcpy(A->jds_col, jcol.data(), sizeof(int) * (size_t)groups * W, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(A->jds_val, jval.data(), es * (size_t)groups * W, hipMemcpyHostToDevice)) != hipSuccess)
        return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: jagged slices: %s", hipGetErrorString(e));
    A->jds_groups = groups;
    A->jds_short_nnz = short_nnz;
    return MIK_OK;
}

// ---------------------------------------------------------------------------------------------
// operator upload
// ---------------------------------------------------------------------------------------------
static bool is_device_pointer(const void *p)
{
    hipPointerAttribute_t attr;
    if (!p || hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain host memory: "invalid value"
    return attr.type == hipMemoryTypeDevice;
}

static int csr_create_impl(mik_ctx *ctx, int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *ptr, const int64_t *idx,
                           const void *val, int index_base, int is_csc, bool on_device, mik_csr **out);

extern "C" int mik_csr_create(mik_ctx *ctx, int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz,
                              const int64_t *ptr, const int64_t *idx, const void *val, int index_base,
                              int is_csc, mik_csr **out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    if (dtype != MIK_F64 && dtype != MIK_F32) return mik_fail(ctx, MIK_ERR_INVALID, "mik_csr_create: bad dtype %d", dtype);
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || !ptr || (nnz && (!idx || !val)))
        return mik_fail(ctx, MIK_ERR_INVALID, "mik_csr_create: bad sizes or NULL arrays");
    if (n_rows >= (int64_t)INT32_MAX - MIK_BLOCK || n_cols >= INT32_MAX || nnz >= (int64_t)INT32_MAX - 2 * MIK_SPMV_TILE)
        return mik_fail(ctx, MIK_ERR_NOTIMPL, "mik_csr_create: sizes exceed the Int32 device index range");
    // The three arrays may also live in device memory (a ROCSparseMatrixCSC, a torch tensor): the upload pipeline then starts
    // from them without a host copy.  Mixed placement is refused.
    (void)hipSetDevice(ctx->device);
    const bool dev_in = is_device_pointer(ptr);
    if (nnz && (is_device_pointer(idx) != dev_in || is_device_pointer(val) != dev_in))
        return mik_fail(ctx, MIK_ERR_INVALID, "mik_csr_create: ptr / idx / val must all be host arrays or all device arrays");
    if (!dev_in) return csr_create_impl(ctx, dtype, n_rows, n_cols, nnz, ptr, idx, val, index_base, is_csc, false, out);
    int rc = csr_create_impl(ctx, dtype, n_rows, n_cols, nnz, ptr, idx, val, index_base, is_csc, true, out);
    if (rc != MIK_ERR_NOTIMPL) return rc;
    // the host path's matrices (long rows, duplicates, MIK_KNOB_UPLOAD): stage the arrays on the host once
    const int64_t n_major = is_csc ? n_cols : n_rows;
    const size_t es = mik_dtype_size(dtype);
    std::vector<int64_t> hp, hi;
    std::vector<unsigned char> hv;
    try {
        hp.resize((size_t)n_major + 1);
        hi.resize((size_t)nnz);
        hv.resize((size_t)nnz * es);
    } catch (const std::bad_alloc &) {
        return mik_fail(ctx, MIK_ERR_NOMEM, "mik_csr_create: host staging allocation failed");
    }
    if (hipMemcpy(hp.data(), ptr, sizeof(int64_t) * hp.size(), hipMemcpyDeviceToHost) != hipSuccess ||
        (nnz && (hipMemcpy(hi.data(), idx, sizeof(int64_t) * hi.size(), hipMemcpyDeviceToHost) != hipSuccess ||
                 hipMemcpy(hv.data(), val, hv.size(), hipMemcpyDeviceToHost) != hipSuccess)))
        return mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create: reading the device arrays failed");
    return csr_create_impl(ctx, dtype, n_rows, n_cols, nnz, hp.data(), hi.data(), hv.data(), index_base, is_csc, false, out);
}

// SparseMatrixCSC{T, Int32}: widen the index arrays on the host, then the Int64 entry (the device pipeline starts from host arrays).
extern "C" int mik_csr_create_i32(mik_ctx *ctx, int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *ptr, const int32_t *idx,
                                  const void *val, int index_base, int is_csc, mik_csr **out)
{
    if (!ctx || !out) return MIK_ERR_INVALID;
    *out = nullptr;
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || !ptr || (nnz && (!idx || !val))) return mik_fail(ctx, MIK_ERR_INVALID, "mik_csr_create_i32: bad sizes or NULL arrays");
    if (dtype != MIK_F64 && dtype != MIK_F32) return mik_fail(ctx, MIK_ERR_INVALID, "mik_csr_create_i32: bad dtype %d", dtype);
    const int64_t n_major = is_csc ? n_cols : n_rows;
    (void)hipSetDevice(ctx->device);
    const bool dev_in = is_device_pointer(ptr);
    if (nnz && (is_device_pointer(idx) != dev_in || is_device_pointer(val) != dev_in))
        return mik_fail(ctx, MIK_ERR_INVALID, "mik_csr_create_i32: ptr / idx / val must all be host arrays or all device arrays");
    std::vector<int64_t> p64, i64;
    std::vector<int32_t> p32, i32;
    std::vector<unsigned char> hv;
    try {
        p64.resize((size_t)n_major + 1);
        i64.resize((size_t)nnz);
        if (dev_in) { p32.resize((size_t)n_major + 1); i32.resize((size_t)nnz); hv.resize((size_t)nnz * mik_dtype_size(dtype)); }
    } catch (const std::bad_alloc &) {
        return mik_fail(ctx, MIK_ERR_NOMEM, "mik_csr_create_i32: host staging allocation failed");
    }
    if (dev_in) {
        if (hipMemcpy(p32.data(), ptr, sizeof(int32_t) * p32.size(), hipMemcpyDeviceToHost) != hipSuccess ||
            (nnz && (hipMemcpy(i32.data(), idx, sizeof(int32_t) * i32.size(), hipMemcpyDeviceToHost) != hipSuccess ||
                     hipMemcpy(hv.data(), val, hv.size(), hipMemcpyDeviceToHost) != hipSuccess)))
            return mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create_i32: reading the device arrays failed");
        ptr = p32.data(); idx = i32.data(); val = hv.data();
    }
    for (int64_t j = 0; j <= n_major; ++j) p64[(size_t)j] = ptr[j];
    for (int64_t k = 0; k < nnz; ++k) i64[(size_t)k] = idx[k];
    return mik_csr_create(ctx, dtype, n_rows, n_cols, nnz, p64.data(), i64.data(), val, index_base, is_csc, out);
}

// on_device: ptr / idx / val are device arrays; only the device pipeline can consume them (MIK_ERR_NOTIMPL otherwise)
static int csr_create_impl(mik_ctx *ctx, int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *ptr, const int64_t *idx,
                           const void *val, int index_base, int is_csc, bool on_device, mik_csr **out)
{
    const int64_t n_major = is_csc ? n_cols : n_rows;   // length of ptr - 1
    const int64_t n_minor = is_csc ? n_rows : n_cols;   // range of idx
    int64_t ends[2] = {0, 0};
    if (on_device) {
        if (hipMemcpy(&ends[0], ptr, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&ends[1], ptr + n_major, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess)
            return mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create: reading ptr from the device failed");
    } else {
        ends[0] = ptr[0];
        ends[1] = ptr[n_major];
    }
    if (ends[0] != index_base || ends[1] - index_base != nnz)
        return mik_fail(ctx, MIK_ERR_MISMATCH, "mik_csr_create: ptr[0]=%lld, ptr[end]=%lld inconsistent with nnz=%lld, base=%d",
                        (long long)ends[0], (long long)ends[1], (long long)nnz, index_base);
    const size_t es = mik_dtype_size(dtype);
    std::vector<int> rowptr, col;
    std::vector<unsigned char> v;
    // Default: everything past the raw host-to-device copy happens on the device (mik_upload.hip).  MIK_ERR_NOTIMPL from it
    // = a matrix the host path below handles (long rows, duplicate entries, no room for the raw copy); MIK_KNOB_UPLOAD:
    // 1 = host path only.
    if (ctx->tuning[MIK_KNOB_UPLOAD] != 1 && nnz > 0 && n_rows > 0 && n_cols > 0 && n_rows < 0x7f000000) {       // (row ids below the "no row yet" pattern of the analysis)
        mik_csr *A = new (std::nothrow) mik_csr();
        if (!A) return mik_fail(ctx, MIK_ERR_NOMEM, "mik_csr_create: host allocation failed");
        A->ctx = ctx; A->dtype = dtype; A->n_rows = n_rows; A->n_cols = n_cols; A->nnz = nnz;
        (void)hipSetDevice(ctx->device);
        int rc = mik_upload_device(ctx, A, dtype, n_rows, n_cols, nnz, ptr, idx, val, index_base, is_csc);
        // no <= 8-offset layout: the wide slice-constant form, then the jagged slices, both built on the device from A's CSR arrays
        // (MIK_KNOB_UPLOAD = 2: through the host builders, from a copy of the device CSR)
        if (rc == MIK_OK && !A->sdia_val && !A->sdia_pats && (ctx->tuning[MIK_KNOB_LAYOUTS] & 1) == 0 && ctx->tuning[MIK_KNOB_UPLOAD] != 2) {
            rc = mik_build_sdiaw_device(ctx, A);
            if (rc == MIK_OK) rc = mik_build_jds_device(ctx, A);
            if (rc == MIK_OK) rc = mik_build_xwin_device(ctx, A);
        } else if (rc == MIK_OK && !A->sdia_val && !A->sdia_pats && (ctx->tuning[MIK_KNOB_LAYOUTS] & 1) == 0) {
            try {
                rowptr.resize((size_t)n_rows + 1);
                col.resize((size_t)nnz);
                v.resize((size_t)nnz * es);
            } catch (const std::bad_alloc &) {
                mik_csr_destroy(A);
                return mik_fail(ctx, MIK_ERR_NOMEM, "mik_csr_create: host staging allocation failed");
            }
            if (hipMemcpy(rowptr.data(), A->rowptr, sizeof(int) * ((size_t)n_rows + 1), hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(col.data(), A->col, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(v.data(), A->val, es * (size_t)nnz, hipMemcpyDeviceToHost) != hipSuccess) {
                mik_csr_destroy(A);
                return mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create: reading the device CSR back failed");
            }
            rc = csr_build_sdiaw(ctx, A, rowptr, col, v, es, n_rows, n_cols, nnz);
            if (rc == MIK_OK) rc = csr_build_jds(ctx, A, rowptr, col, v, es, n_rows, nullptr);
            if (rc == MIK_OK) rc = csr_build_xwin(ctx, A, rowptr, col, es, n_rows, n_cols, A->max_row_nnz);
            if (rc == MIK_OK) rc = mik_build_rperm_host(ctx, A, rowptr.data());
        }
        if (rc == MIK_OK) rc = sdiaw_chunk_bits(ctx, A);
        if (rc == MIK_OK) { *out = A; return MIK_OK; }
        mik_csr_destroy(A);
        if (rc != MIK_ERR_NOTIMPL) return rc;
    }
    if (on_device) return MIK_ERR_NOTIMPL;             // the caller stages the arrays on the host and comes back
    try {
        rowptr.assign((size_t)n_rows + 1, 0);
        col.resize((size_t)nnz);
        v.resize((size_t)nnz * es);
    } catch (const std::bad_alloc &) {
        return mik_fail(ctx, MIK_ERR_NOMEM, "mik_csr_create: host staging allocation failed");
    }
    for (int64_t j = 0; j < n_major; ++j)
        if (ptr[j + 1] < ptr[j]) return mik_fail(ctx, MIK_ERR_INVALID, "mik_csr_create: ptr not monotone at %lld", (long long)j);
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t i = idx[k] - index_base;
        if (i < 0 || i >= n_minor) return mik_fail(ctx, MIK_ERR_INVALID, "mik_csr_create: index %lld out of range at %lld", (long long)idx[k], (long long)k);
    }
    int max_row = 0;
    if (is_csc) {
        // CSC -> CSR: counting-sort transpose; walking columns in ascending order keeps every row's
        // entries in ascending column order = the order Julia's column scatter reaches that row.
        for (int64_t k = 0; k < nnz; ++k) rowptr[(size_t)(idx[k] - index_base) + 1]++;
        for (int64_t r = 0; r < n_rows; ++r) {
            max_row = std::max(max_row, rowptr[r + 1]);
            rowptr[r + 1] += rowptr[r];
        }
        std::vector<int> cursor(rowptr.begin(), rowptr.end() - 1);
        for (int64_t j = 0; j < n_cols; ++j)
            for (int64_t k = ptr[j] - index_base; k < ptr[j + 1] - index_base; ++k) {
                const int64_t r = idx[k] - index_base;
                const int dst = cursor[r]++;
                col[dst] = (int)j;
                memcpy(&v[(size_t)dst * es], (const unsigned char *)val + (size_t)k * es, es);
            }
    } else {
        for (int64_t r = 0; r <= n_rows; ++r) rowptr[r] = (int)(ptr[r] - index_base);
        for (int64_t r = 0; r < n_rows; ++r) max_row = std::max(max_row, rowptr[r + 1] - rowptr[r]);
        for (int64_t k = 0; k < nnz; ++k) col[k] = (int)(idx[k] - index_base);
        if (nnz) memcpy(v.data(), val, (size_t)nnz * es);
    }

    // Banded operators (stencils): distance, in 256-row blocks, between a row and its farthest in-block
    // neighbour row, rounded up to a multiple of 8 -> the "strips" workgroup map of spmv_block_map, which
    // keeps a row-block, its line neighbours and its plane neighbours in ONE XCD's L2 (x is then fetched from
    // the fabric about once instead of three times).  Columns >= n_rows are halo entries of a row partition.
    int strip = 0;
    {
        int64_t bw = 0;
        for (int64_t r = 0; r < n_rows; ++r)
            for (int k = rowptr[r]; k < rowptr[r + 1]; ++k)
                if (col[k] < n_rows) bw = std::max<int64_t>(bw, std::llabs((long long)col[k] - (long long)r));
        strip = mik_strip_for(ctx, bw, (n_rows + MIK_BLOCK - 1) / MIK_BLOCK);
    }

    // Long rows (> MIK_LONG_ROW entries) leave the row-block layout: their entries move behind all
    // short-row entries and a wave-per-row kernel sums them (still serially, in column order).  In the
    // short part a long row becomes an empty row, so the row-block kernel keeps its contiguous ranges.
    std::vector<int> long_rows, long_start, long_len;          // virtual rows of the long part: whole rows and segments of cut rows
    std::vector<int> seg_row, cut_row, cut_first, cut_nseg;    // segment -> cut row; cut row -> matrix row, first segment, segments
    std::vector<unsigned char> is_long;
    // development knob [4]: > 0 overrides the threshold, < 0 disables the split
    const int long_row = MIK_LONG_ROW;
    if (max_row > long_row) {
        is_long.assign((size_t)n_rows, 0);
        std::vector<int> rp2((size_t)n_rows + 1, 0), col2;
        std::vector<unsigned char> v2;
        int64_t short_nnz = 0, long_store = 0;
        for (int64_t r = 0; r < n_rows; ++r) {
            const int64_t len = rowptr[r + 1] - rowptr[r];
            if (len <= long_row) short_nnz += len; else long_store += (len + 3) & ~(int64_t)3;
        }
        // the long part starts 16-byte aligned and EVERY long row does (padded to a multiple of 4 entries with {column 0, value 0}:
        // a lane reads its group of 4 with one 16-byte load, spmv_longrow_wave; the padding is gathered, multiplied and never added)
        int64_t ps = 0, pl = (short_nnz + 3) & ~(int64_t)3;
        if (pl + long_store > INT32_MAX) return mik_fail(ctx, MIK_ERR_NOTIMPL, "mik_csr_create: more than 2^31 stored entries");
        col2.assign((size_t)(pl + long_store), 0);
        v2.assign(col2.size() * es, 0);
        for (int64_t r = 0; r < n_rows; ++r) {
            const int len = rowptr[r + 1] - rowptr[r];
            rp2[r] = (int)ps;
            if (len <= long_row) {
                if (len) { memcpy(&col2[ps], &col[rowptr[r]], sizeof(int) * len); memcpy(&v2[(size_t)ps * es], &v[(size_t)rowptr[r] * es], es * len); }
                ps += len;
            } else {
                is_long[r] = 1;
                long_rows.push_back((int)r); long_start.push_back((int)pl); long_len.push_back(len);
                memcpy(&col2[pl], &col[rowptr[r]], sizeof(int) * len);
                memcpy(&v2[(size_t)pl * es], &v[(size_t)rowptr[r] * es], es * len);
                pl += (len + 3) & ~3;
            }
        }
        rp2[n_rows] = (int)ps;
        rowptr.swap(rp2); col.swap(col2); v.swap(v2);
        {   // rows longer than one segment are cut: every segment becomes a virtual row of its own (csrc/mik_spmv.h, LongTab); the
            // list stays in (row, segment) order -- the four waves of a workgroup then walk neighbouring pieces of one row
            const int seg = ((ctx->tuning[MIK_KNOB_LONG_SEGMENT] > 0 ? ctx->tuning[MIK_KNOB_LONG_SEGMENT] : MIK_LONG_SEG) + 3) & ~3;    // development knob: segment length (whole groups of 4)
            std::vector<int> vr, vs, vl;
            for (size_t q = 0; q < long_rows.size(); ++q) {
                if (long_len[q] <= seg) { vr.push_back(long_rows[q]); vs.push_back(long_start[q]); vl.push_back(long_len[q]); continue; }
                const int ns = (long_len[q] + seg - 1) / seg;
                cut_row.push_back(long_rows[q]); cut_first.push_back((int)seg_row.size()); cut_nseg.push_back(ns);
                for (int z = 0; z < ns; ++z) {
                    vr.push_back(-((int)seg_row.size() + 1));
                    vs.push_back(long_start[q] + z * seg);
                    vl.push_back(std::min(seg, long_len[q] - z * seg));
                    seg_row.push_back((int)cut_row.size() - 1);
                }
            }
            // List order = order of the waves (4 per workgroup): by the FIRST COLUMN of the virtual row.  The gathers of a long row are
            // what its time goes into -- a 128-byte line of x fetched from L2 for the ~4 entries the row has in it, 400 MB of L2 -> L1
            // traffic for the 100 MB of operator streams of the banded configs[4] stand-in, and that traffic (not the texture path,
            // not HBM) is the bound: the long part ran at the same ~10 TB/s of L2 -> L1 bandwidth alone and merged into the row-block
            // launch (profiles/r04_c5_banded_*).  Long rows that are neighbours in the matrix reference almost the same columns; with the
            // list sorted by first column the four waves of a workgroup walk the SAME 32 KB of x side by side, and three of them hit
            // in L1.  (Sorting is a pure scheduling matter: every virtual row keeps its shape, the segment sums their order.)
            std::vector<int> ord(vr.size());
            for (size_t q = 0; q < ord.size(); ++q) ord[q] = (int)q;
                    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return col[(size_t)vs[(size_t)a]] < col[(size_t)vs[(size_t)b]]; });
            long_rows.resize(vr.size()); long_start.resize(vr.size()); long_len.resize(vr.size());
            for (size_t q = 0; q < ord.size(); ++q) { long_rows[q] = vr[(size_t)ord[q]]; long_start[q] = vs[(size_t)ord[q]]; long_len[q] = vl[(size_t)ord[q]]; }
        }
    }
    const int64_t nnz_store = (int64_t)col.size();                     // entries physically stored (>= nnz when re-laid out)

    int max_rb = 0;
    for (int64_t r0 = 0; r0 < n_rows; r0 += MIK_BLOCK)
        max_rb = std::max(max_rb, rowptr[std::min<int64_t>(r0 + MIK_BLOCK, n_rows)] - rowptr[r0]);

    mik_csr *A = new (std::nothrow) mik_csr();
    if (!A) return mik_fail(ctx, MIK_ERR_NOMEM, "mik_csr_create: host allocation failed");
    A->max_rowblock_nnz = max_rb;
    A->n_long = (int)long_rows.size();
    A->ctx = ctx; A->dtype = dtype; A->n_rows = n_rows; A->n_cols = n_cols; A->nnz = nnz; A->max_row_nnz = max_row; A->strip = strip;
    const size_t pad = 2 * MIK_SPMV_TILE;   // slack so tile-granular reads never leave the allocation
    auto cleanup = [&]() { mik_csr_destroy(A); };
    hipError_t e;
    (void)hipSetDevice(ctx->device);
    const size_t ns = (size_t)nnz_store;
    if ((e = hipMalloc((void **)&A->rowptr, sizeof(int) * ((size_t)n_rows + 1 + 256))) != hipSuccess ||
        (e = hipMalloc((void **)&A->col, sizeof(int) * (ns + pad))) != hipSuccess ||
        (e = hipMalloc(&A->val, es * (ns + pad))) != hipSuccess) {
        cleanup();
        return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: hipMalloc: %s", hipGetErrorString(e));
    }
    if ((e = hipMemsetAsync(A->col + ns, 0, sizeof(int) * pad, ctx->stream)) != hipSuccess ||
        (e = hipMemsetAsync((unsigned char *)A->val + es * ns, 0, es * pad, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(A->rowptr, rowptr.data(), sizeof(int) * ((size_t)n_rows + 1), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (ns && (e = hipMemcpyAsync(A->col, col.data(), sizeof(int) * ns, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) ||
        (ns && (e = hipMemcpyAsync(A->val, v.data(), es * ns, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) ||
        (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) {
        cleanup();
        return mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create: upload: %s", hipGetErrorString(e));
    }
    if (A->n_long) {
        const size_t nl = (size_t)A->n_long, nsg = seg_row.size(), nc = cut_row.size();
        A->n_seg = (int)nsg; A->n_cut = (int)nc;
        std::vector<int> tab;
        tab.insert(tab.end(), long_rows.begin(), long_rows.end());
        tab.insert(tab.end(), long_start.begin(), long_start.end());
        tab.insert(tab.end(), long_len.begin(), long_len.end());
        tab.insert(tab.end(), seg_row.begin(), seg_row.end());
        tab.insert(tab.end(), cut_row.begin(), cut_row.end());
        tab.insert(tab.end(), cut_first.begin(), cut_first.end());
        tab.insert(tab.end(), cut_nseg.begin(), cut_nseg.end());
        tab.insert(tab.end(), nc, 0);                                          // tickets
        if ((e = hipMalloc((void **)&A->long_rows, sizeof(int) * std::max<size_t>(tab.size(), 1))) != hipSuccess ||
            (e = hipMalloc((void **)&A->is_long, (size_t)n_rows)) != hipSuccess ||
            (e = hipMalloc(&A->seg_sum, es * std::max<size_t>(nsg, 1))) != hipSuccess ||
            (e = hipMemcpy(A->long_rows, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(A->is_long, is_long.data(), (size_t)n_rows, hipMemcpyHostToDevice)) != hipSuccess) {
            cleanup();
            return mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create: long-row tables: %s", hipGetErrorString(e));
        }
        (void)nl;
    }
    // device layouts for banded / stencil operators (see the two builders above)
    int rc_layout = csr_build_sdia(ctx, A, rowptr, col, v, es, n_rows, n_cols, nnz, max_row);
    if (rc_layout == MIK_OK) rc_layout = csr_build_sdiaw(ctx, A, rowptr, col, v, es, n_rows, n_cols, nnz);
    if (rc_layout == MIK_OK) rc_layout = csr_build_jds(ctx, A, rowptr, col, v, es, n_rows, is_long.empty() ? nullptr : is_long.data());
    if (rc_layout == MIK_OK) rc_layout = csr_build_xwin(ctx, A, rowptr, col, es, n_rows, n_cols, max_row);
    if (rc_layout == MIK_OK) rc_layout = mik_build_rperm_host(ctx, A, rowptr.data());
    if (rc_layout == MIK_OK && A->n_long && (ctx->tuning[MIK_KNOB_LAYOUTS] & 32) == 0 && !A->jds_val) {
        // windows of x for the long-row workgroups (csrc/mik_spmv.h, spmv_long_window): workgroup g sums virtual rows 4 g .. 4 g + 3 of the
        // list (sorted by first column); its window starts at their smallest first column and is as long as the LDS of a row-block
        // workgroup ([product tile][wave sums][x window of the short rows]); -1 where less than about half of the workgroup's columns fit
        const int TILE = MIK_SPMV_TILE * (int)(8 / es), XP = (int)(1024 / es), W = (int)(16 / es);
        const int64_t LW = (int64_t)((size_t)(TILE + 12 + A->xwin_span) * es / 1024) * XP;
        if (LW > 0 && n_cols >= LW + W) {
            const size_t nwg = (long_rows.size() + 3) / 4;
            std::vector<int> lwin(nwg, -1);
            size_t with_window = 0;
            for (size_t g2 = 0; g2 < nwg; ++g2) {
                int64_t lo = INT64_MAX, hi = -1;
                for (size_t q = 4 * g2; q < std::min(4 * g2 + 4, long_rows.size()); ++q) {
                    lo = std::min<int64_t>(lo, col[(size_t)long_start[q]]);
                    hi = std::max<int64_t>(hi, col[(size_t)long_start[q] + (size_t)long_len[q] - 1]);
                }
                lo &= ~(int64_t)(W - 1);
                if (lo + LW > n_cols) lo = (n_cols - LW) & ~(int64_t)(W - 1);
                if (hi - lo < 2 * LW) { lwin[g2] = (int)lo; ++with_window; }
            }
            A->long_spread = 2 * with_window >= nwg;
            if (!A->long_spread) {
                // the long rows of this operator share no columns worth a window (uniformly random columns): no windows, and the list goes
                // back to LONGEST FIRST -- the order that starts the longest chains earliest (the random configs[4] stand-in: 186 us that
                // way, 191 us in first-column order)
                std::fill(lwin.begin(), lwin.end(), -1);
                std::vector<int> ord(long_rows.size());
                for (size_t q = 0; q < ord.size(); ++q) ord[q] = (int)q;
                std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return long_len[(size_t)a] > long_len[(size_t)b]; });
                std::vector<int> tab3;
                for (int q : ord) tab3.push_back(long_rows[(size_t)q]);
                for (int q : ord) tab3.push_back(long_start[(size_t)q]);
                for (int q : ord) tab3.push_back(long_len[(size_t)q]);
                if (hipMemcpy(A->long_rows, tab3.data(), sizeof(int) * tab3.size(), hipMemcpyHostToDevice) != hipSuccess) {
                    cleanup();
                    return mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create: long-row table");
                }
            }
            hipError_t e2;
            if ((e2 = hipMalloc((void **)&A->long_win, sizeof(int) * nwg)) != hipSuccess ||
                (e2 = hipMemcpy(A->long_win, lwin.data(), sizeof(int) * nwg, hipMemcpyHostToDevice)) != hipSuccess) {
                cleanup();
                return mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create: long-row windows: %s", hipGetErrorString(e2));
            }
            A->long_lw = (int)LW;
        }
    }
    if (rc_layout == MIK_OK) rc_layout = sdiaw_chunk_bits(ctx, A);
    if (rc_layout != MIK_OK) { cleanup(); return rc_layout; }
    *out = A;
    return MIK_OK;
}

extern "C" int mik_csr_destroy(mik_csr *A)
{
    if (!A) return MIK_OK;
    if (A->ctx) (void)hipStreamSynchronize(A->ctx->stream);
    if (A->rowptr) (void)hipFree(A->rowptr);
    if (A->col) (void)hipFree(A->col);
    if (A->val) (void)hipFree(A->val);
    if (A->long_rows) (void)hipFree(A->long_rows);
    if (A->is_long) (void)hipFree(A->is_long);
    if (A->seg_sum) (void)hipFree(A->seg_sum);
    if (A->sdia_ptr) (void)hipFree(A->sdia_ptr);
    if (A->sdia_off) (void)hipFree(A->sdia_off);
    if (A->sdia_tri) (void)hipFree(A->sdia_tri);
    if (A->sdia_mask) (void)hipFree(A->sdia_mask);
    if (A->sdia_val) (void)hipFree(A->sdia_val);
    if (A->sdia_pats) (void)hipFree(A->sdia_pats);
    if (A->sdia_pat_id) (void)hipFree(A->sdia_pat_id);
    if (A->sdia_recs) (void)hipFree(A->sdia_recs);
    if (A->sdiaw_pats) (void)hipFree(A->sdiaw_pats);
    if (A->sdiaw_pat_id) (void)hipFree(A->sdiaw_pat_id);
    if (A->sdiaw_mask) (void)hipFree(A->sdiaw_mask);
    if (A->sdiaw_uz) (void)hipFree(A->sdiaw_uz);
    if (A->xwin_lo) (void)hipFree(A->xwin_lo);
    if (A->rperm) (void)hipFree(A->rperm);
    if (A->long_win) (void)hipFree(A->long_win);
    if (A->jds_ptr) (void)hipFree(A->jds_ptr);
    if (A->jds_len) (void)hipFree(A->jds_len);
    if (A->jds_col) (void)hipFree(A->jds_col);
    if (A->jds_val) (void)hipFree(A->jds_val);
    delete A;
    return MIK_OK;
}

static int spmv_kernel_choice(const mik_csr *A);
extern "C" int mik_csr_layout(const mik_csr *A, int *layout)
{
    if (!A || !layout) return MIK_ERR_INVALID;
    *layout = spmv_kernel_choice(A);
    return MIK_OK;
}

extern "C" int mik_csr_set_layout(mik_csr *A, int layout)
{
    if (!A) return MIK_ERR_INVALID;
    if (layout != 0 && layout != -1) return mik_fail(A->ctx, MIK_ERR_NOTIMPL, "mik_csr_set_layout: only 0 (CSR arrays) and -1 (automatic) can be requested");
    if (layout == 0 && !A->col) return mik_fail(A->ctx, MIK_ERR_NOTIMPL, "mik_csr_set_layout: the CSR arrays were released (mik_csr_compact)");
    A->force_layout = layout;
    return MIK_OK;
}

extern "C" int mik_csr_stored_bytes(const mik_csr *A, int64_t *bytes)
{
    if (!A || !bytes) return MIK_ERR_INVALID;
    int layout = 0;
    (void)mik_csr_layout(A, &layout);
    const int64_t es = (int64_t)mik_dtype_size(A->dtype);
    const int64_t nb = (A->n_rows + MIK_BLOCK - 1) / MIK_BLOCK;
    switch (layout) {
    case 5: *bytes = A->n_rows + nb * ((A->sdia_recs && A->ctx->tuning[MIK_KNOB_SDIA_KERNEL] != 1) ? 64 : 4) + (int64_t)A->sdia_npat * (80 + 8 * es); break;
    case 6: *bytes = A->n_rows * 4 + nb * 4 + (nb + 1) / 2 * 64 + (int64_t)A->sdiaw_npat * A->sdiaw_pat_bytes; break;
    case 4: *bytes = A->sdia_entries * es + A->n_rows + nb * 36; break;
    case 1: *bytes = A->jds_groups * (16 / es) * (es + 4) + A->n_rows * 2 + ((A->n_rows + 63) / 64 + 1) * 4 +
                     (A->n_long ? (A->nnz - A->jds_short_nnz) * (es + 4) + 12LL * A->n_long + 4LL * A->n_seg + 16LL * A->n_cut : 0); break;
    default: *bytes = A->nnz * (es + 4) + (A->n_rows + 1) * 4 + (A->n_long ? A->n_rows + 12LL * A->n_long + 4LL * A->n_seg + 16LL * A->n_cut : 0) +
                      ((A->xwin_lo && !spmv_csr_rowgather(A)) ? nb * 4 : 0) + ((A->rperm && !spmv_csr_rowgather(A)) ? A->n_rows : 0); break;
    }
    return MIK_OK;
}

extern "C" int mik_csr_info(const mik_csr *A, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int *dtype)
{
    if (!A) return MIK_ERR_INVALID;
    if (n_rows) *n_rows = A->n_rows;
    if (n_cols) *n_cols = A->n_cols;
    if (nnz) *nnz = A->nnz;
    if (dtype) *dtype = A->dtype;
    return MIK_OK;
}

// ---------------------------------------------------------------------------------------------
// SpMV
// ---------------------------------------------------------------------------------------------
template <typename T>
int mik_spmv_launch_range(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done, int rb_begin,
                          int rb_count);

// Which kernel mik_spmv_launch_range picks for this operator under the current development knobs -- ONE function, used
// by the launcher, by mik_csr_layout and by mik_spmv_can_split, so the three can never disagree.
//   5 per-slice offsets + slice-constant values, 4 sliced-ELL + per-slice offsets, 2 sliced-ELL + 8-bit codes, 1 sliced-ELL, 0 CSR
static int spmv_kernel_choice(const mik_csr *A)
{
    const bool csr = A->col != nullptr;                     // false after mik_csr_compact: the development knobs cannot fall back to CSR
    if (A->force_layout == 0 && csr) return 0;              // mik_csr_set_layout
    if ((A->ctx->tuning[MIK_KNOB_LAYOUTS] & 1) == 0 || !csr) {
        if (A->sdia_pats && ((A->ctx->tuning[MIK_KNOB_LAYOUTS] & 4) == 0 || !csr)) return 5;
        if (A->sdia_val && ((A->ctx->tuning[MIK_KNOB_LAYOUTS] & 4) == 0 || !csr)) return 4;
        if (A->sdiaw_pats && ((A->ctx->tuning[MIK_KNOB_LAYOUTS] & 4) == 0 || !csr)) return 6;
        if (A->jds_val && (A->ctx->tuning[MIK_KNOB_LAYOUTS] & 8) == 0) return 1;
    }
    return 0;
}

// The CSR arrays (12 B per entry at fp64) are what every matrix can run on, and what the development knobs fall back to;
// an operator whose active layout is one of the sliced forms never reads them.  mik_csr_compact releases them.
extern "C" int mik_csr_compact(mik_csr *A)
{
    if (!A) return MIK_ERR_INVALID;
    if (!A->col) return MIK_OK;
    if (!(A->sdia_pats || A->sdia_val || A->sdiaw_pats || A->jds_val) || A->n_long)
        return mik_fail(A->ctx, MIK_ERR_NOTIMPL, "mik_csr_compact: this operator runs on its CSR arrays");
    if (A->ctx) { (void)hipSetDevice(A->ctx->device); (void)hipStreamSynchronize(A->ctx->stream); }
    (void)hipFree(A->rowptr); (void)hipFree(A->col); (void)hipFree(A->val);
    A->rowptr = nullptr; A->col = nullptr; A->val = nullptr;
    return MIK_OK;
}

static inline bool spmv_csr_rowgather(const mik_csr *A)
{
    // operators with x windows (irregular rows inside a band, csr_build_xwin) run on the product tile, which gathers from them
    return A->ctx->tuning[MIK_KNOB_CSR_KERNEL] == 2 || (A->ctx->tuning[MIK_KNOB_CSR_KERNEL] == 0 && A->n_long == 0 && !A->xwin_lo && !A->rperm);
}

// k_spmv_sdiab2 (two rows per lane): the operator's class has the lane-neighbour shape, n is even, and no development
// knob asks for another form (16: slices per workgroup; 18: no compiled-in class; 19: 1 = one row per lane)
static bool sdiab2_applies(const mik_csr *A)
{
    return A->sdia_buf_ok && A->sdia_cls >= 1 && (A->n_rows & 1) == 0 && A->ctx->tuning[MIK_KNOB_SDIA_KERNEL] == 0;
}

// k_spmv_sdiab2, the two CSR kernels and the jagged slices (whole launches with the fused dot, no split-off long rows) take the epilogue
// y = A x + c w, dot(x, y) -- the Lanczos step of MINRES -- and dot(z, y) in place of dot(x, y) -- sigma and rho of BiCGStab(l)
bool mik_spmv_has_epilogue(const mik_csr *A)
{
    if (!A || A->ctx->tuning[MIK_KNOB_SOLVER_FORM] == 2) return false;                    // MIK_KNOB_SOLVER_FORM = 2: never
    const int kc = spmv_kernel_choice(A);
    if (kc == 5) return A->sdia_buf_ok && A->ctx->tuning[MIK_KNOB_SDIA_KERNEL] != 1 && sdiab2_applies(A);
    return (kc == 0 || kc == 1) && A->n_long == 0;                       // the CSR kernels and the jagged slices: one row per thread, no split-off long rows
}

// the operator's SpMV moves little more than x and y (the slice-constant layout): the CG step then picks other cache hints
bool mik_spmv_is_light(const mik_csr *A) { return A && (spmv_kernel_choice(A) == 5 || spmv_kernel_choice(A) == 6); }

extern "C" int mik_spmv_kernel(const mik_csr *A, char *name, int len)
{
    if (!A || !name || len <= 0) return MIK_ERR_INVALID;
    const char *k = "k_spmv_rowblock";
    switch (spmv_kernel_choice(A)) {
    case 5: k = A->sdia_buf_ok && A->ctx->tuning[MIK_KNOB_SDIA_KERNEL] != 1 ? (sdiab2_applies(A) ? "k_spmv_sdiab2" : "k_spmv_sdiab") : "k_spmv_sdiac"; break;
    case 6: k = ((A->n_rows & 1) == 0 && A->ctx->tuning[MIK_KNOB_SDIA_KERNEL] != 3) ? "k_spmv_sdiaw2" : "k_spmv_sdiaw"; break;
    case 4: k = "k_spmv_sdia"; break;
    case 1: k = "k_spmv_jds"; break;
    default: k = spmv_csr_rowgather(A) ? "k_spmv_rowgather" : ((A->xwin_lo && (A->ctx->tuning[MIK_KNOB_LAYOUTS] & 96) == 0) ? "k_spmv_rowblock+xwin" : "k_spmv_rowblock"); break;
    }
    snprintf(name, (size_t)len, "%s", k);
    return MIK_OK;
}

// Can mik_spmv_launch_range serve a sub-range of row-blocks for this operator?  The sliced-ELL kernels and the default
// CSR kernel take a first row-block; the dictionary-coded kernel, k_spmv_rowblock (MIK_KNOB_CSR_KERNEL = 1) and operators with
// split-off long rows (their wave-per-row launch covers the whole matrix) do not.
bool mik_spmv_can_split(const mik_csr *A)
{
    const int kc = spmv_kernel_choice(A);
    if (kc == 0) return spmv_csr_rowgather(A) && A->n_long == 0;
    if (kc == 1) return A->n_long == 0;
    return true;
}

template <typename T>
int mik_spmv_launch(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done)
{
    return mik_spmv_launch_range<T>(ctx, A, x, y, fuse_dot, seg_out, done, 0, -1);
}

// Row-blocks [rb_begin, rb_begin + rb_count) only (rb_count < 0: all).  Partial ranges need mik_spmv_can_split.
// rows OUTSIDE the row-blocks [skip_begin, skip_end): one launch for the slice-constant layout's buffer kernel, else two ranges
template <typename T>
static int spmv_launch_impl(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done, int rb_begin, int rb_count,
                            int skip_at, int skip_len);

template <typename T>
int mik_spmv_launch_range(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done, int rb_begin,
                          int rb_count)
{
    return spmv_launch_impl<T>(ctx, A, x, y, fuse_dot, seg_out, done, rb_begin, rb_count, 0x7fffffff, 0);
}

template <typename T>
int mik_spmv_launch_outside(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done, int skip_begin, int skip_end)
{
    const int nb_all = (int)mik_spmv_nwg(A->n_rows);
    if (skip_begin < 0 || skip_end < skip_begin || skip_end > nb_all) return mik_fail(ctx, MIK_ERR_INVALID, "SpMV: bad interior range");
    if (spmv_kernel_choice(A) == 5 && A->sdia_buf_ok && A->ctx->tuning[MIK_KNOB_SDIA_KERNEL] != 1 && skip_begin + (nb_all - skip_end) > 0)
        return spmv_launch_impl<T>(ctx, A, x, y, fuse_dot, seg_out, done, 0, skip_begin + (nb_all - skip_end), skip_begin, skip_end - skip_begin);
    MIK_TRY(mik_spmv_launch_range<T>(ctx, A, x, y, fuse_dot, seg_out, done, 0, skip_begin));
    return mik_spmv_launch_range<T>(ctx, A, x, y, fuse_dot, seg_out, done, skip_end, nb_all - skip_end);
}
template int mik_spmv_launch_outside<double>(mik_ctx *, const mik_csr *, const double *, double *, bool, double *, const int *, int, int);
template int mik_spmv_launch_outside<float>(mik_ctx *, const mik_csr *, const float *, float *, bool, float *, const int *, int, int);

template <typename T>
static int spmv_launch_impl(mik_ctx *ctx, const mik_csr *A, const T *x, T *y, bool fuse_dot, T *seg_out, const int *done, int rb_begin, int rb_count,
                            int skip_at, int skip_len)
{
    const int n = (int)A->n_rows;
    if (n == 0) return MIK_OK;
    const int nb_all = (int)mik_spmv_nwg(n);
    const bool whole = rb_count < 0 || (rb_begin == 0 && rb_count == nb_all);
    if (!whole && !mik_spmv_can_split(A)) return mik_fail(ctx, MIK_ERR_NOTIMPL, "SpMV over a row-block range is not available for this operator layout");
    if (!whole && (rb_begin < 0 || rb_begin + rb_count > nb_all)) return mik_fail(ctx, MIK_ERR_INVALID, "SpMV row-block range out of bounds");
    if (!whole && rb_count == 0) return MIK_OK;
    const int rb0 = whole ? 0 : rb_begin;
    // development knobs (mik_set_tuning): [0] 1 = temporal (cached) streams, [1] 1 = narrow loads,
    // [2] block map (0 = the operator's own choice: strips for banded operators, else identity; < 0 identity;
    //     1 contiguous range per XCD; P >= 8 strips of P row-blocks)
    const int nb = whole ? nb_all : rb_count;
    const bool wide = true;
    int map_mode = A->strip;
    if (!whole && map_mode >= 8 && (rb0 % map_mode != 0 || nb % map_mode != 0)) map_mode = 0;   // strips need whole planes
    if (skip_len > 0) map_mode = 0;
    const int choice = spmv_kernel_choice(A);
    if ((ctx->spmv_ep_w || ctx->spmv_ep_c || ctx->spmv_ep_z) && !(whole && fuse_dot && mik_spmv_has_epilogue(A)))
        return mik_fail(ctx, MIK_ERR_NOTIMPL, "SpMV: the y = A x + c w / dot(z, y) epilogue is not available for this operator's kernel");
    if (choice == 5) {
        // slice patterns {offsets, values} + one mask byte per row (mik_sell.h); G slices per workgroup
        const int G = MIK_SDIAC_G;
        const int wgs = ((nb + G - 1) / G + 7) / 8 * 8;                         // a multiple of 8: see k_spmv_sdiac
        if (A->sdia_buf_ok && ctx->tuning[MIK_KNOB_SDIA_KERNEL] != 1) {                          // development knob MIK_KNOB_SDIA_KERNEL = 1: the flat-load kernel
            // strips of a power-of-two number of row-blocks are mapped by shifts; any other map runs as identity here
            int sshift = -1, nfull = 0;
            if (map_mode >= 8) {
                const int S = map_mode >> 3;
                if ((S & (S - 1)) == 0) { sshift = 0; while ((1 << sshift) < S) ++sshift; nfull = nb / map_mode * map_mode; }
            }
#define MIK_SDIAB_GO4(FD, NTV, GG, C)                                                                                                      \
    hipLaunchKernelGGL((k_spmv_sdiab<T, FD, NTV, GG, mik_sdiab_cls_ns(C), mik_sdiab_cls_cq(C)>), dim3(wgs), dim3(MIK_BLOCK), 0, ctx->stream, n, (int)A->n_cols, A->sdia_koff, \
                       rb0, nb, nfull, sshift, skip_at, skip_len, (const SdiaSliceRec *)A->sdia_recs, (const SdiaPattern<T> *)A->sdia_pats, A->sdia_mask, x, y, seg_out, done)
#define MIK_SDIAB_GO3(FD, NTV, GG)                                                                                                          \
    do { if (cls == 1) MIK_SDIAB_GO4(FD, NTV, GG, 1); else if (cls == 2) MIK_SDIAB_GO4(FD, NTV, GG, 2); else if (cls == 3) MIK_SDIAB_GO4(FD, NTV, GG, 3); \
         else MIK_SDIAB_GO4(FD, NTV, GG, 0); } while (0)
#define MIK_SDIAB_GO(FD, NTV)                                                                      \
    do { if (G == 1) MIK_SDIAB_GO4(FD, NTV, 1, 0); else if (G == 4) MIK_SDIAB_GO4(FD, NTV, 4, 0); else MIK_SDIAB_GO3(FD, NTV, 2); } while (0)
            const int cls = ctx->tuning[MIK_KNOB_SDIA_KERNEL] == 2 ? 0 : A->sdia_cls;            // development knob MIK_KNOB_SDIA_KERNEL = 2: slot-by-slot path only
            if (sdiab2_applies(A) && skip_len == 0 && (rb0 & 1) == 0 && ((nb & 1) == 0 || rb0 + nb == nb_all)) {
                // two rows per lane, 16-byte gathers (k_spmv_sdiab2): workgroups over PAIRS of slices; strips halve with them
                const int np = (nb + 1) / 2, pb0 = rb0 / 2, wg2 = (np + 7) / 8 * 8;
                const int ps = sshift >= 1 ? sshift - 1 : -1, pfull = sshift >= 1 ? nfull / 2 : 0;
#define MIK_SDIAB2_GO4(FD, NTV, C)                                                                                                       \
    hipLaunchKernelGGL((k_spmv_sdiab2<T, FD, NTV, mik_sdiab_cls_ns(C), mik_sdiab_cls_cq(C)>), dim3(wg2), dim3(MIK_BLOCK), 0, ctx->stream, n, (int)A->n_cols, A->sdia_koff, \
                       pb0, np, pfull, ps, nb_all, (const SdiaSliceRec *)A->sdia_recs, (const SdiaPattern<T> *)A->sdia_pats, A->sdia_mask, x, y, seg_out, done, \
                       (const T *)ctx->spmv_ep_w, (const T *)ctx->spmv_ep_c, (const T *)ctx->spmv_ep_z)
#define MIK_SDIAB2_GO(FD, NTV) do { if (cls == 1) MIK_SDIAB2_GO4(FD, NTV, 1); else if (cls == 2) MIK_SDIAB2_GO4(FD, NTV, 2); else MIK_SDIAB2_GO4(FD, NTV, 3); } while (0)
                if (fuse_dot) { MIK_SDIAB2_GO(true, true); }
                else          { MIK_SDIAB2_GO(false, true); }
#undef MIK_SDIAB2_GO4
#undef MIK_SDIAB2_GO
                MIK_LAUNCH_CHECK(ctx);
                return MIK_OK;
            }
            if (fuse_dot) { MIK_SDIAB_GO(true, true); }
            else          { MIK_SDIAB_GO(false, true); }
#undef MIK_SDIAB_GO4
#undef MIK_SDIAB_GO3
#undef MIK_SDIAB_GO
            MIK_LAUNCH_CHECK(ctx);
            return MIK_OK;
        }
#define MIK_SDIAC_GO3(FD, NTV, GG)                                                                                             \
    hipLaunchKernelGGL((k_spmv_sdiac<T, FD, NTV, GG>), dim3(wgs), dim3(MIK_BLOCK), 0, ctx->stream, n, (int)A->n_cols, rb0, nb, map_mode, A->sdia_pat_id, \
                       (const SdiaPattern<T> *)A->sdia_pats, A->sdia_mask, x, y, seg_out, done)
#define MIK_SDIAC_GO(FD, NTV)                                                                      \
    do { if (G == 1) MIK_SDIAC_GO3(FD, NTV, 1); else if (G == 4) MIK_SDIAC_GO3(FD, NTV, 4); else MIK_SDIAC_GO3(FD, NTV, 2); } while (0)
        if (fuse_dot) { MIK_SDIAC_GO(true, true); }
        else          { MIK_SDIAC_GO(false, true); }
#undef MIK_SDIAC_GO3
#undef MIK_SDIAC_GO
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    if (choice == 6 && (n & 1) == 0 && (uint64_t)A->n_cols * sizeof(T) < 0x7FFFFFF0ull && ctx->tuning[MIK_KNOB_SDIA_KERNEL] != 3 && (rb0 & 1) == 0 &&
        ((nb & 1) == 0 || rb0 + nb == nb_all)) {
        // ... two rows per lane (k_spmv_sdiaw2): workgroups over PAIRS of slices (MIK_KNOB_SDIA_KERNEL = 3: one row per lane)
        const int np = (nb + 1) / 2, pb0 = rb0 / 2;
        const int pmode = map_mode >= 16 ? (map_mode / 2 + 7) / 8 * 8 : 0;       // strips of slice PAIRS
#define MIK_SDIAW2_GO(FD, NTV)                                                                                                \
    hipLaunchKernelGGL((k_spmv_sdiaw2<T, FD, NTV>), dim3(np), dim3(MIK_BLOCK), 0, ctx->stream, n, (int)A->n_cols, A->sdiaw_koff, pb0, np, pmode, nb_all, \
                       A->sdiaw_pat_id, (const SdiawPattern<T> *)A->sdiaw_pats, A->sdiaw_mask, (const uint4 *)A->sdiaw_uz, x, y, seg_out, done)
        if (fuse_dot) { MIK_SDIAW2_GO(true, true); }
        else          { MIK_SDIAW2_GO(false, true); }
#undef MIK_SDIAW2_GO
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    if (choice == 6) {
        // slice patterns of up to 32 {offset, value} pairs + one mask word per row (mik_sell.h)
#define MIK_SDIAW_GO(FD, NTV)                                                                                                 \
    hipLaunchKernelGGL((k_spmv_sdiaw<T, FD, NTV>), dim3(nb), dim3(MIK_BLOCK), 0, ctx->stream, n, A->sdiaw_koff, rb0, nb, map_mode, A->sdiaw_pat_id, \
                       (const SdiawPattern<T> *)A->sdiaw_pats, A->sdiaw_mask, x, y, seg_out, done)
        if (fuse_dot) { MIK_SDIAW_GO(true, true); }
        else          { MIK_SDIAW_GO(false, true); }
#undef MIK_SDIAW_GO
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    if (choice == 4) {
        // sliced-ELL values + per-slice offsets + row masks (mik_sell.h)
#define MIK_SDIA_GO(FD, NTV)                                                                                                  \
    hipLaunchKernelGGL((k_spmv_sdia<T, FD, NTV>), dim3(nb), dim3(MIK_BLOCK), 0, ctx->stream, n, (int)A->n_cols, rb0, nb, map_mode, A->sdia_ptr, \
                       A->sdia_off, A->sdia_tri, A->sdia_mask, (const T *)A->sdia_val, x, y, seg_out, done)
        if (fuse_dot) { MIK_SDIA_GO(true, true); }
        else          { MIK_SDIA_GO(false, true); }
#undef MIK_SDIA_GO
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    const int nlong = A->n_long;
    LongTab lt{};
    if (nlong) {
        int *tb = A->long_rows;
        lt.rows = tb; lt.starts = tb + nlong; lt.lens = tb + 2 * nlong;
        lt.seg_row = tb + 3 * nlong;
        lt.cut_row = lt.seg_row + A->n_seg; lt.cut_first = lt.cut_row + A->n_cut; lt.cut_nseg = lt.cut_first + A->n_cut;
        lt.tickets = (unsigned *)(tb + 3 * nlong + A->n_seg + 3 * A->n_cut);
        lt.seg_sum = A->seg_sum; lt.nlong = nlong;
    }
    const int nlb = (nlong + 3) / 4;                    // one wave per virtual row (a whole row or a segment of a cut row)
    const int *lwin = (A->long_win && (ctx->tuning[MIK_KNOB_LAYOUTS] & 96) == 0 && mik_aligned16(x)) ? A->long_win : nullptr;   // the long-row workgroups' windows of x in LDS
    if (choice == 1) {
        // jagged slices (mik_jds.h): one row per lane, 16-byte operator streams; the workgroups of split-off long rows lead the same
        // launch.  dot(x, y) is formed inside unless long rows exist (their sums arrive from other workgroups): then by k_rowdot.
        using IV = typename WideVec<T>::idx;
        using VV = typename WideVec<T>::val;
        const bool inside = fuse_dot && nlong == 0;
#define MIK_JDS_GO(FD, NTV, MG)                                                                                                        \
    hipLaunchKernelGGL((k_spmv_jds<T, FD, NTV, MG>), dim3(nb + (MG ? nlb : 0)), dim3(MIK_BLOCK), 0, ctx->stream, n, rb0, A->jds_ptr, A->jds_len, \
                       (const IV *)A->jds_col, (const VV *)A->jds_val, x, y, seg_out, done, nlb, lt, A->col, (const T *)A->val, (const T *)ctx->spmv_ep_w, (const T *)ctx->spmv_ep_c, (const T *)ctx->spmv_ep_z)
#define MIK_JDS_GO2(NTV)                                                                      \
    do {                                                                                      \
        if (inside) MIK_JDS_GO(true, NTV, false);                                             \
        else if (nlong) MIK_JDS_GO(false, NTV, true);                                         \
        else MIK_JDS_GO(false, NTV, false);                                                   \
    } while (0)
        MIK_JDS_GO2(true);
#undef MIK_JDS_GO2
#undef MIK_JDS_GO
        MIK_LAUNCH_CHECK(ctx);
        if (fuse_dot && !inside) {
            hipLaunchKernelGGL((k_rowdot<T>), dim3(nb_all), dim3(MIK_BLOCK), 0, ctx->stream, n, x, (const T *)y, seg_out, done);
            MIK_LAUNCH_CHECK(ctx);
        }
        return MIK_OK;
    }
    // CSR kernels.  k_spmv_rowgather (row-block tile filled by LDS-DMA, per-row gather) unless the operator has split-off
    // long rows: then k_spmv_rowblock, whose launch carries the long-row workgroups along (one launch instead of two:
    // 180 vs 196 us on the random configs[4] stand-in, 107 vs 121 us on the banded one).  MIK_KNOB_CSR_KERNEL: 0 = that rule,
    // 1 = always k_spmv_rowblock, 2 = always k_spmv_rowgather.  Same results bit for bit (tests/test_gpu_layouts.py).
    if (spmv_csr_rowgather(A)) {
        if (nlong) {   // long rows first (whole launches only, see mik_spmv_can_split); the row kernel then picks y[r] up
            hipLaunchKernelGGL((k_spmv_longrows<T>), dim3(nlb), dim3(MIK_BLOCK), lwin ? sizeof(T) * (size_t)A->long_lw : 0, ctx->stream, lt, A->col, (const T *)A->val, x, y, done,
                               lwin, A->long_lw);
            MIK_LAUNCH_CHECK(ctx);
        }
#define MIK_RG_GO(FD, NTV)                                                                                                      \
    hipLaunchKernelGGL((k_spmv_rowgather<T, FD, NTV>), dim3(nb), dim3(MIK_BLOCK), 0, ctx->stream, n, rb0, nb, map_mode, A->rowptr, \
                       A->col, (const T *)A->val, x, y, seg_out, done, A->is_long, (const T *)ctx->spmv_ep_w, (const T *)ctx->spmv_ep_c, (const T *)ctx->spmv_ep_z)
        if (fuse_dot) { MIK_RG_GO(true, true); }
        else          { MIK_RG_GO(false, true); }
#undef MIK_RG_GO
        MIK_LAUNCH_CHECK(ctx);
        return MIK_OK;
    }
    const bool merge = nlong > 0 && !fuse_dot;           // one launch: long-row workgroups first, then row-blocks
    // The operator streams of the product tile are read with the DEFAULT cache policy when x is served from LDS windows: non-temporal
    // streams, right for the stencil operators, cost this kernel 10 % there (banded configs[4] stand-in: 72.8 us streamed, 65.7 us
    // cached -- profiles/r04_c5_*).  (A/B in round 5: cached 186.1 us, streamed 191.0 us on `random` -- profiles/r05_c5_random_stream_policy_ab.txt.)
    // (Only where x comes from LDS windows: without them -- the `random` stand-in -- cached streams gain 3 % back to back and lose 6 % inside
    // gmres!, where they push the Krylov basis out of the caches: 260 -> 275 us per inner iteration.)
    const bool nt_rb = !A->xwin_lo;
    if (nlong && !merge) {
        // fused dot: long rows first in their own launch, the row-block kernel then picks y[r] up
        hipLaunchKernelGGL((k_spmv_longrows<T>), dim3(nlb), dim3(MIK_BLOCK), lwin ? sizeof(T) * (size_t)A->long_lw : 0, ctx->stream, lt, A->col, (const T *)A->val, x, y, done,
                           lwin, A->long_lw);
        MIK_LAUNCH_CHECK(ctx);
    }
    const dim3 grid(nb + (merge ? nlb : 0)), block(MIK_BLOCK);
    // x served from an LDS window per row-block (csr_build_xwin; MIK_KNOB_LAYOUTS bit 6: off at launch) -- wide loads and an aligned x only
    const bool xwin = A->xwin_lo && wide && (ctx->tuning[MIK_KNOB_LAYOUTS] & 96) == 0 && mik_aligned16(x);
    constexpr int RB_TILE = MIK_SPMV_TILE * (int)(8 / sizeof(T));
    const size_t dyn = sizeof(T) * ((size_t)RB_TILE + 12 + (xwin ? (size_t)A->xwin_span : 0));          // [product tile + 8][4 wave sums][x window]
    const int lw_launch = (int)std::min<size_t>((size_t)A->long_lw, dyn / 1024 * (1024 / sizeof(T)));  // what a long-row workgroup of this launch can hold
    const bool rp = A->rperm && wide && (ctx->tuning[MIK_KNOB_LAYOUTS] & 96) == 0;         // rows of a block over its threads by length (mik_build_rperm_host)
#define MIK_SPMV_GO(FD, NT, WD, MG, XW, RP)                                                                      \
    hipLaunchKernelGGL((k_spmv_rowblock<T, FD, NT, WD, MG, XW, RP>), grid, block, dyn, ctx->stream, n, nb, map_mode, A->rowptr, \
                       A->col, (const T *)A->val, x, y, seg_out, done, A->is_long, nlb, lt, (const int *)A->xwin_lo, A->xwin_span, (const unsigned char *)A->rperm, \
                       MG ? lwin : (const int *)nullptr, (MG && lwin && A->long_spread) ? -lw_launch : lw_launch, (const T *)ctx->spmv_ep_w, (const T *)ctx->spmv_ep_c, (const T *)ctx->spmv_ep_z)
#define MIK_SPMV_GO3(FD, NT, MG)                                                          \
    do {                                                                                  \
        if (xwin) { if (rp) MIK_SPMV_GO(FD, NT, true, MG, true, true); else MIK_SPMV_GO(FD, NT, true, MG, true, false); }   \
        else if (rp) MIK_SPMV_GO(FD, NT, true, MG, false, true);                          \
        else if (wide) MIK_SPMV_GO(FD, NT, true, MG, false, false);                       \
        else MIK_SPMV_GO(FD, NT, false, MG, false, false);                                \
    } while (0)
#define MIK_SPMV_GO2(FD, MG) do { if (nt_rb) MIK_SPMV_GO3(FD, true, MG); else MIK_SPMV_GO3(FD, false, MG); } while (0)
    if (fuse_dot) MIK_SPMV_GO2(true, false);
    else if (merge) MIK_SPMV_GO2(false, true);
    else MIK_SPMV_GO2(false, false);
#undef MIK_SPMV_GO2
#undef MIK_SPMV_GO3
#undef MIK_SPMV_GO
    MIK_LAUNCH_CHECK(ctx);
    return MIK_OK;
}
template int mik_spmv_launch<double>(mik_ctx *, const mik_csr *, const double *, double *, bool, double *, const int *);
template int mik_spmv_launch<float>(mik_ctx *, const mik_csr *, const float *, float *, bool, float *, const int *);
template int mik_spmv_launch_range<double>(mik_ctx *, const mik_csr *, const double *, double *, bool, double *, const int *, int, int);
template int mik_spmv_launch_range<float>(mik_ctx *, const mik_csr *, const float *, float *, bool, float *, const int *, int, int);

extern "C" int mik_spmv(mik_ctx *ctx, const mik_csr *A, const void *x, void *y)
{
    if (!ctx || !A || !x || !y) return MIK_ERR_INVALID;
    if (x == y) return mik_fail(ctx, MIK_ERR_INVALID, "mik_spmv: x and y must not alias");
    return A->dtype == MIK_F64 ? mik_spmv_launch<double>(ctx, A, (const double *)x, (double *)y, false, nullptr, nullptr)
                               : mik_spmv_launch<float>(ctx, A, (const float *)x, (float *)y, false, nullptr, nullptr);
}

template <typename T>
static int time_spmv_impl(mik_ctx *ctx, const mik_csr *A, const void *x, void *y, int fused, int reps, double *avg_ms)
{
    int rc = mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>((A->n_rows + MIK_BLOCK - 1) / MIK_BLOCK, 1));
    if (rc) return rc;
    hipEvent_t e0, e1;
    MIK_HIP(ctx, hipEventCreate(&e0));
    MIK_HIP(ctx, hipEventCreate(&e1));
    MIK_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < reps; ++i) {
        rc = mik_spmv_launch<T>(ctx, A, (const T *)x, (T *)y, fused != 0, (T *)ctx->partials, nullptr);
        if (rc) return rc;
    }
    MIK_HIP(ctx, hipEventRecord(e1, ctx->stream));
    MIK_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    MIK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / reps;
    return MIK_OK;
}

extern "C" int mik_time_spmv(mik_ctx *ctx, const mik_csr *A, const void *x, void *y, int fused_dot, int reps, double *avg_ms)
{
    if (!ctx || !A || !x || !y || reps <= 0 || !avg_ms) return MIK_ERR_INVALID;
    return A->dtype == MIK_F64 ? time_spmv_impl<double>(ctx, A, x, y, fused_dot, reps, avg_ms)
                               : time_spmv_impl<float>(ctx, A, x, y, fused_dot, reps, avg_ms);
}

// ---------------------------------------------------------------------------------------------
// BLAS-1 forms
// ---------------------------------------------------------------------------------------------
template <typename T> static int reduce_to_host(mik_ctx *ctx, int64_t n, T *out)
{
    // level 2 of the segment sums sitting in ctx->partials -> ctx->coef[0] -> host
    const int64_t nseg = mik_nseg<T>(n);
    hipLaunchKernelGGL((k_finalize_store<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, nseg,
                       (int64_t)0, (T *)ctx->coef, (const int *)nullptr);
    MIK_LAUNCH_CHECK(ctx);
    return mik_read_scalars<T>(ctx, (const T *)ctx->coef, 1, out);
}

// norm(x) from the sum of squares t the caller has just reduced: sqrt(t) (host sqrt: IEEE correctly rounded) when t
// is inside the safe range, the scaled recomputation otherwise (include/mik.h "Norms").
template <typename T> static int norm_from_sumsq(mik_ctx *ctx, int64_t n, const T *x, T t, T *out)
{
    if (mik_nrm_in_range(t)) { *out = (T)std::sqrt(t); return MIK_OK; }
    return mik_safe_norm_slow<T>(ctx, n, x, out);
}

template <typename T> int mik_safe_norm_slow(mik_ctx *ctx, int64_t n, const T *x, T *out)
{
    if (n <= 0) { *out = T(0); return MIK_OK; }
    T *scr = (T *)((unsigned char *)ctx->coef + mik_ctx::COEF_SAFE_SLOT);
    const int grid = (int)std::min<int64_t>((n + MIK_BLOCK - 1) / MIK_BLOCK, 1024);
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(std::max<int64_t>(mik_nseg<T>(n), grid), 1)));
    hipLaunchKernelGGL((k_amax<T>), dim3(grid), dim3(MIK_BLOCK), 0, ctx->stream, n, x, (T *)ctx->partials);
    MIK_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL((k_amax<T>), dim3(1), dim3(MIK_BLOCK), 0, ctx->stream, (int64_t)grid, (const T *)ctx->partials, scr);
    MIK_LAUNCH_CHECK(ctx);
    T amax;
    MIK_TRY(mik_read_scalars<T>(ctx, scr, 1, &amax));
    if (amax == T(0) || amax != amax || amax > std::numeric_limits<T>::max()) { *out = amax; return MIK_OK; }   // 0, NaN, Inf as they are
    int e;
    (void)std::frexp((double)amax, &e);                 // amax = f * 2^e, f in [0.5, 1)
    e = std::max(-NrmRange<T>::EC, std::min(NrmRange<T>::EC, e));   // keep s and 1 / s normal numbers of T
    const T sc = (T)std::ldexp(1.0, -e), sinv = (T)std::ldexp(1.0, e);
    OpScaledSq<T> op{x, sc};
    MIK_TRY((launch_map<T>(ctx, n, op, mik_aligned16(x), (T *)ctx->partials, nullptr)));
    hipLaunchKernelGGL((k_finalize_store<T>), dim3(1), dim3(MIK_FIN_THREADS), 0, ctx->stream, (const T *)ctx->partials, mik_nseg<T>(n),
                       (int64_t)0, scr, (const int *)nullptr);
    MIK_LAUNCH_CHECK(ctx);
    T t2;
    MIK_TRY(mik_read_scalars<T>(ctx, scr, 1, &t2));
    *out = (T)std::sqrt(t2) * sinv;
    return MIK_OK;
}
template int mik_safe_norm_slow<double>(mik_ctx *, int64_t, const double *, double *);
template int mik_safe_norm_slow<float>(mik_ctx *, int64_t, const float *, float *);

template <typename T> static int dot_impl(mik_ctx *ctx, int64_t n, const void *x, const void *y, void *out, bool nrm)
{
    int rc = mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(mik_nseg<T>(n), 1));
    if (rc) return rc;
    OpDot<T> op{(const T *)x, (const T *)y};
    rc = launch_map<T>(ctx, n, op, mik_aligned16(x) && mik_aligned16(y), (T *)ctx->partials, nullptr);
    if (rc) return rc;
    T v;
    MIK_TRY(reduce_to_host<T>(ctx, n, &v));
    if (!nrm) { *(T *)out = v; return MIK_OK; }
    return norm_from_sumsq<T>(ctx, n, (const T *)x, v, (T *)out);
}

extern "C" int mik_dot(mik_ctx *ctx, int dtype, int64_t n, const void *x, const void *y, void *out)
{
    if (!ctx || n < 0 || !out || (n && (!x || !y))) return MIK_ERR_INVALID;
    return dtype == MIK_F64 ? dot_impl<double>(ctx, n, x, y, out, false) : dot_impl<float>(ctx, n, x, y, out, false);
}

extern "C" int mik_nrm2(mik_ctx *ctx, int dtype, int64_t n, const void *x, void *out)
{
    if (!ctx || n < 0 || !out || (n && !x)) return MIK_ERR_INVALID;
    return dtype == MIK_F64 ? dot_impl<double>(ctx, n, x, x, out, true) : dot_impl<float>(ctx, n, x, x, out, true);
}

extern "C" int mik_axpy(mik_ctx *ctx, int dtype, int64_t n, const void *alpha, const void *x, void *y)
{
    if (!ctx || n < 0 || !alpha || (n && (!x || !y))) return MIK_ERR_INVALID;
    const bool vec = mik_aligned16(x) && mik_aligned16(y);
    if (dtype == MIK_F64) { OpAxpy<double> op{(const double *)x, (double *)y, coef_val(*(const double *)alpha)}; return launch_map<double>(ctx, n, op, vec, (double *)nullptr, nullptr); }
    if (dtype == MIK_F32) { OpAxpy<float> op{(const float *)x, (float *)y, coef_val(*(const float *)alpha)}; return launch_map<float>(ctx, n, op, vec, (float *)nullptr, nullptr); }
    return MIK_ERR_INVALID;
}

extern "C" int mik_xpby(mik_ctx *ctx, int dtype, int64_t n, const void *x, const void *beta, void *y)
{
    if (!ctx || n < 0 || !beta || (n && (!x || !y))) return MIK_ERR_INVALID;
    const bool vec = mik_aligned16(x) && mik_aligned16(y);
    if (dtype == MIK_F64) { OpXpby<double> op{(const double *)x, (double *)y, coef_val(*(const double *)beta)}; return launch_map<double>(ctx, n, op, vec, (double *)nullptr, nullptr); }
    if (dtype == MIK_F32) { OpXpby<float> op{(const float *)x, (float *)y, coef_val(*(const float *)beta)}; return launch_map<float>(ctx, n, op, vec, (float *)nullptr, nullptr); }
    return MIK_ERR_INVALID;
}

extern "C" int mik_sub(mik_ctx *ctx, int dtype, int64_t n, const void *x, void *y)
{
    if (!ctx || n < 0 || (n && (!x || !y))) return MIK_ERR_INVALID;
    const bool vec = mik_aligned16(x) && mik_aligned16(y);
    if (dtype == MIK_F64) { OpSub<double> op{(const double *)x, (double *)y}; return launch_map<double>(ctx, n, op, vec, (double *)nullptr, nullptr); }
    if (dtype == MIK_F32) { OpSub<float> op{(const float *)x, (float *)y}; return launch_map<float>(ctx, n, op, vec, (float *)nullptr, nullptr); }
    return MIK_ERR_INVALID;
}

extern "C" int mik_scal(mik_ctx *ctx, int dtype, int64_t n, const void *alpha, void *x)
{
    if (!ctx || n < 0 || !alpha || (n && !x)) return MIK_ERR_INVALID;
    const bool vec = mik_aligned16(x);
    if (dtype == MIK_F64) { OpScal<double> op{(double *)x, coef_val(*(const double *)alpha)}; return launch_map<double>(ctx, n, op, vec, (double *)nullptr, nullptr); }
    if (dtype == MIK_F32) { OpScal<float> op{(float *)x, coef_val(*(const float *)alpha)}; return launch_map<float>(ctx, n, op, vec, (float *)nullptr, nullptr); }
    return MIK_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// fused sweeps of the widened solvers (one pass over HBM instead of the reference's 2-6)
// ---------------------------------------------------------------------------------------------
template <typename T>
static int axpy_dot_impl(mik_ctx *ctx, int64_t n, const void *alpha, const void *x, void *y, const void *z, void *out, int hints)
{
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(mik_nseg<T>(n), 1)));
    OpAxpyDot<T> op{(const T *)x, (T *)y, (const T *)z, coef_val(x ? *(const T *)alpha : T(0)), hints};
    const bool vec = mik_aligned16(y) && (!x || mik_aligned16(x)) && (!z || mik_aligned16(z));
    MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)ctx->partials, nullptr)));
    T v;
    MIK_TRY(reduce_to_host<T>(ctx, n, &v));
    if (z) { *(T *)out = v; return MIK_OK; }
    return norm_from_sumsq<T>(ctx, n, (const T *)y, v, (T *)out);           // norm(y)
}

extern "C" int mik_axpy_dot(mik_ctx *ctx, int dtype, int64_t n, const void *alpha, const void *x, void *y, const void *z, void *out,
                            int hints)
{
    if (!ctx || n < 0 || !out || (x && !alpha) || (n && !y)) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return axpy_dot_impl<double>(ctx, n, alpha, x, y, z, out, hints);
    if (dtype == MIK_F32) return axpy_dot_impl<float>(ctx, n, alpha, x, y, z, out, hints);
    return MIK_ERR_INVALID;
}

template <typename T>
static int axpy2_nrm2_impl(mik_ctx *ctx, int64_t n, const void *alpha, const void *u, void *x, const void *c, void *r, void *out, int hints)
{
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(mik_nseg<T>(n), 1)));
    OpCgUpdate<T> op{(T *)x, (T *)r, (const T *)u, (const T *)c, coef_val(*(const T *)alpha), hints & 7};
    const bool vec = mik_aligned16(u) && mik_aligned16(x) && mik_aligned16(c) && mik_aligned16(r);
    MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)ctx->partials, nullptr)));
    T v;
    MIK_TRY(reduce_to_host<T>(ctx, n, &v));
    return norm_from_sumsq<T>(ctx, n, (const T *)r, v, (T *)out);           // norm(r)
}

extern "C" int mik_axpy2_nrm2(mik_ctx *ctx, int dtype, int64_t n, const void *alpha, const void *u, void *x, const void *c, void *r,
                              void *out, int hints)
{
    if (!ctx || n < 0 || !out || !alpha || (n && (!u || !x || !c || !r))) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return axpy2_nrm2_impl<double>(ctx, n, alpha, u, x, c, r, out, hints);
    if (dtype == MIK_F32) return axpy2_nrm2_impl<float>(ctx, n, alpha, u, x, c, r, out, hints);
    return MIK_ERR_INVALID;
}

template <typename T> static int xpby_nrm2_impl(mik_ctx *ctx, int64_t n, const void *x, const void *beta, void *y, void *out)
{
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(mik_nseg<T>(n), 1)));
    OpXpbyNrm<T> op{(const T *)x, (T *)y, *(const T *)beta};
    MIK_TRY((launch_map<T>(ctx, n, op, mik_aligned16(x) && mik_aligned16(y), (T *)ctx->partials, nullptr)));
    T v;
    MIK_TRY(reduce_to_host<T>(ctx, n, &v));
    return norm_from_sumsq<T>(ctx, n, (const T *)y, v, (T *)out);           // norm(y)
}

extern "C" int mik_xpby_nrm2(mik_ctx *ctx, int dtype, int64_t n, const void *x, const void *beta, void *y, void *out)
{
    if (!ctx || n < 0 || !out || !beta || (n && (!x || !y))) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return xpby_nrm2_impl<double>(ctx, n, x, beta, y, out);
    if (dtype == MIK_F32) return xpby_nrm2_impl<float>(ctx, n, x, beta, y, out);
    return MIK_ERR_INVALID;
}

template <typename T> static int lsqr_update_impl(mik_ctx *ctx, int64_t n, const void *t1, const void *t2, const void *inv_rho, void *x, void *w, const void *v, void *out)
{
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(mik_nseg<T>(n), 1)));
    OpLsqrUpdate<T> op{(T *)x, (T *)w, (const T *)v, *(const T *)t1, *(const T *)t2, *(const T *)inv_rho};
    MIK_TRY((launch_map<T>(ctx, n, op, mik_aligned16(x) && mik_aligned16(w) && mik_aligned16(v), (T *)ctx->partials, nullptr)));
    T s;
    MIK_TRY(reduce_to_host<T>(ctx, n, &s));
    if (mik_nrm_in_range(s) || n == 0) { *(T *)out = n ? (T)std::sqrt(s) : T(0); return MIK_OK; }
    // |wrho|^2 outside the range of a plain sum of squares: wrho = w .* inv(rho) in a scratch vector, then the scaled norm (rare; allocation and all)
    T *tmp = nullptr;
    MIK_HIP(ctx, hipMalloc((void **)&tmp, sizeof(T) * (size_t)n));
    int rc = MIK_OK;
    hipError_t e = hipMemcpyAsync(tmp, w, sizeof(T) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) rc = mik_fail(ctx, MIK_ERR_HIP, "mik_lsqr_update: %s", hipGetErrorString(e));
    if (!rc) { OpScal<T> sc{tmp, coef_val(*(const T *)inv_rho)}; rc = launch_map<T>(ctx, n, sc, mik_aligned16(tmp), (T *)nullptr, nullptr); }
    if (!rc) rc = mik_safe_norm_slow<T>(ctx, n, tmp, (T *)out);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    return rc;
}

extern "C" int mik_lsqr_update(mik_ctx *ctx, int dtype, int64_t n, const void *t1, const void *t2, const void *inv_rho, void *x, void *w, const void *v, void *out)
{
    if (!ctx || n < 0 || !out || !t1 || !t2 || !inv_rho || (n && (!x || !w || !v))) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return lsqr_update_impl<double>(ctx, n, t1, t2, inv_rho, x, w, v, out);
    if (dtype == MIK_F32) return lsqr_update_impl<float>(ctx, n, t1, t2, inv_rho, x, w, v, out);
    return MIK_ERR_INVALID;
}

template <typename T> static int lsmr_update_impl(mik_ctx *ctx, int64_t n, const void *c1, const void *c2, const void *c3, void *hbar, void *h, void *x, const void *v, void *out)
{
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(mik_nseg<T>(n), 1)));
    OpLsmrUpdate<T> op{(T *)hbar, (T *)h, (T *)x, (const T *)v, *(const T *)c1, *(const T *)c2, *(const T *)c3};
    MIK_TRY((launch_map<T>(ctx, n, op, mik_aligned16(hbar) && mik_aligned16(h) && mik_aligned16(x) && mik_aligned16(v), (T *)ctx->partials, nullptr)));
    T s;
    MIK_TRY(reduce_to_host<T>(ctx, n, &s));
    return norm_from_sumsq<T>(ctx, n, (const T *)x, s, (T *)out);           // norm(x)
}

extern "C" int mik_lsmr_update(mik_ctx *ctx, int dtype, int64_t n, const void *c1, const void *c2, const void *c3, void *hbar, void *h, void *x, const void *v, void *out)
{
    if (!ctx || n < 0 || !out || !c1 || !c2 || !c3 || (n && (!hbar || !h || !x || !v))) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return lsmr_update_impl<double>(ctx, n, c1, c2, c3, hbar, h, x, v, out);
    if (dtype == MIK_F32) return lsmr_update_impl<float>(ctx, n, c1, c2, c3, hbar, h, x, v, out);
    return MIK_ERR_INVALID;
}

template <typename T> static int axpy2_dot_impl(mik_ctx *ctx, int64_t n, const void *a, const void *x1, const void *b, const void *x2, void *y, const void *z, void *out)
{
    MIK_TRY(mik_ensure_partials(ctx, sizeof(T) * (size_t)std::max<int64_t>(mik_nseg<T>(n), 1)));
    OpAxpy2Dot<T> op{(T *)y, (const T *)x1, (const T *)x2, (const T *)z, *(const T *)a, x2 ? *(const T *)b : T(0)};
    const bool vec = mik_aligned16(y) && mik_aligned16(x1) && (!x2 || mik_aligned16(x2)) && (!z || mik_aligned16(z));
    MIK_TRY((launch_map<T>(ctx, n, op, vec, (T *)ctx->partials, nullptr)));
    if (!z) return MIK_OK;
    return reduce_to_host<T>(ctx, n, (T *)out);
}

extern "C" int mik_axpy2_dot(mik_ctx *ctx, int dtype, int64_t n, const void *a, const void *x1, const void *b, const void *x2, void *y, const void *z, void *out)
{
    if (!ctx || n < 0 || !a || (x2 && !b) || (z && !out) || (n && (!x1 || !y))) return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return axpy2_dot_impl<double>(ctx, n, a, x1, b, x2, y, z, out);
    if (dtype == MIK_F32) return axpy2_dot_impl<float>(ctx, n, a, x1, b, x2, y, z, out);
    return MIK_ERR_INVALID;
}

extern "C" int mik_scal2(mik_ctx *ctx, int dtype, int64_t n, const void *a, void *x, const void *b, void *y)
{
    if (!ctx || n < 0 || !a || !b || (n && (!x || !y))) return MIK_ERR_INVALID;
    const bool vec = mik_aligned16(x) && mik_aligned16(y);
    if (dtype == MIK_F64) { OpScal2<double> op{(double *)x, (double *)y, *(const double *)a, *(const double *)b}; return launch_map<double>(ctx, n, op, vec, (double *)nullptr, nullptr); }
    if (dtype == MIK_F32) { OpScal2<float> op{(float *)x, (float *)y, *(const float *)a, *(const float *)b}; return launch_map<float>(ctx, n, op, vec, (float *)nullptr, nullptr); }
    return MIK_ERR_INVALID;
}

extern "C" int mik_qmr_update(mik_ctx *ctx, int dtype, int64_t n, const void *v, const void *neg_h1, const void *p_curr, const void *neg_h0, const void *p_prev,
                              const void *inv, const void *g, void *x, void *p_out)
{
    if (!ctx || n < 0 || !inv || !g || (p_curr && !neg_h1) || (p_prev && !neg_h0) || (n && (!v || !x || !p_out))) return MIK_ERR_INVALID;
    if (p_out == v || p_out == p_curr) return MIK_ERR_INVALID;      // p_out may only be p_prev's storage (element i is read before it is written) or a vector of its own
    const bool vec = mik_aligned16(v) && mik_aligned16(x) && mik_aligned16(p_out) && (!p_curr || mik_aligned16(p_curr)) && (!p_prev || mik_aligned16(p_prev));
    if (dtype == MIK_F64) {
        OpQmrUpdate<double> op{(const double *)v, (const double *)p_curr, (const double *)p_prev, (double *)p_out, (double *)x,
                               p_curr ? *(const double *)neg_h1 : 0.0, p_prev ? *(const double *)neg_h0 : 0.0, *(const double *)inv, *(const double *)g};
        return launch_map<double>(ctx, n, op, vec, (double *)nullptr, nullptr);
    }
    if (dtype == MIK_F32) {
        OpQmrUpdate<float> op{(const float *)v, (const float *)p_curr, (const float *)p_prev, (float *)p_out, (float *)x,
                              p_curr ? *(const float *)neg_h1 : 0.0f, p_prev ? *(const float *)neg_h0 : 0.0f, *(const float *)inv, *(const float *)g};
        return launch_map<float>(ctx, n, op, vec, (float *)nullptr, nullptr);
    }
    return MIK_ERR_INVALID;
}

extern "C" int mik_cheb_direction(mik_ctx *ctx, int dtype, int64_t n, const void *r, const void *pl_diag, const void *beta, int first,
                                  void *u)
{
    if (!ctx || n < 0 || (!first && !beta) || (n && (!r || !u))) return MIK_ERR_INVALID;
    const bool vec = mik_aligned16(r) && mik_aligned16(u) && (!pl_diag || mik_aligned16(pl_diag));
    if (dtype == MIK_F64) {
        OpChebDirection<double> op{(const double *)r, (const double *)pl_diag, (double *)u, first ? 0.0 : *(const double *)beta, first};
        return launch_map<double>(ctx, n, op, vec, (double *)nullptr, nullptr);
    }
    if (dtype == MIK_F32) {
        OpChebDirection<float> op{(const float *)r, (const float *)pl_diag, (float *)u, first ? 0.0f : *(const float *)beta, first};
        return launch_map<float>(ctx, n, op, vec, (float *)nullptr, nullptr);
    }
    return MIK_ERR_INVALID;
}

template <typename T>
static int minres_update_impl(mik_ctx *ctx, int64_t n, const void *inv_h3, void *v_next, const void *v_curr, const void *neg_h1,
                              const void *w_curr, const void *neg_h0, const void *w_prev, const void *inv_h2, void *w_next,
                              const void *rhs0, void *x, int hints)
{
    OpMinresUpdate<T> op{(T *)v_next, (const T *)v_curr, (const T *)w_curr, (const T *)w_prev, (T *)w_next, (T *)x,
                         coef_val(*(const T *)inv_h3), coef_val(w_curr ? *(const T *)neg_h1 : T(0)), coef_val(w_prev ? *(const T *)neg_h0 : T(0)),
                         coef_val(*(const T *)inv_h2), coef_val(*(const T *)rhs0), hints};
    const bool vec = mik_aligned16(v_next) && mik_aligned16(v_curr) && mik_aligned16(w_next) && mik_aligned16(x) &&
                     (!w_curr || mik_aligned16(w_curr)) && (!w_prev || mik_aligned16(w_prev));
    return launch_map<T>(ctx, n, op, vec, (T *)nullptr, nullptr);
}

extern "C" int mik_minres_update(mik_ctx *ctx, int dtype, int64_t n, const void *inv_h3, void *v_next, const void *v_curr,
                                 const void *neg_h1, const void *w_curr, const void *neg_h0, const void *w_prev, const void *inv_h2,
                                 void *w_next, const void *rhs0, void *x, int hints)
{
    if (!ctx || n < 0 || !inv_h3 || !inv_h2 || !rhs0 || (w_curr && !neg_h1) || (w_prev && !neg_h0) ||
        (n && (!v_next || !v_curr || !w_next || !x)))
        return MIK_ERR_INVALID;
    if (dtype == MIK_F64) return minres_update_impl<double>(ctx, n, inv_h3, v_next, v_curr, neg_h1, w_curr, neg_h0, w_prev, inv_h2, w_next, rhs0, x, hints);
    if (dtype == MIK_F32) return minres_update_impl<float>(ctx, n, inv_h3, v_next, v_curr, neg_h1, w_curr, neg_h0, w_prev, inv_h2, w_next, rhs0, x, hints);
    return MIK_ERR_INVALID;
}

extern "C" int mik_divide(mik_ctx *ctx, int dtype, int64_t n, const void *x, const void *d, void *y)
{
    if (!ctx || n < 0 || (n && (!x || !d || !y))) return MIK_ERR_INVALID;
    const bool vec = mik_aligned16(x) && mik_aligned16(d) && mik_aligned16(y);
    if (dtype == MIK_F64) { OpDivide<double> op{(const double *)x, (const double *)d, (double *)y}; return launch_map<double>(ctx, n, op, vec, (double *)nullptr, nullptr); }
    if (dtype == MIK_F32) { OpDivide<float> op{(const float *)x, (const float *)d, (float *)y}; return launch_map<float>(ctx, n, op, vec, (float *)nullptr, nullptr); }
    return MIK_ERR_INVALID;
}
