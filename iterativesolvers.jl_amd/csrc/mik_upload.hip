// mik_upload.hip -- device-side operator upload (mik_csr_create's default path).
//
// The host hands over the SparseMatrixCSC fields as they are (1-based Int64 colptr / rowval + nzval, the layout of
// test/laplace_matrix.jl:12; or a CSR triple).  They are copied to the device once, raw, and everything else happens
// there: validation, Int64 -> Int32 / 0-based conversion, the CSC -> CSR transpose (row histogram, exclusive scan, scatter
// by atomic cursor, per-row sort by column -- the order Julia's column scatter reaches a row), the operator statistics
// (longest row, bandwidth for the XCD strip map, densest row-block) and the analysis for the per-slice-offset layouts of
// csrc/mik_sell.h (slot pattern per 256-row slice, row masks, slice-constancy of the slot values, value slots).  Only the
// per-slice descriptions (144 B per 256 rows) travel back for the pattern table (mik_sdiac_finish).  No host pass over
// the nnz entries, no host staging copies.
//
// Falls back to the host path of mik_core.hip (return MIK_ERR_NOTIMPL, nothing allocated in A) for what is rare and
// order-sensitive there: rows longer than the long-row threshold, duplicate (row, column) entries in CSC input, matrices
// that need the other sliced-ELL layouts.
#include "mik_internal.h"
#include "mik_sell.h"
#include "mik_jds.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

namespace {

constexpr int NO_ROW = 0x7f7f7f7f;             // first_row[] after its memset: no row has this slot

struct UploadStats {
    int bad_ptr, bad_idx, dup, max_row, max_rb, sdia_bad, not_constant, pad_;
    long long bw, slots;
};

// ---------------------------------------------------------------------------------------------
// exclusive scan of an int array (in place), total returned in *total (device)
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_PER = 8;                       // elements per thread
constexpr int SCAN_TILE = MIK_BLOCK * SCAN_PER;   // 2048 per workgroup

__global__ __launch_bounds__(MIK_BLOCK) void k_scan_tiles(int *__restrict__ data, long long n, int *__restrict__ sums)
{
    __shared__ int part[MIK_BLOCK];
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)t * SCAN_PER;
    int v[SCAN_PER], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) {
        v[i] = base + i < n ? data[base + i] : 0;
        s += v[i];
    }
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < MIK_BLOCK; d <<= 1) {     // Hillis-Steele over the 256 thread sums
        const int add = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int run = part[t] - s;                        // exclusive prefix of this thread inside the tile
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) {
        if (base + i < n) data[base + i] = run;
        run += v[i];
    }
    if (t == MIK_BLOCK - 1) sums[blockIdx.x] = part[t];
}

__global__ __launch_bounds__(MIK_BLOCK) void k_scan_add(int *__restrict__ data, long long n, const int *__restrict__ offs)
{
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_PER;
    const int o = offs[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i)
        if (base + i < n) data[base + i] += o;
}

// data[0..n) -> exclusive prefix sums; data[n] (must be allocated) = total.  scratch: >= n / 2048 + n / 2048^2 + 8 ints
static hipError_t device_exclusive_scan(hipStream_t st, int *data, long long n, int *scratch)
{
    if (n <= 0) return hipMemsetAsync(data, 0, sizeof(int), st);
    const long long tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    int *sums = scratch;
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(MIK_BLOCK), 0, st, data, n, sums);
    if (tiles > 1) {
        hipError_t e = device_exclusive_scan(st, sums, tiles, scratch + tiles + 1);      // sums[tiles] = grand total
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)tiles), dim3(MIK_BLOCK), 0, st, data, n, sums);
        return hipMemcpyAsync(data + n, sums + tiles, sizeof(int), hipMemcpyDeviceToDevice, st);
    }
    return hipMemcpyAsync(data + n, sums, sizeof(int), hipMemcpyDeviceToDevice, st);
}

// ---------------------------------------------------------------------------------------------
// validation, conversion, transpose
// ---------------------------------------------------------------------------------------------
__global__ void k_up_check_ptr(const long long *__restrict__ ptr, long long n_major, UploadStats *st)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_major && ptr[j + 1] < ptr[j]) st->bad_ptr = 1;
}

// CSC: count entries per row (rowlen[r + 1]), check the row indices
__global__ void k_up_count_rows(const long long *__restrict__ idx, long long nnz, long long n_minor, int base, int *__restrict__ rowlen, UploadStats *st)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const long long i = idx[k] - base;
    if (i < 0 || i >= n_minor) { st->bad_idx = 1; return; }
    atomicAdd(&rowlen[i], 1);
}

// CSC: column j hands its entries to their rows (position by atomic cursor; the rows are sorted afterwards)
template <typename T>
__global__ void k_up_scatter(const long long *__restrict__ ptr, const long long *__restrict__ idx, const T *__restrict__ val, long long n_cols, int base,
                             const int *__restrict__ rowptr, int *__restrict__ cursor, int *__restrict__ col, T *__restrict__ out)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cols) return;
    for (long long k = ptr[j] - base; k < ptr[j + 1] - base; ++k) {
        const long long r = idx[k] - base;
        const int dst = rowptr[r] + atomicAdd(&cursor[r], 1);
        col[dst] = (int)j;
        out[dst] = val[k];
    }
}

// every row: insertion sort by column (rows are short here; longer ones leave for the host path); duplicates are reported
template <typename T>
__global__ void k_up_sort_rows(const int *__restrict__ rowptr, long long n_rows, int *__restrict__ col, T *__restrict__ val, UploadStats *st)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int a = rowptr[r], b = rowptr[r + 1];
    if (b - a > 256) return;                                     // long row: the host path takes this matrix
    for (int i = a + 1; i < b; ++i) {
        const int c = col[i];
        const T v = val[i];
        int j = i - 1;
        while (j >= a && col[j] > c) { col[j + 1] = col[j]; val[j + 1] = val[j]; --j; }
        col[j + 1] = c;
        val[j + 1] = v;
    }
    for (int i = a + 1; i < b; ++i)
        if (col[i] == col[i - 1]) st->dup = 1;
}

// CSR input: convert in place of a transpose
template <typename T>
__global__ void k_up_convert(const long long *__restrict__ idx, const T *__restrict__ val, long long nnz, long long n_minor, int base,
                             int *__restrict__ col, T *__restrict__ out, UploadStats *st)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const long long i = idx[k] - base;
    if (i < 0 || i >= n_minor) { st->bad_idx = 1; return; }
    col[k] = (int)i;
    out[k] = val[k];
}
__global__ void k_up_rowptr(const long long *__restrict__ ptr, long long n_rows, int base, int *__restrict__ rowptr)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_rows) rowptr[r] = (int)(ptr[r] - base);
}

// longest row, |column - row| of in-block columns (strip map), densest 256-row block
__global__ __launch_bounds__(MIK_BLOCK) void k_up_stats(const int *__restrict__ rowptr, const int *__restrict__ col, long long n_rows, UploadStats *st)
{
    const long long r = (long long)blockIdx.x * MIK_BLOCK + threadIdx.x;
    int len = 0;
    long long bw = 0;
    if (r < n_rows) {
        const int a = rowptr[r], b = rowptr[r + 1];
        len = b - a;
        for (int k = a; k < b; ++k)
            if (col[k] < n_rows) bw = max(bw, (long long)llabs((long long)col[k] - r));
    }
    __shared__ int slen[MIK_BLOCK];
    __shared__ long long sbw[MIK_BLOCK];
    slen[threadIdx.x] = len;
    sbw[threadIdx.x] = bw;
    __syncthreads();
    for (int d = MIK_BLOCK / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            slen[threadIdx.x] = max(slen[threadIdx.x], slen[threadIdx.x + d]);
            sbw[threadIdx.x] = max(sbw[threadIdx.x], sbw[threadIdx.x + d]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(&st->max_row, slen[0]);
        atomicMax((unsigned long long *)&st->bw, (unsigned long long)sbw[0]);
        const long long r0 = (long long)blockIdx.x * MIK_BLOCK;
        atomicMax(&st->max_rb, rowptr[min(r0 + MIK_BLOCK, n_rows)] - rowptr[r0]);
    }
}

// ---------------------------------------------------------------------------------------------
// per-slice-offset layouts (the host builder csr_build_sdia of mik_core.hip, statement for statement)
// ---------------------------------------------------------------------------------------------
// one thread per 256-row slice: the slice's slot pattern = a common super-sequence of its rows' offset sequences, built by
// merging row after row; nslot[b] = slots * 256 (scanned into the value-slot pointer afterwards)
__global__ void k_up_slice_pattern(const int *__restrict__ rowptr, const int *__restrict__ col, long long n_rows, long long nb, int *__restrict__ doff,
                                   int *__restrict__ dtri, int *__restrict__ nslot, UploadStats *st)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    int offs8[8];
    int ns = 0;
    bool ok = true;
    const long long rend = min((b + 1) * MIK_BLOCK, n_rows);
    for (long long r = b * MIK_BLOCK; r < rend && ok; ++r) {
        int p = 0;                                               // next admissible pattern position for this row
        for (int k2 = rowptr[r]; k2 < rowptr[r + 1]; ++k2) {
            const int d = col[k2] - (int)r;
            int q = 0;
            while (q < ns && offs8[q] != d) ++q;
            if (q < ns) {
                if (q < p) { ok = false; break; }                // two rows order the same offsets differently
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
    if (!ok) { st->sdia_bad = 1; ns = 0; }
    int tri = -1;
    for (int q = 0; q < 8; ++q) doff[b * 8 + q] = q < ns ? offs8[q] : 0;
    for (int q = 0; q + 2 < ns; ++q)
        if (offs8[q + 1] == offs8[q] + 1 && offs8[q + 2] == offs8[q] + 2) { tri = q; break; }
    dtri[b] = tri;
    nslot[b] = ns * MIK_BLOCK;
}

// one thread per row: which slots the row has (mask byte) and, per (slice, slot), the first row that has it
__global__ void k_up_row_masks(const int *__restrict__ rowptr, const int *__restrict__ col, long long n_rows, const int *__restrict__ doff,
                               const int *__restrict__ dptr, unsigned char *__restrict__ mask, int *__restrict__ first_row, UploadStats *st)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const long long b = r / MIK_BLOCK;
    const int ns = (dptr[b + 1] - dptr[b]) / MIK_BLOCK;
    const int *so = doff + b * 8;
    int q = 0, prevq = -1, m = 0;
    for (int k2 = rowptr[r]; k2 < rowptr[r + 1]; ++k2) {
        const int d = col[k2] - (int)r;
        while (q < ns && so[q] != d) ++q;                        // columns ascend within a row, so do the slots
        if (q >= ns || q <= prevq) { st->sdia_bad = 1; break; }  // unsorted or duplicate column: keep the other layouts
        m |= 1 << q;
        atomicMin(&first_row[b * 8 + q], (int)r);
        prevq = q;
    }
    mask[r] = (unsigned char)m;
}

template <typename T> struct BitsOf;
template <> struct BitsOf<double> { using type = unsigned long long; };
template <> struct BitsOf<float> { using type = unsigned; };

// one thread per row: does every slot value equal (bit for bit) the value the slot's first row carries?
template <typename T>
__global__ void k_up_row_constancy(const int *__restrict__ rowptr, const int *__restrict__ col, const T *__restrict__ val, long long n_rows,
                                   const int *__restrict__ doff, const int *__restrict__ dptr, const int *__restrict__ first_row, UploadStats *st)
{
    using B = typename BitsOf<T>::type;
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const long long b = r / MIK_BLOCK;
    const int ns = (dptr[b + 1] - dptr[b]) / MIK_BLOCK;
    const int *so = doff + b * 8;
    int q = 0;
    for (int k2 = rowptr[r]; k2 < rowptr[r + 1]; ++k2) {
        const int d = col[k2] - (int)r;
        while (q < ns && so[q] != d) ++q;
        if (q >= ns) break;
        const int fr = first_row[b * 8 + q];
        if (fr == NO_ROW || fr == (int)r) continue;
        for (int k3 = rowptr[fr]; k3 < rowptr[fr + 1]; ++k3)
            if (col[k3] - fr == d) {
                if (__builtin_bit_cast(B, val[k3]) != __builtin_bit_cast(B, val[k2])) st->not_constant = 1;
                break;
            }
    }
}

// one thread per slice: the SdiaPattern-shaped description {ns, tri, cq, dfull, off[8], soff[8] = 0, val[8]}
template <typename T>
__global__ void k_up_slice_desc(const int *__restrict__ rowptr, const int *__restrict__ col, const T *__restrict__ val, long long n_rows, long long nb,
                                const int *__restrict__ doff, const int *__restrict__ dtri, const int *__restrict__ dptr,
                                const unsigned char *__restrict__ mask, const int *__restrict__ first_row, SdiaPattern<T> *__restrict__ desc)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    SdiaPattern<T> P;
    memset(&P, 0, sizeof(P));
    P.ns = (dptr[b + 1] - dptr[b]) / MIK_BLOCK;
    P.tri = dtri[b];
    P.cq = -1;
    for (int q = 0; q < 8; ++q) {
        P.off[q] = doff[b * 8 + q];
        if (q < P.ns && P.off[q] == 0) P.cq = q;
        const int fr = q < P.ns ? first_row[b * 8 + q] : NO_ROW;
        if (fr != NO_ROW)
            for (int k3 = rowptr[fr]; k3 < rowptr[fr + 1]; ++k3)
                if (col[k3] - fr == P.off[q]) P.val[q] = val[k3];
    }
    int dfull = P.cq >= 0;
    const long long rend = min((b + 1) * MIK_BLOCK, n_rows);
    for (long long r = b * MIK_BLOCK; dfull && r < rend; ++r) dfull = (mask[r] >> P.cq) & 1;
    P.dfull = dfull;
    desc[b] = P;
}

// one thread per row: the per-row value slots of k_spmv_sdia (operators whose coefficients vary inside a slice)
template <typename T>
__global__ void k_up_row_values(const int *__restrict__ rowptr, const int *__restrict__ col, const T *__restrict__ val, long long n_rows,
                                const int *__restrict__ doff, const int *__restrict__ dptr, T *__restrict__ dval)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const long long b = r / MIK_BLOCK;
    const int t = (int)(r % MIK_BLOCK);
    const int ns = (dptr[b + 1] - dptr[b]) / MIK_BLOCK;
    const int *so = doff + b * 8;
    int q = 0;
    for (int k2 = rowptr[r]; k2 < rowptr[r + 1]; ++k2) {
        const int d = col[k2] - (int)r;
        while (q < ns && so[q] != d) ++q;
        if (q >= ns) break;
        dval[(size_t)dptr[b] + (size_t)q * MIK_BLOCK + t] = val[k2];
    }
}

struct Scratch {                                   // device temporaries of one upload, freed on every exit
    std::vector<void *> ptrs;
    ~Scratch() { for (void *p : ptrs) if (p) (void)hipFree(p); }
    template <typename P> hipError_t alloc(P **p, size_t bytes)
    {
        hipError_t e = hipMalloc((void **)p, bytes ? bytes : 8);
        if (e == hipSuccess) ptrs.push_back((void *)*p);
        return e;
    }
    void keep(void *p) { for (auto &q : ptrs) if (q == p) q = nullptr; }    // ownership moves to the operator
    void release(void *p) { for (auto &q : ptrs) if (q == p && p) { (void)hipFree(p); q = nullptr; } }   // done with it early
};

inline unsigned blocks_for(long long n) { return (unsigned)((n + MIK_BLOCK - 1) / MIK_BLOCK); }

template <typename T>
int upload_device_t(mik_ctx *ctx, mik_csr *A, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *ptr, const int64_t *idx, const void *val,
                    int index_base, int is_csc)
{
    const size_t es = sizeof(T);
    const int64_t n_major = is_csc ? n_cols : n_rows, n_minor = is_csc ? n_rows : n_cols;
    const int64_t nb = (n_rows + MIK_BLOCK - 1) / MIK_BLOCK;
    hipStream_t st = ctx->stream;
    Scratch S;
    hipError_t e;
    long long *d_ptr = nullptr, *d_idx = nullptr;
    T *d_val = nullptr;
    UploadStats *d_st = nullptr;
    int *cursor = nullptr, *scan_tmp = nullptr;
    const size_t pad = 2 * MIK_SPMV_TILE;          // slack so tile-granular reads never leave the allocation
    const size_t scan_ints = (size_t)(std::max<int64_t>(n_rows, nb) / SCAN_TILE + 1) * 2 + 64;
#define UP_TRY(call) do { if ((e = (call)) != hipSuccess) goto hip_fail; } while (0)
    UploadStats hs;
    int long_row;
    int rc = MIK_OK;
    UP_TRY(S.alloc(&d_ptr, sizeof(long long) * ((size_t)n_major + 1)));
    UP_TRY(S.alloc(&d_idx, sizeof(long long) * (size_t)nnz));
    UP_TRY(S.alloc(&d_val, es * (size_t)nnz));
    UP_TRY(S.alloc(&d_st, sizeof(UploadStats)));
    UP_TRY(S.alloc(&scan_tmp, sizeof(int) * scan_ints));
    UP_TRY(hipMalloc((void **)&A->rowptr, sizeof(int) * ((size_t)n_rows + 1 + 256)));
    UP_TRY(hipMalloc((void **)&A->col, sizeof(int) * ((size_t)nnz + pad)));
    UP_TRY(hipMalloc(&A->val, es * ((size_t)nnz + pad)));
    UP_TRY(hipMemsetAsync(d_st, 0, sizeof(UploadStats), st));
    UP_TRY(hipMemsetAsync(A->rowptr, 0, sizeof(int) * ((size_t)n_rows + 1 + 256), st));
    UP_TRY(hipMemsetAsync(A->col + nnz, 0, sizeof(int) * pad, st));
    UP_TRY(hipMemsetAsync((unsigned char *)A->val + es * (size_t)nnz, 0, es * pad, st));
    UP_TRY(hipMemcpyAsync(d_ptr, ptr, sizeof(long long) * ((size_t)n_major + 1), hipMemcpyDefault, st));
    UP_TRY(hipMemcpyAsync(d_idx, idx, sizeof(long long) * (size_t)nnz, hipMemcpyDefault, st));
    UP_TRY(hipMemcpyAsync(d_val, val, es * (size_t)nnz, hipMemcpyDefault, st));
    hipLaunchKernelGGL(k_up_check_ptr, dim3(blocks_for(n_major)), dim3(MIK_BLOCK), 0, st, d_ptr, (long long)n_major, d_st);
    if (is_csc) {
        UP_TRY(S.alloc(&cursor, sizeof(int) * ((size_t)n_rows + 1)));
        UP_TRY(hipMemsetAsync(cursor, 0, sizeof(int) * ((size_t)n_rows + 1), st));
        hipLaunchKernelGGL(k_up_count_rows, dim3(blocks_for(nnz)), dim3(MIK_BLOCK), 0, st, d_idx, (long long)nnz, (long long)n_minor, index_base, A->rowptr, d_st);
    } else {
        hipLaunchKernelGGL(k_up_rowptr, dim3(blocks_for(n_rows + 1)), dim3(MIK_BLOCK), 0, st, d_ptr, (long long)n_rows, index_base, A->rowptr);
        hipLaunchKernelGGL((k_up_convert<T>), dim3(blocks_for(nnz)), dim3(MIK_BLOCK), 0, st, d_idx, d_val, (long long)nnz, (long long)n_minor, index_base, A->col,
                           (T *)A->val, d_st);
    }
    // nothing below may run on a pointer array that is not monotone or on indices out of range
    UP_TRY(hipMemcpyAsync(&hs, d_st, sizeof(hs), hipMemcpyDeviceToHost, st));
    UP_TRY(hipStreamSynchronize(st));
    if (hs.bad_ptr) { rc = mik_fail(ctx, MIK_ERR_INVALID, "mik_csr_create: ptr not monotone"); goto give_up; }
    if (hs.bad_idx) { rc = mik_fail(ctx, MIK_ERR_INVALID, "mik_csr_create: index out of range"); goto give_up; }
    if (is_csc) {
        UP_TRY(device_exclusive_scan(st, A->rowptr, n_rows, scan_tmp));
        hipLaunchKernelGGL((k_up_scatter<T>), dim3(blocks_for(n_cols)), dim3(MIK_BLOCK), 0, st, d_ptr, d_idx, d_val, (long long)n_cols, index_base, A->rowptr,
                           cursor, A->col, (T *)A->val);
    }
    if (is_csc)
        hipLaunchKernelGGL((k_up_sort_rows<T>), dim3(blocks_for(n_rows)), dim3(MIK_BLOCK), 0, st, A->rowptr, (long long)n_rows, A->col, (T *)A->val, d_st);
    hipLaunchKernelGGL(k_up_stats, dim3((unsigned)nb), dim3(MIK_BLOCK), 0, st, A->rowptr, A->col, (long long)n_rows, d_st);
    UP_TRY(hipMemcpyAsync(&hs, d_st, sizeof(hs), hipMemcpyDeviceToHost, st));
    UP_TRY(hipStreamSynchronize(st));
    S.release(d_ptr); S.release(d_idx); S.release(d_val); S.release(cursor);       // the raw copy (2 GB at 256^3) is consumed
    long_row = MIK_LONG_ROW;
    // the host path's business: duplicates, rows to split off -- and ANY row beyond 256 entries, which k_up_sort_rows does not sort
    // (an unsorted row would break the ascending-column order = Julia's scatter order)
    if (hs.dup || hs.max_row > long_row || hs.max_row > 256) { rc = MIK_ERR_NOTIMPL; goto give_up; }
    A->max_row_nnz = hs.max_row;
    A->max_rowblock_nnz = hs.max_rb;
    A->strip = mik_strip_for(ctx, hs.bw, nb);
    // ---- the per-slice-offset layouts -------------------------------------------------------------------------
    if ((ctx->tuning[MIK_KNOB_LAYOUTS] & 1) == 0 && (ctx->tuning[MIK_KNOB_LAYOUTS] & 4) == 0 && n_cols > 0) {
        int *doff = nullptr, *dtri = nullptr, *dptr = nullptr, *first_row = nullptr;
        unsigned char *mask = nullptr;
        SdiaPattern<T> *desc = nullptr;
        int slots_total = 0;
        UP_TRY(S.alloc(&doff, sizeof(int) * (size_t)nb * 8));
        UP_TRY(S.alloc(&dtri, sizeof(int) * (size_t)nb));
        UP_TRY(S.alloc(&dptr, sizeof(int) * ((size_t)nb + 1)));
        UP_TRY(S.alloc(&first_row, sizeof(int) * (size_t)nb * 8));
        UP_TRY(S.alloc(&mask, (size_t)n_rows));
        UP_TRY(hipMemsetAsync(first_row, 0x7f, sizeof(int) * (size_t)nb * 8, st));       // 0x7f7f7f7f: "no row yet"
        hipLaunchKernelGGL(k_up_slice_pattern, dim3(blocks_for(nb)), dim3(MIK_BLOCK), 0, st, A->rowptr, A->col, (long long)n_rows, (long long)nb, doff, dtri, dptr,
                           d_st);
        UP_TRY(device_exclusive_scan(st, dptr, nb, scan_tmp));
        UP_TRY(hipMemcpyAsync(&slots_total, dptr + nb, sizeof(int), hipMemcpyDeviceToHost, st));
        UP_TRY(hipMemcpyAsync(&hs, d_st, sizeof(hs), hipMemcpyDeviceToHost, st));
        UP_TRY(hipStreamSynchronize(st));
        // (the scan is in 32 bits: more than 2^31 slots cannot qualify anyway, and nnz < 2^31 bounds slots by the test below only if it did not wrap)
        if (!hs.sdia_bad && slots_total >= 0 && (int64_t)slots_total <= nnz + nnz / 8 + 8 * MIK_BLOCK && (int64_t)nb * 8 * MIK_BLOCK < INT32_MAX) {
            hipLaunchKernelGGL(k_up_row_masks, dim3(blocks_for(n_rows)), dim3(MIK_BLOCK), 0, st, A->rowptr, A->col, (long long)n_rows, doff, dptr, mask, first_row, d_st);
            if ((ctx->tuning[MIK_KNOB_LAYOUTS] & 2) == 0)
                hipLaunchKernelGGL((k_up_row_constancy<T>), dim3(blocks_for(n_rows)), dim3(MIK_BLOCK), 0, st, A->rowptr, A->col, (const T *)A->val, (long long)n_rows,
                                   doff, dptr, first_row, d_st);
            UP_TRY(hipMemcpyAsync(&hs, d_st, sizeof(hs), hipMemcpyDeviceToHost, st));
            UP_TRY(hipStreamSynchronize(st));
            if (!hs.sdia_bad) {
                const bool constant = (ctx->tuning[MIK_KNOB_LAYOUTS] & 2) == 0 && !hs.not_constant;
                if (constant) {
                    std::vector<unsigned char> hdesc((size_t)nb * sizeof(SdiaPattern<T>));
                    UP_TRY(S.alloc(&desc, sizeof(SdiaPattern<T>) * (size_t)nb));
                    hipLaunchKernelGGL((k_up_slice_desc<T>), dim3(blocks_for(nb)), dim3(MIK_BLOCK), 0, st, A->rowptr, A->col, (const T *)A->val, (long long)n_rows,
                                       (long long)nb, doff, dtri, dptr, mask, first_row, desc);
                    UP_TRY(hipMemcpyAsync(hdesc.data(), desc, hdesc.size(), hipMemcpyDeviceToHost, st));
                    UP_TRY(hipStreamSynchronize(st));
                    A->sdia_mask = mask;
                    S.keep(mask);
                    rc = mik_sdiac_finish(ctx, A, hdesc, nb, es, (int64_t)slots_total);
                    if (rc != MIK_OK) goto give_up;
                } else {
                    UP_TRY(hipMalloc(&A->sdia_val, es * (size_t)std::max(slots_total, 1)));
                    UP_TRY(hipMemsetAsync(A->sdia_val, 0, es * (size_t)std::max(slots_total, 1), st));
                    hipLaunchKernelGGL((k_up_row_values<T>), dim3(blocks_for(n_rows)), dim3(MIK_BLOCK), 0, st, A->rowptr, A->col, (const T *)A->val, (long long)n_rows,
                                       doff, dptr, (T *)A->sdia_val);
                    UP_TRY(hipStreamSynchronize(st));
                    A->sdia_ptr = dptr; A->sdia_off = doff; A->sdia_tri = dtri; A->sdia_mask = mask;
                    S.keep(dptr); S.keep(doff); S.keep(dtri); S.keep(mask);
                    A->sdia_entries = slots_total;
                }
            }
        }
    }
    UP_TRY(hipGetLastError());
    return MIK_OK;
hip_fail:
    rc = (e == hipErrorOutOfMemory) ? MIK_ERR_NOTIMPL      // not enough room for the raw copy next to the operator: the host path needs less
                                    : mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create (device upload): %s", hipGetErrorString(e));
give_up:
    (void)hipStreamSynchronize(st);
    (void)hipGetLastError();
    return rc;
#undef UP_TRY
}

}  // namespace

// MIK_OK: A holds the CSR arrays, the statistics and (if the pattern qualifies) a per-slice-offset layout, all built on the
// device.  MIK_ERR_NOTIMPL: take the host path (A's device arrays, if any, are released by the caller through
// mik_csr_destroy-style cleanup of the fields this function sets).  Anything else: the error to return.
int mik_upload_device(mik_ctx *ctx, mik_csr *A, int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *ptr, const int64_t *idx,
                      const void *val, int index_base, int is_csc)
{
    if (dtype == MIK_F64) return upload_device_t<double>(ctx, A, n_rows, n_cols, nnz, ptr, idx, val, index_base, is_csc);
    return upload_device_t<float>(ctx, A, n_rows, n_cols, nnz, ptr, idx, val, index_base, is_csc);
}

// =============================================================================================
// device builders of the layouts for operators WITHOUT a <= 8-offset structure (round 3): the wide slice-constant form
// (k_spmv_sdiaw, <= 32 offsets per slice) and the jagged slices (k_spmv_jds), from the device CSR arrays -- no host pass over
// the entries (the host builders of mik_core.hip took 4.2 s for a 27-point 256^3 operator and 0.4 s for a 62 M-entry FE one)
// =============================================================================================
namespace {

struct SdiawDesc { int ns, pad_; int d[32]; unsigned long long bits[32]; };     // one per slice: sorted offsets and their value bits
struct WideStats { int fail; int pad_; };

template <typename T> __device__ __forceinline__ unsigned long long value_bits(T v);
template <> __device__ __forceinline__ unsigned long long value_bits<double>(double v) { return __builtin_bit_cast(unsigned long long, v); }
template <> __device__ __forceinline__ unsigned long long value_bits<float>(float v) { return (unsigned long long)__builtin_bit_cast(unsigned, v); }
template <typename T> __device__ __forceinline__ bool value_finite(T v) { return v - v == T(0); }

// one thread per slice: the set of (column - row) offsets of its rows with the ONE value each carries (sorted by offset)
template <typename T>
__global__ void k_up_wide_desc(const int *__restrict__ rowptr, const int *__restrict__ col, const T *__restrict__ val, long long n_rows, long long nb,
                               SdiawDesc *__restrict__ desc, WideStats *st)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb || st->fail) return;
    int d[32];
    unsigned long long bits[32];
    int ns = 0;
    const long long r0 = b * MIK_BLOCK, r1 = r0 + MIK_BLOCK < n_rows ? r0 + MIK_BLOCK : n_rows;
    for (long long r = r0; r < r1; ++r)
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
            const int dd = col[k] - (int)r;
            const T v = val[k];
            int q = 0;
            while (q < ns && d[q] != dd) ++q;
            if (q == ns) {
                if (ns == 32 || !value_finite(v)) { st->fail = 1; return; }
                d[ns] = dd; bits[ns] = value_bits<T>(v); ++ns;
            } else if (bits[q] != value_bits<T>(v)) { st->fail = 1; return; }
        }
    for (int i = 1; i < ns; ++i) {                       // insertion sort by offset
        const int dk = d[i];
        const unsigned long long bk = bits[i];
        int j = i - 1;
        while (j >= 0 && d[j] > dk) { d[j + 1] = d[j]; bits[j + 1] = bits[j]; --j; }
        d[j + 1] = dk; bits[j + 1] = bk;
    }
    SdiawDesc o;
    o.ns = ns; o.pad_ = 0;
    for (int q = 0; q < 32; ++q) { o.d[q] = q < ns ? d[q] : 0; o.bits[q] = q < ns ? bits[q] : 0ull; }
    desc[b] = o;
}

// one thread per row: the presence mask over its slice's sorted offsets (columns ascend within a row, and so do the slots)
__global__ void k_up_wide_masks(const int *__restrict__ rowptr, const int *__restrict__ col, long long n_rows, const SdiawDesc *__restrict__ desc,
                                unsigned *__restrict__ mask, WideStats *st)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const SdiawDesc *ds = desc + r / MIK_BLOCK;
    const int ns = ds->ns;
    unsigned m = 0;
    int q = 0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int dd = col[k] - (int)r;
        while (q < ns && ds->d[q] != dd) ++q;
        if (q == ns || ((m >> q) & 1u)) { st->fail = 1; return; }     // unsorted or duplicate columns: not this layout
        m |= 1u << q;
    }
    mask[r] = m;
}

// 128-bit hash of a slice description (two independent 64-bit mixes): equal descriptions hash alike; the assignment of pattern
// numbers by hash is VERIFIED word for word by k_up_wide_verify
__device__ __forceinline__ unsigned long long mix64(unsigned long long h, unsigned long long x, unsigned long long k)
{
    h ^= x * k; h = (h << 27) | (h >> 37); h *= 0x9E3779B97F4A7C15ull; return h ^ (h >> 31);
}
__global__ void k_up_wide_hash(const SdiawDesc *__restrict__ desc, long long nb, unsigned long long *__restrict__ hash)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const SdiawDesc *ds = desc + b;
    unsigned long long h1 = 0x243F6A8885A308D3ull ^ (unsigned long long)ds->ns, h2 = 0x13198A2E03707344ull + (unsigned long long)ds->ns;
    for (int q = 0; q < ds->ns; ++q) {
        h1 = mix64(h1, (unsigned long long)(unsigned)ds->d[q], 0xBF58476D1CE4E5B9ull); h1 = mix64(h1, ds->bits[q], 0x94D049BB133111EBull);
        h2 = mix64(h2, ds->bits[q], 0xD6E8FEB86659FD93ull); h2 = mix64(h2, (unsigned long long)(unsigned)ds->d[q], 0xA0761D6478BD642Full);
    }
    hash[2 * b] = h1; hash[2 * b + 1] = h2;
}
__global__ void k_up_wide_verify(const SdiawDesc *__restrict__ desc, long long nb, const int *__restrict__ pat_id, const int *__restrict__ rep, WideStats *st)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const SdiawDesc *a = desc + b, *c = desc + rep[pat_id[b]];
    bool same = a->ns == c->ns;
    for (int q = 0; q < 32 && same; ++q) same = a->d[q] == c->d[q] && a->bits[q] == c->bits[q];
    if (!same) st->fail = 1;
}

// ---- jagged slices ----------------------------------------------------------------------------------------------------------------
struct JdsStats { unsigned long long iters; int maxlen, pad_; };

// one wave per 64-row slice: groups of the slice, its longest row (in groups), the row lengths
template <int W>
__global__ __launch_bounds__(MIK_BLOCK) void k_up_jds_count(const int *__restrict__ rowptr, long long n_rows, long long nsl, int *__restrict__ sgroups,
                                                            unsigned short *__restrict__ jlen, JdsStats *st)
{
    const long long s = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (s >= nsl) return;
    const long long r = s * 64 + lane;
    const int len = r < n_rows ? rowptr[r + 1] - rowptr[r] : 0;
    if (r < n_rows) jlen[r] = (unsigned short)(len < 0xFFFF ? len : 0xFFFE);
    int g = (len + W - 1) / W, mg = g, ml = len;
    for (int off = 32; off >= 1; off >>= 1) { g += __shfl_down(g, off); mg = max(mg, __shfl_down(mg, off)); ml = max(ml, __shfl_down(ml, off)); }
    if (lane == 0) {
        sgroups[s] = g;
        atomicAdd(&st->iters, (unsigned long long)mg);
        atomicMax(&st->maxlen, ml);
    }
}

// one wave per slice: its rows' groups, pass by pass, lane order (the layout k_spmv_jds walks); padding = the row's last column, value 0
template <typename T>
__global__ __launch_bounds__(MIK_BLOCK) void k_up_jds_fill(const int *__restrict__ rowptr, const int *__restrict__ col, const T *__restrict__ val,
                                                           long long n_rows, long long nsl, const int *__restrict__ jptr, int *__restrict__ jcol,
                                                           T *__restrict__ jval)
{
    constexpr int W = (int)(16 / sizeof(T));
    const long long s = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (s >= nsl) return;
    const long long r = s * 64 + lane;
    const int k0 = r < n_rows ? rowptr[r] : 0, len = r < n_rows ? rowptr[r + 1] - k0 : 0;
    const int ng = (len + W - 1) / W;
    long long base = jptr[s];
    for (int g = 0;; ++g) {
        const unsigned long long m = __ballot(g < ng);
        if (m == 0ull) break;
        if (g < ng) {
            const long long idx = base + (long long)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const int j = g * W + e;
                jcol[idx * W + e] = col[k0 + (j < len ? j : len - 1)];
                jval[idx * W + e] = j < len ? val[k0 + j] : T(0);
            }
        }
        base += __popcll(m);
    }
}

}  // namespace

int mik_sdiaw_finish(mik_ctx *ctx, mik_csr *A, const std::vector<std::vector<std::pair<int, uint64_t>>> &pats, size_t es, int64_t n_cols);

// Wide slice-constant layout from the device CSR arrays of A.  MIK_OK with A->sdiaw_pats set when the operator qualifies, MIK_OK
// without when it does not; errors otherwise.
template <typename T> static int build_sdiaw_device_t(mik_ctx *ctx, mik_csr *A)
{
    const int64_t n_rows = A->n_rows, nb = (n_rows + MIK_BLOCK - 1) / MIK_BLOCK;
    if ((uint64_t)n_rows * sizeof(T) >= 0xFFFFFFF0ull) return MIK_OK;
    hipStream_t st = ctx->stream;
    Scratch S;
    hipError_t e;
    SdiawDesc *desc = nullptr;
    WideStats *d_st = nullptr, hs;
    unsigned long long *hash = nullptr;
    int *d_rep = nullptr;
    unsigned *mask = nullptr;
    int *pat_id = nullptr;
#define WD_TRY(call) do { if ((e = (call)) != hipSuccess) return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: wide slice patterns: %s", hipGetErrorString(e)); } while (0)
    WD_TRY(S.alloc(&desc, sizeof(SdiawDesc) * (size_t)nb));
    WD_TRY(S.alloc(&d_st, sizeof(WideStats)));
    WD_TRY(hipMemsetAsync(d_st, 0, sizeof(WideStats), st));
    hipLaunchKernelGGL((k_up_wide_desc<T>), dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, A->rowptr, A->col, (const T *)A->val, (long long)n_rows, (long long)nb, desc, d_st);
    WD_TRY(hipMemcpyAsync(&hs, d_st, sizeof(hs), hipMemcpyDeviceToHost, st));
    WD_TRY(hipStreamSynchronize(st));
    if (hs.fail) return MIK_OK;
    WD_TRY(S.alloc(&mask, sizeof(unsigned) * (size_t)n_rows));
    WD_TRY(S.alloc(&hash, sizeof(unsigned long long) * 2 * (size_t)nb));
    hipLaunchKernelGGL(k_up_wide_masks, dim3(blocks_for(n_rows)), dim3(MIK_BLOCK), 0, st, A->rowptr, A->col, (long long)n_rows, desc, mask, d_st);
    hipLaunchKernelGGL(k_up_wide_hash, dim3(blocks_for(nb)), dim3(MIK_BLOCK), 0, st, desc, (long long)nb, hash);
    std::vector<unsigned long long> hh(2 * (size_t)nb);
    WD_TRY(hipMemcpyAsync(hh.data(), hash, sizeof(unsigned long long) * hh.size(), hipMemcpyDeviceToHost, st));
    WD_TRY(hipMemcpyAsync(&hs, d_st, sizeof(hs), hipMemcpyDeviceToHost, st));
    WD_TRY(hipStreamSynchronize(st));
    if (hs.fail) return MIK_OK;
    // pattern numbers by hash (first slice with a hash represents it), verified word for word on the device below
    std::map<std::pair<unsigned long long, unsigned long long>, int> idof;
    std::vector<int> pid((size_t)nb), rep;
    for (int64_t b = 0; b < nb; ++b) {
        const auto key = std::make_pair(hh[2 * (size_t)b], hh[2 * (size_t)b + 1]);
        auto it = idof.find(key);
        if (it == idof.end()) {
            if (rep.size() >= 65536) return MIK_OK;
            it = idof.emplace(key, (int)rep.size()).first;
            rep.push_back((int)b);
        }
        pid[(size_t)b] = it->second;
    }
    WD_TRY(S.alloc(&pat_id, sizeof(int) * (size_t)nb));
    WD_TRY(S.alloc(&d_rep, sizeof(int) * rep.size()));
    WD_TRY(hipMemcpyAsync(pat_id, pid.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice, st));
    WD_TRY(hipMemcpyAsync(d_rep, rep.data(), sizeof(int) * rep.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_up_wide_verify, dim3(blocks_for(nb)), dim3(MIK_BLOCK), 0, st, desc, (long long)nb, pat_id, d_rep, d_st);
    std::vector<SdiawDesc> rd(rep.size());
    if (rep.size() <= 256) {
        for (size_t i = 0; i < rep.size(); ++i) WD_TRY(hipMemcpyAsync(&rd[i], desc + rep[i], sizeof(SdiawDesc), hipMemcpyDeviceToHost, st));
    } else {                                              // many distinct patterns (values that change from slice to slice): one copy of everything
        std::vector<SdiawDesc> all((size_t)nb);
        WD_TRY(hipMemcpyAsync(all.data(), desc, sizeof(SdiawDesc) * (size_t)nb, hipMemcpyDeviceToHost, st));
        WD_TRY(hipStreamSynchronize(st));
        for (size_t i = 0; i < rep.size(); ++i) rd[i] = all[(size_t)rep[i]];
    }
    WD_TRY(hipMemcpyAsync(&hs, d_st, sizeof(hs), hipMemcpyDeviceToHost, st));
    WD_TRY(hipStreamSynchronize(st));
    if (hs.fail) return MIK_OK;                           // a hash collision (never seen): the host builders decide
    std::vector<std::vector<std::pair<int, uint64_t>>> plist(rep.size());
    for (size_t i = 0; i < rep.size(); ++i)
        for (int q = 0; q < rd[i].ns; ++q) plist[i].emplace_back(rd[i].d[q], (uint64_t)rd[i].bits[q]);
    const int rcf = mik_sdiaw_finish(ctx, A, plist, sizeof(T), A->n_cols);
    if (rcf == MIK_ERR_NOTIMPL) return MIK_OK;
    if (rcf != MIK_OK) return rcf;
    A->sdiaw_mask = mask; S.keep(mask);
    A->sdiaw_pat_id = pat_id; S.keep(pat_id);
#undef WD_TRY
    return MIK_OK;
}

int mik_build_sdiaw_device(mik_ctx *ctx, mik_csr *A)
{
    if (!A->rowptr || A->n_long || A->sdia_val || A->sdia_pats || A->nnz <= 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 1) != 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 4) != 0) return MIK_OK;
    return A->dtype == MIK_F64 ? build_sdiaw_device_t<double>(ctx, A) : build_sdiaw_device_t<float>(ctx, A);
}

// Jagged slices from the device CSR arrays of A (no split-off long rows on this path); same criterion as the host builder.
template <typename T> static int build_jds_device_t(mik_ctx *ctx, mik_csr *A)
{
    constexpr int W = (int)(16 / sizeof(T));
    const int64_t n_rows = A->n_rows, nsl = (n_rows + 63) / 64;
    hipStream_t st = ctx->stream;
    Scratch S;
    hipError_t e;
    int *jptr = nullptr, *scan_tmp = nullptr, *jcol = nullptr;
    unsigned short *jlen = nullptr;
    T *jval = nullptr;
    JdsStats *d_st = nullptr, hs;
    int total = 0;
#define JD_TRY(call) do { if ((e = (call)) != hipSuccess) return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: jagged slices: %s", hipGetErrorString(e)); } while (0)
    JD_TRY(S.alloc(&jptr, sizeof(int) * ((size_t)nsl + 8)));
    JD_TRY(S.alloc(&jlen, sizeof(unsigned short) * (size_t)n_rows));
    JD_TRY(S.alloc(&d_st, sizeof(JdsStats)));
    JD_TRY(S.alloc(&scan_tmp, sizeof(int) * ((size_t)(nsl / SCAN_TILE + 1) * 2 + 64)));
    JD_TRY(hipMemsetAsync(d_st, 0, sizeof(JdsStats), st));
    hipLaunchKernelGGL((k_up_jds_count<W>), dim3((unsigned)((nsl + 3) / 4)), dim3(MIK_BLOCK), 0, st, A->rowptr, (long long)n_rows, (long long)nsl, jptr, jlen, d_st);
    JD_TRY(device_exclusive_scan(st, jptr, nsl, scan_tmp));
    JD_TRY(hipMemcpyAsync(&total, jptr + nsl, sizeof(int), hipMemcpyDeviceToHost, st));
    JD_TRY(hipMemcpyAsync(&hs, d_st, sizeof(hs), hipMemcpyDeviceToHost, st));
    JD_TRY(hipStreamSynchronize(st));
    const int64_t groups = total, short_nnz = A->nnz;
    if (groups <= 0 || hs.maxlen >= MIK_JDS_LONG || groups * W >= INT32_MAX) return MIK_OK;
    const int64_t jds_bytes = groups * W * (int64_t)(sizeof(T) + 4) + 2 * n_rows, csr_bytes = short_nnz * (int64_t)(sizeof(T) + 4) + 4 * n_rows;
    if ((ctx->tuning[MIK_KNOB_LAYOUTS] & 16) == 0 && !((int64_t)hs.iters * 64 * 4 <= groups * 5 && (hs.maxlen > 32 || jds_bytes * 10 <= csr_bytes * 11))) return MIK_OK;
    const size_t pad = 64 * 4 * (size_t)W;                  // = 64 * MIK_JDS_U * W: lanes without a group read (and gather through) the tail
    JD_TRY(S.alloc(&jcol, sizeof(int) * ((size_t)groups * W + pad)));
    JD_TRY(S.alloc(&jval, sizeof(T) * ((size_t)groups * W + pad)));
    JD_TRY(hipMemsetAsync(jcol + (size_t)groups * W, 0, sizeof(int) * pad, st));
    JD_TRY(hipMemsetAsync(jval + (size_t)groups * W, 0, sizeof(T) * pad, st));
    for (int q = 1; q < 8; ++q) JD_TRY(hipMemcpyAsync(jptr + nsl + q, jptr + nsl, sizeof(int), hipMemcpyDeviceToDevice, st));    // waves of the last workgroup beyond the last slice
    hipLaunchKernelGGL((k_up_jds_fill<T>), dim3((unsigned)((nsl + 3) / 4)), dim3(MIK_BLOCK), 0, st, A->rowptr, A->col, (const T *)A->val, (long long)n_rows,
                       (long long)nsl, jptr, jcol, jval);
    JD_TRY(hipStreamSynchronize(st));
    A->jds_ptr = jptr; S.keep(jptr);
    A->jds_len = jlen; S.keep(jlen);
    A->jds_col = jcol; S.keep(jcol);
    A->jds_val = jval; S.keep(jval);
    A->jds_groups = groups;
    A->jds_short_nnz = short_nnz;
#undef JD_TRY
    return MIK_OK;
}

int mik_build_jds_device(mik_ctx *ctx, mik_csr *A)
{
    if (!A->rowptr || A->n_long || A->sdia_val || A->sdia_pats || A->sdiaw_pats || A->nnz <= 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 1) != 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 8) != 0)
        return MIK_OK;
    return A->dtype == MIK_F64 ? build_jds_device_t<double>(ctx, A) : build_jds_device_t<float>(ctx, A);
}


// ---- windows of x for the product-tile kernel (csr_build_xwin in mik_core.hip is the host form of the same rule) -------------
namespace {
__global__ __launch_bounds__(MIK_BLOCK) void k_up_xwin(const int *__restrict__ rowptr, const int *__restrict__ col, long long n_rows, int *__restrict__ mn_out,
                                                       int *__restrict__ mx_out, int *__restrict__ cnt_out)
{
    __shared__ int smn[MIK_BLOCK], smx[MIK_BLOCK];
    const long long r0 = (long long)blockIdx.x * MIK_BLOCK, r1 = r0 + MIK_BLOCK < n_rows ? r0 + MIK_BLOCK : n_rows;
    const int ka = rowptr[r0], kb = rowptr[r1];
    int mn = INT32_MAX, mx = -1;
    for (int k = ka + (int)threadIdx.x; k < kb; k += MIK_BLOCK) { const int c = col[k]; mn = c < mn ? c : mn; mx = c > mx ? c : mx; }
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
    __syncthreads();
    for (int off = MIK_BLOCK / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            smn[threadIdx.x] = smn[threadIdx.x] < smn[threadIdx.x + off] ? smn[threadIdx.x] : smn[threadIdx.x + off];
            smx[threadIdx.x] = smx[threadIdx.x] > smx[threadIdx.x + off] ? smx[threadIdx.x] : smx[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { mn_out[blockIdx.x] = smn[0]; mx_out[blockIdx.x] = smx[0]; cnt_out[blockIdx.x] = kb - ka; }
}
}  // namespace

int mik_build_xwin_device(mik_ctx *ctx, mik_csr *A)
{
    if (!A->rowptr || A->n_long || A->sdia_val || A->sdia_pats || A->sdiaw_pats || A->jds_val || A->nnz <= 0 || A->n_rows <= 0 || (ctx->tuning[MIK_KNOB_LAYOUTS] & 1) != 0 ||
        (ctx->tuning[MIK_KNOB_LAYOUTS] & 32) != 0 || A->max_row_nnz <= 32)
        return MIK_OK;
    {   // the row permutation of the product tile (irregular rows): from a host copy of the row pointer (4 B per row, once)
        std::vector<int> rp((size_t)A->n_rows + 1);
        MIK_HIP(ctx, hipMemcpy(rp.data(), A->rowptr, sizeof(int) * rp.size(), hipMemcpyDeviceToHost));
        MIK_TRY(mik_build_rperm_host(ctx, A, rp.data()));
    }
    const int64_t nb = (A->n_rows + MIK_BLOCK - 1) / MIK_BLOCK;
    int *d = nullptr;
    MIK_HIP(ctx, hipMalloc((void **)&d, sizeof(int) * 3 * (size_t)nb));
    hipLaunchKernelGGL(k_up_xwin, dim3((unsigned)nb), dim3(MIK_BLOCK), 0, ctx->stream, (const int *)A->rowptr, (const int *)A->col, (long long)A->n_rows, d, d + nb, d + 2 * nb);
    std::vector<int> h((size_t)(3 * nb));
    hipError_t e = hipMemcpyAsync(h.data(), d, sizeof(int) * h.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return mik_fail(ctx, MIK_ERR_HIP, "mik_csr_create: window statistics: %s", hipGetErrorString(e));
    std::vector<int> lo;
    int span = 0;
    if (!mik_xwin_plan(nb, h.data(), h.data() + nb, h.data() + 2 * nb, mik_dtype_size(A->dtype), A->n_cols, A->nnz, lo, &span)) return MIK_OK;
    if ((e = hipMalloc((void **)&A->xwin_lo, sizeof(int) * (size_t)nb)) != hipSuccess ||
        (e = hipMemcpy(A->xwin_lo, lo.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice)) != hipSuccess)
        return mik_fail(ctx, e == hipErrorOutOfMemory ? MIK_ERR_NOMEM : MIK_ERR_HIP, "mik_csr_create: window table: %s", hipGetErrorString(e));
    A->xwin_span = span;
    return MIK_OK;
}
