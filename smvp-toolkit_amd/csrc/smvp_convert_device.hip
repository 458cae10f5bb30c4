// smvp_convert_device.hip -- COO -> CSR and COO -> TJDS on the GPU.
//
// Same outputs, bit for bit, as the host converters in smvp_convert.cpp (and so
// as the reference's main-cli.c:340-365 and :766-967), for inputs that are
// already in HBM or too large to convert comfortably on one host thread: the
// reference's own TJDS build is O(nnz * cols) (main-cli.c:894-904), the host
// converter here takes 15 s for 119 M entries, this path tens of milliseconds.
//
// The heavy lifting is two stable LSD radix sorts (smvp_prim.h: hand-written, 8 bits per pass --
// setup work, not the timed product) on keys packed as major * 2^bits(minor) + minor,
// with the entry's input position as the value, so ties keep input order exactly
// like the host's stable counting sorts.
#include "smvp_common.h"
#include "smvp_prim.h"

#include <hip/hip_runtime.h>

#include <cstring>

#include <algorithm>
#include <vector>

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return smvp::fail(SMVP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

typedef unsigned long long u64;

// Frees everything it handed out when it goes out of scope.
struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch()
    {
        for (void *p : ptrs)
            (void)hipFree(p);
    }
    template <class T>
    hipError_t get(T **out, size_t count)
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 4) * sizeof(T));
        if (e == hipSuccess)
            ptrs.push_back(p);
        *out = (T *)p;
        return e;
    }
};

int bits_for(int n)  // bits needed for values 0 .. n-1
{
    int b = 1;
    while (b < 31 && (1ll << b) < n)
        ++b;
    return b;
}

__global__ __launch_bounds__(256) void pack_keys(const smvp_coo_t *__restrict__ coo, int nnz, int rows, int cols,
                                                 int minor_bits, bool row_major, u64 *__restrict__ keys,
                                                 unsigned *__restrict__ idx, int *__restrict__ major_count,
                                                 int *__restrict__ bad)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz)
        return;
    const int r = coo[i].row, c = coo[i].col;
    if (r < 0 || r >= rows || c < 0 || c >= cols) {
        atomicExch(bad, i + 1);
        keys[i] = 0;
        idx[i] = (unsigned)i;
        return;
    }
    const int major = row_major ? r : c, minor = row_major ? c : r;
    keys[i] = ((u64)(unsigned)major << minor_bits) | (u64)(unsigned)minor;
    idx[i] = (unsigned)i;
    atomicAdd(&major_count[major], 1);
}

__global__ __launch_bounds__(256) void gather_csr(const smvp_coo_t *__restrict__ coo, const unsigned *__restrict__ order,
                                                  int nnz, int *__restrict__ col_ind, double *__restrict__ val)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nnz)
        return;
    const smvp_coo_t e = coo[order[t]];
    col_ind[t] = e.col;
    val[t] = e.val;
}

__global__ __launch_bounds__(256) void column_lengths(const int *__restrict__ start, int cols, int *__restrict__ len)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < cols)
        len[c] = start[c + 1] - start[c];
}

__global__ __launch_bounds__(256) void length_keys(const int *__restrict__ col_len, int cols, unsigned *__restrict__ key,
                                                   unsigned *__restrict__ col)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols)
        return;
    key[c] = 0x7fffffffu - (unsigned)col_len[c];  // ascending key == descending length
    col[c] = (unsigned)c;
}

__global__ __launch_bounds__(256) void invert_perm(const unsigned *__restrict__ perm_u, int cols, int *__restrict__ perm,
                                                   int *__restrict__ where)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cols)
        return;
    const int c = (int)perm_u[k];
    perm[k] = c;
    where[c] = k;
}

// width[d] = number of columns longer than d = first k with len_sorted[k] <= d (lengths never increase with k)
__global__ __launch_bounds__(256) void diagonal_widths(const unsigned *__restrict__ len_key_sorted, int cols, int longest,
                                                       int *__restrict__ width)
{
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d > longest)
        return;
    if (d == longest) {
        width[d] = 0;
        return;
    }
    int lo = 0, hi = cols;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int len = (int)(0x7fffffffu - len_key_sorted[mid]);
        if (len > d)
            lo = mid + 1;
        else
            hi = mid;
    }
    width[d] = lo;
}

__global__ __launch_bounds__(256) void scatter_tjds(const smvp_coo_t *__restrict__ coo, const unsigned *__restrict__ order,
                                                    int nnz, const int *__restrict__ col_start,
                                                    const int *__restrict__ where, const int *__restrict__ start_pos,
                                                    int *__restrict__ row_ind, double *__restrict__ val)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nnz)
        return;
    const smvp_coo_t e = coo[order[t]];
    const int d = t - col_start[e.col];  // rank inside the column == jagged-diagonal number
    const int j = start_pos[d] + where[e.col];
    row_ind[j] = e.row;
    val[j] = e.val;
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

// order[] = input positions sorted by (major, minor), ties in input order; major_start[] = exclusive scan of counts
int sort_entries(Scratch &sc, const smvp_coo_t *d_coo, int rows, int cols, int nnz, bool row_major, hipStream_t st,
                 unsigned **order_out, int **major_start_out)
{
    const int n_major = row_major ? rows : cols, n_minor = row_major ? cols : rows;
    const int minor_bits = bits_for(n_minor), major_bits = bits_for(n_major);
    u64 *k0, *k1;
    unsigned *i0, *i1;
    int *count, *start, *bad;
    HIP_TRY(sc.get(&k0, (size_t)nnz));
    HIP_TRY(sc.get(&k1, (size_t)nnz));
    HIP_TRY(sc.get(&i0, (size_t)nnz));
    HIP_TRY(sc.get(&i1, (size_t)nnz));
    HIP_TRY(sc.get(&count, (size_t)n_major + 1));
    HIP_TRY(sc.get(&start, (size_t)n_major + 1));
    HIP_TRY(sc.get(&bad, 1));
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * ((size_t)n_major + 1), st));
    HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), st));
    if (nnz > 0) {
        hipLaunchKernelGGL(pack_keys, dim3(blocks_for(nnz)), dim3(256), 0, st, d_coo, nnz, rows, cols, minor_bits,
                           row_major, k0, i0, count, bad);
        HIP_TRY(hipGetLastError());
        size_t tmp_bytes = 0;
        HIP_TRY(smvp::prim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, i0, i1, (size_t)nnz, 0u,
                                          (unsigned)(minor_bits + major_bits), st));
        char *tmp;
        HIP_TRY(sc.get(&tmp, tmp_bytes));
        HIP_TRY(smvp::prim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, i0, i1, (size_t)nnz, 0u,
                                          (unsigned)(minor_bits + major_bits), st));
    }
    {
        size_t tmp_bytes = 0;
        HIP_TRY(smvp::prim::exclusive_scan(nullptr, tmp_bytes, count, start, 0, (size_t)n_major + 1, st));
        char *tmp;
        HIP_TRY(sc.get(&tmp, tmp_bytes));
        HIP_TRY(smvp::prim::exclusive_scan(tmp, tmp_bytes, count, start, 0, (size_t)n_major + 1, st));
    }
    int h_bad = 0;
    HIP_TRY(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_bad)
        return smvp::fail(SMVP_ERR_INVALID, "device conversion: entry %d lies outside %d x %d", h_bad - 1, rows, cols);
    *order_out = i1;
    *major_start_out = start;
    return SMVP_OK;
}

int check_device(const void *p, const char *who)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return smvp::fail(SMVP_ERR_NO_DEVICE, "%s: no HIP device is visible", who);
    (void)p;
    return SMVP_OK;
}

}  // namespace

namespace {

__global__ __launch_bounds__(256) void row_keys(const int *__restrict__ row_ind, int nnz, unsigned *__restrict__ key,
                                                unsigned *__restrict__ pos, int *__restrict__ row_count)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nnz)
        return;
    const int r = row_ind[j];
    key[j] = (unsigned)r;
    pos[j] = (unsigned)j;
    atomicAdd(&row_count[r], 1);
}

}  // namespace

namespace smvp {

// Row-inverted index of a scattered product (TJDS): inv_pos[inv_ptr[r] .. inv_ptr[r+1]) lists, in ascending
// order, the positions j with row_ind[j] == r.  One stable radix sort of (row, position) + histogram + scan.
// row_ind must already be range-checked (it indexes the histogram).
int build_row_inverse(const int *d_row_ind, int nnz, int rows, int *d_inv_ptr, int *d_inv_pos, hipStream_t st)
{
    Scratch sc;
    unsigned *k0, *k1, *p0;
    int *count;
    HIP_TRY(sc.get(&k0, (size_t)nnz));
    HIP_TRY(sc.get(&k1, (size_t)nnz));
    HIP_TRY(sc.get(&p0, (size_t)nnz));
    HIP_TRY(sc.get(&count, (size_t)rows + 1));
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * ((size_t)rows + 1), st));
    if (nnz > 0) {
        hipLaunchKernelGGL(row_keys, dim3(blocks_for(nnz)), dim3(256), 0, st, d_row_ind, nnz, k0, p0, count);
        HIP_TRY(hipGetLastError());
        size_t tmp_bytes = 0;
        HIP_TRY(smvp::prim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, p0, (unsigned *)d_inv_pos, (size_t)nnz, 0u,
                                          (unsigned)bits_for(std::max(rows, 2)), st));
        char *tmp;
        HIP_TRY(sc.get(&tmp, tmp_bytes));
        HIP_TRY(smvp::prim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, p0, (unsigned *)d_inv_pos, (size_t)nnz, 0u,
                                          (unsigned)bits_for(std::max(rows, 2)), st));
    }
    size_t tmp_bytes = 0;
    HIP_TRY(smvp::prim::exclusive_scan(nullptr, tmp_bytes, count, d_inv_ptr, 0, (size_t)rows + 1, st));
    char *tmp;
    HIP_TRY(sc.get(&tmp, tmp_bytes));
    HIP_TRY(smvp::prim::exclusive_scan(tmp, tmp_bytes, count, d_inv_ptr, 0, (size_t)rows + 1, st));
    HIP_TRY(hipStreamSynchronize(st));
    return SMVP_OK;
}

}  // namespace smvp

namespace {

// jagged diagonal d of TJDS position p: start_pos[d] <= p < start_pos[d + 1]; permuted column k = p - start_pos[d]
__global__ __launch_bounds__(256) void positions_to_columns(const int *__restrict__ pos, int nnz,
                                                            const int *__restrict__ start_pos, int num_diag,
                                                            int *__restrict__ kcol)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz)
        return;
    const int p = pos[e];
    int lo = 0, hi = num_diag - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (start_pos[mid] <= p)
            lo = mid;
        else
            hi = mid - 1;
    }
    kcol[e] = p - start_pos[lo];
}

}  // namespace

namespace {

// where[p] = place e of TJDS position p in the row-major stream (pos[e] = p)
__global__ __launch_bounds__(256) void invert_positions(const int *__restrict__ pos, int nnz, int *__restrict__ where)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < nnz)
        where[pos[e]] = e;
}

// One thread per 128-byte line of val (16 positions): how many different tiles of the row-major stream do its entries
// belong to?  A line whose entries scatter over `min_tiles` tiles or more would be pulled through the L2 once per tile
// for 8 useful bytes each time (the long columns' entries far down the jagged diagonals: neighbours in TJDS order,
// unrelated rows); its entries are marked and get their values from the tiles' own cache instead.
__global__ __launch_bounds__(256) void mark_scattered_lines(const int *__restrict__ where, int nnz, int tile, int min_tiles,
                                                            unsigned char *__restrict__ line_flag)
{
    const int line = blockIdx.x * 256 + threadIdx.x;
    const long long p0 = (long long)line * 16;
    if (p0 >= nnz)
        return;
    int owner[16];
    int n = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (p0 + i < nnz)
            owner[n++] = where[p0 + i] / tile;
    int distinct = 0;
    for (int i = 0; i < n; ++i) {
        bool seen = false;
        for (int j = 0; j < i; ++j)
            seen = seen || owner[j] == owner[i];
        distinct += seen ? 0 : 1;
    }
    line_flag[line] = distinct >= min_tiles ? 1 : 0;
}

__device__ __forceinline__ int diagonal_of(const int *__restrict__ start_pos, int num_diag, int p)
{
    int lo = 0, hi = num_diag - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (start_pos[mid] <= p)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

// sort key of stream entry e: (tile, cached?, TJDS position) -- a tile's in-place entries first, in TJDS order, then
// its cached ones (by_column: those by their permuted column instead, which is all the product needs of them)
__global__ __launch_bounds__(256) void window_keys(const int *__restrict__ pos, int nnz, int tile,
                                                   const unsigned char *__restrict__ line_flag, const int *__restrict__ start_pos,
                                                   int num_diag, int by_column, u64 *__restrict__ key,
                                                   unsigned *__restrict__ slot, int *__restrict__ cached_count)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz)
        return;
    const unsigned p = (unsigned)pos[e];
    const unsigned cached = line_flag ? line_flag[p >> 4] : 0u;
    const unsigned word = cached && by_column ? p - (unsigned)start_pos[diagonal_of(start_pos, num_diag, (int)p)] : p;
    key[e] = ((u64)(unsigned)(e / tile) << 33) | ((u64)cached << 32) | word;
    slot[e] = (unsigned)(e % tile);
    if (cached)
        atomicAdd(&cached_count[e / tile], 1);
}

// in-place entry: (TJDS position, slot | diagonal << slot_bits); cached entry: (permuted column, slot) -- diagonal 0, so
// that the kernel's "position - start_pos[diagonal]" is the column for both -- and its value into the tile's cache
__global__ __launch_bounds__(256) void window_streams(const u64 *__restrict__ key, const unsigned *__restrict__ slot, int nnz,
                                                      int tile, const int *__restrict__ start_pos, int num_diag, int slot_bits,
                                                      const int *__restrict__ cache_ptr, const double *__restrict__ val,
                                                      int *__restrict__ pos_sorted, int *__restrict__ meta,
                                                      double *__restrict__ val_cache)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz)
        return;
    const int p = (int)(unsigned)(key[e] & 0xffffffffu);
    const int d = diagonal_of(start_pos, num_diag, p);
    if ((key[e] >> 32) & 1u) {
        const int b = e / tile;
        const long long tile_end = (long long)(b + 1) * tile < nnz ? (long long)(b + 1) * tile : nnz;
        const int c0 = cache_ptr[b], c1 = cache_ptr[b + 1];
        const int first_cached = (int)(tile_end - (c1 - c0));  // stream place of the tile's first cached entry
        pos_sorted[e] = p - start_pos[d];
        meta[e] = (int)slot[e];
        val_cache[c0 + (e - first_cached)] = val[p];
    } else {
        pos_sorted[e] = p;
        meta[e] = (int)(slot[e] | ((unsigned)d << slot_bits));
    }
}

__global__ __launch_bounds__(256) void overflow_entries(const int *__restrict__ pos, const int *__restrict__ ovf_ptr,
                                                        int ntiles, int tile, int nnz, const int *__restrict__ start_pos,
                                                        int num_diag, const double *__restrict__ val, double *__restrict__ ovf_val,
                                                        int *__restrict__ ovf_k, int positions)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ovf_ptr[ntiles])
        return;
    int lo = 0, hi = ntiles - 1;  // last tile b with ovf_ptr[b] <= i (tiles without overflow share a start: take the last)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ovf_ptr[mid] <= i)
            lo = mid;
        else
            hi = mid - 1;
    }
    const long long e = (long long)(lo + 1) * tile;  // a tile with overflow entries is a full one
    const int p = pos[e + (i - ovf_ptr[lo])];
    // the entry's VALUE, not its position: the tile that finishes the row reads it coalesced instead of gathering val[position]
    // (2.2 M overflow entries on memplus x944, each a gather of its own: 158 MB of 2.33 GB, profiles/r05_tjds_traffic_by_stream.txt)
    if (positions) {  // (unit operand: the values are written anew before every product -- the tile gathers val[position] itself)
        ovf_k[i] = p;
        return;
    }
    ovf_val[i] = val[p];
    ovf_k[i] = p - start_pos[diagonal_of(start_pos, num_diag, p)];
}

}  // namespace

namespace {

// kFlavorTjdsH, pass 1 over the sorted windows.  A RUN is a stretch of one tile's sorted entries whose 32-bit word -- the TJDS
// position of an in-place entry, the permuted column of a cached one -- lies in one aligned block of 2^16 and, in place, in
// one jagged diagonal: inside a run the word is the run's base + 16 bits.  head[e] = 1 where a run starts, diag[e] = the
// entry's diagonal (-1: cached).
constexpr int kRunSpanBits = 16;

__device__ __forceinline__ void run_class(u64 key, const int *__restrict__ start_pos, int num_diag, int *diag, unsigned *block)
{
    const bool cached = (key >> 32) & 1u;
    const unsigned word = (unsigned)(key & 0xffffffffu);
    *diag = cached ? -1 : diagonal_of(start_pos, num_diag, (int)word);
    *block = word >> kRunSpanBits;
}

__global__ __launch_bounds__(256) void diagonal_run_heads(const u64 *__restrict__ key, int nnz, int tile,
                                                          const int *__restrict__ start_pos, int num_diag,
                                                          int *__restrict__ diag, int *__restrict__ head,
                                                          int *__restrict__ runs_in_tile)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz)
        return;
    int d, pd;
    unsigned blk, pblk;
    run_class(key[e], start_pos, num_diag, &d, &blk);
    int h = 1;
    if (e % tile != 0) {
        run_class(key[e - 1], start_pos, num_diag, &pd, &pblk);
        h = pd != d || pblk != blk ? 1 : 0;
    }
    diag[e] = d;
    head[e] = h;
    if (h)
        atomicAdd(&runs_in_tile[e / tile], 1);
}

// pass 2: per entry ONE 32-bit word: the low 16 bits of its position (column) | (slot | run number mod 32 << 11) << 16; per run {base, sub}: an
// entry's word is base + its 16 bits, its operand x_perm[word - sub] (in place: sub = start_pos of the run's diagonal and
// val[word] the value; cached: the word is the column, sub = 0, the value goes into the tile's run of the cache); per group
// of 32 entries its first entry's run
__global__ __launch_bounds__(256) void half_streams(const u64 *__restrict__ key, const unsigned *__restrict__ slot, int nnz,
                                                    int tile, const int *__restrict__ pos, const int *__restrict__ start_pos,
                                                    const int *__restrict__ diag, const int *__restrict__ head,
                                                    const int *__restrict__ run_id, const int *__restrict__ run_ptr,
                                                    const int *__restrict__ cache_ptr, const double *__restrict__ val,
                                                    unsigned *__restrict__ word32,
                                                    int *__restrict__ run_tab, unsigned short *__restrict__ group_run,
                                                    double *__restrict__ val_cache)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz)
        return;
    const int b = e / tile, idx = e % tile;
    const unsigned word = (unsigned)(key[e] & 0xffffffffu);
    const int d = diag[e];
    const int r = run_id[e] - 1;  // run_id = inclusive scan of head
    const int local = r - run_ptr[b];
    if (d < 0) {  // cached: its value (val at the position the row-major stream holds for its slot) into the tile's run of the cache
        const long long tile_end = (long long)(b + 1) * tile < nnz ? (long long)(b + 1) * tile : nnz;
        const int c0 = cache_ptr[b], c1 = cache_ptr[b + 1];
        val_cache[c0 + (e - (int)(tile_end - (c1 - c0)))] = val[pos[(long long)b * tile + slot[e]]];
    }
    if (head[e]) {
        run_tab[2 * (size_t)r] = (int)(word & ~((1u << kRunSpanBits) - 1u));
        run_tab[2 * (size_t)r + 1] = d < 0 ? 0 : start_pos[d];
    }
    if (idx % 32 == 0)
        group_run[(size_t)b * (tile / 32) + idx / 32] = (unsigned short)local;
    word32[e] = (word & ((1u << kRunSpanBits) - 1u)) | ((slot[e] | ((unsigned)(local & 31) << 11)) << 16);
}

}  // namespace

namespace smvp {

// kFlavorTjdsS: the row-major stream `d_pos` cut into windows of `tile` entries, every window sorted by TJDS position;
// meta = the entry's place in its window before sorting (its LDS slot) | its jagged diagonal << slot_bits.
// cache_min_tiles > 0: entries of val lines that scatter over that many tiles or more (mark_scattered_lines) come last
// in their window and carry (permuted column, slot); their values are copied, window by window, into *d_val_cache
// (allocated here, hipFree by the caller), d_cache_ptr[ntiles + 1] = where each window's run starts.  0: no cache
// (d_cache_ptr all zero, *d_val_cache a dummy).
int sort_tile_windows(const int *d_pos, int nnz, int tile, const int *d_start_pos, int num_diag, int slot_bits,
                      const double *d_val, int cache_min_tiles, int *d_pos_sorted, int *d_meta, int *d_cache_ptr,
                      double **d_val_cache, int *cached_total, hipStream_t st)
{
    const int ntiles = std::max(1, (int)(((long long)nnz + tile - 1) / tile));
    *d_val_cache = nullptr;
    *cached_total = 0;
    HIP_TRY(hipMemsetAsync(d_cache_ptr, 0, sizeof(int) * ((size_t)ntiles + 1), st));
    if (nnz <= 0) {
        HIP_TRY(hipMalloc((void **)d_val_cache, 4 * sizeof(double)));
        HIP_TRY(hipStreamSynchronize(st));
        return SMVP_OK;
    }
    if (tile > (1 << slot_bits) || ((long long)(num_diag - 1) >> (32 - slot_bits)) != 0)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "sort_tile_windows: tile %d / %d diagonals do not fit the packed word", tile, num_diag);
    Scratch sc;
    u64 *k0, *k1;
    unsigned *s0, *s1;
    int *count;
    unsigned char *flag = nullptr;
    HIP_TRY(sc.get(&k0, (size_t)nnz));
    HIP_TRY(sc.get(&k1, (size_t)nnz));
    HIP_TRY(sc.get(&s0, (size_t)nnz));
    HIP_TRY(sc.get(&s1, (size_t)nnz));
    HIP_TRY(sc.get(&count, (size_t)ntiles + 1));
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * ((size_t)ntiles + 1), st));
    if (cache_min_tiles > 0) {
        int *where;
        const int nlines = (nnz + 15) / 16;
        HIP_TRY(sc.get(&where, (size_t)nnz));
        HIP_TRY(sc.get(&flag, (size_t)nlines));
        hipLaunchKernelGGL(invert_positions, dim3(blocks_for(nnz)), dim3(256), 0, st, d_pos, nnz, where);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(mark_scattered_lines, dim3(blocks_for(nlines)), dim3(256), 0, st, where, nnz, tile, cache_min_tiles, flag);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(window_keys, dim3(blocks_for(nnz)), dim3(256), 0, st, d_pos, nnz, tile, flag, d_start_pos, num_diag, 0, k0, s0, count);
    HIP_TRY(hipGetLastError());
    const unsigned bits = 33u + (unsigned)bits_for(ntiles + 1);
    size_t tmp_bytes = 0;
    HIP_TRY(smvp::prim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, s0, s1, (size_t)nnz, 0u, bits, st));
    char *tmp;
    HIP_TRY(sc.get(&tmp, tmp_bytes));
    HIP_TRY(smvp::prim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, s0, s1, (size_t)nnz, 0u, bits, st));
    {
        size_t scan_bytes = 0;
        HIP_TRY(smvp::prim::exclusive_scan(nullptr, scan_bytes, count, d_cache_ptr, 0, (size_t)ntiles + 1, st));
        char *scan_tmp;
        HIP_TRY(sc.get(&scan_tmp, scan_bytes));
        HIP_TRY(smvp::prim::exclusive_scan(scan_tmp, scan_bytes, count, d_cache_ptr, 0, (size_t)ntiles + 1, st));
    }
    int total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_cache_ptr + ntiles, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (hipMalloc((void **)d_val_cache, sizeof(double) * (size_t)std::max(total, 4)) != hipSuccess)
        return smvp::fail(SMVP_ERR_ALLOC, "sort_tile_windows: cannot allocate the value cache (%d entries)", total);
    *cached_total = total;
    hipLaunchKernelGGL(window_streams, dim3(blocks_for(nnz)), dim3(256), 0, st, k1, s1, nnz, tile, d_start_pos, num_diag, slot_bits,
                       d_cache_ptr, d_val, d_pos_sorted, d_meta, *d_val_cache);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return SMVP_OK;
}

// kFlavorTjdsH: like sort_tile_windows, but an entry is two 16-bit words -- the low half of its position (cached: of its
// column) and slot | run hint -- and the rest comes from the tile's run table ({base, sub} per run, see half_streams).
// Allocates *d_val_cache, *d_run_tab (2 ints per run), *d_group_run (hipFree by the caller); d_cache_ptr / d_run_ptr have
// ntiles + 1 entries.
int build_tile_half_streams(const int *d_pos, int nnz, int tile, const int *d_start_pos, int num_diag, const double *d_val,
                            int cache_min_tiles, unsigned *d_word32, int *d_cache_ptr, int *d_run_ptr,
                            double **d_val_cache, int **d_run_tab, unsigned short **d_group_run, int *cached_total,
                            int *runs_total, hipStream_t st)
{
    const int ntiles = std::max(1, (int)(((long long)nnz + tile - 1) / tile));
    *d_val_cache = nullptr, *d_run_tab = nullptr, *d_group_run = nullptr;
    *cached_total = *runs_total = 0;
    if (tile % 32 != 0 || tile > 2048)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "build_tile_half_streams: tile %d does not fit the 16-bit word", tile);
    HIP_TRY(hipMemsetAsync(d_cache_ptr, 0, sizeof(int) * ((size_t)ntiles + 1), st));
    HIP_TRY(hipMemsetAsync(d_run_ptr, 0, sizeof(int) * ((size_t)ntiles + 1), st));
    Scratch sc;
    u64 *k0, *k1;
    unsigned *s0, *s1;
    int *count, *rcount, *dg, *head, *rid;
    unsigned char *flag = nullptr;
    const size_t n = (size_t)std::max(nnz, 1);
    HIP_TRY(sc.get(&k0, n));
    HIP_TRY(sc.get(&k1, n));
    HIP_TRY(sc.get(&s0, n));
    HIP_TRY(sc.get(&s1, n));
    HIP_TRY(sc.get(&dg, n));
    HIP_TRY(sc.get(&head, n));
    HIP_TRY(sc.get(&rid, n));
    HIP_TRY(sc.get(&count, (size_t)ntiles + 1));
    HIP_TRY(sc.get(&rcount, (size_t)ntiles + 1));
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * ((size_t)ntiles + 1), st));
    HIP_TRY(hipMemsetAsync(rcount, 0, sizeof(int) * ((size_t)ntiles + 1), st));
    int total = 0, nruns = 0;
    if (nnz > 0) {
        if (cache_min_tiles > 0) {
            int *where;
            const int nlines = (nnz + 15) / 16;
            HIP_TRY(sc.get(&where, n));
            HIP_TRY(sc.get(&flag, (size_t)nlines));
            hipLaunchKernelGGL(invert_positions, dim3(blocks_for(nnz)), dim3(256), 0, st, d_pos, nnz, where);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(mark_scattered_lines, dim3(blocks_for(nlines)), dim3(256), 0, st, where, nnz, tile, cache_min_tiles, flag);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(window_keys, dim3(blocks_for(nnz)), dim3(256), 0, st, d_pos, nnz, tile, flag, d_start_pos, num_diag, 1, k0, s0, count);
        HIP_TRY(hipGetLastError());
        const unsigned bits = 33u + (unsigned)bits_for(ntiles + 1);
        size_t tmp_bytes = 0;
        HIP_TRY(smvp::prim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, s0, s1, (size_t)nnz, 0u, bits, st));
        char *tmp;
        HIP_TRY(sc.get(&tmp, tmp_bytes));
        HIP_TRY(smvp::prim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, s0, s1, (size_t)nnz, 0u, bits, st));
        hipLaunchKernelGGL(diagonal_run_heads, dim3(blocks_for(nnz)), dim3(256), 0, st, k1, nnz, tile, d_start_pos, num_diag, dg, head, rcount);
        HIP_TRY(hipGetLastError());
        size_t scan_bytes = 0, scan2 = 0;
        HIP_TRY(smvp::prim::exclusive_scan(nullptr, scan_bytes, count, d_cache_ptr, 0, (size_t)ntiles + 1, st));
        HIP_TRY(smvp::prim::inclusive_scan(nullptr, scan2, head, rid, (size_t)nnz, st));
        char *scan_tmp;
        HIP_TRY(sc.get(&scan_tmp, std::max(scan_bytes, scan2)));
        HIP_TRY(smvp::prim::exclusive_scan(scan_tmp, scan_bytes, count, d_cache_ptr, 0, (size_t)ntiles + 1, st));
        HIP_TRY(smvp::prim::exclusive_scan(scan_tmp, scan_bytes, rcount, d_run_ptr, 0, (size_t)ntiles + 1, st));
        HIP_TRY(smvp::prim::inclusive_scan(scan_tmp, scan2, head, rid, (size_t)nnz, st));
        HIP_TRY(hipMemcpyAsync(&total, d_cache_ptr + ntiles, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&nruns, d_run_ptr + ntiles, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    *cached_total = total;
    *runs_total = nruns;
    const size_t groups = (size_t)ntiles * (size_t)(tile / 32);
    if (hipMalloc((void **)d_val_cache, sizeof(double) * (size_t)std::max(total, 4)) != hipSuccess ||
        hipMalloc((void **)d_run_tab, 2 * sizeof(int) * (size_t)std::max(nruns, 4)) != hipSuccess ||
        hipMalloc((void **)d_group_run, sizeof(unsigned short) * std::max<size_t>(groups, 4)) != hipSuccess)
        return smvp::fail(SMVP_ERR_ALLOC, "build_tile_half_streams: cannot allocate the run tables (%d runs, %d cached values)", nruns, total);
    HIP_TRY(hipMemsetAsync(*d_group_run, 0, sizeof(unsigned short) * std::max<size_t>(groups, 4), st));
    if (nnz > 0) {
        hipLaunchKernelGGL(half_streams, dim3(blocks_for(nnz)), dim3(256), 0, st, k1, s1, nnz, tile, d_pos, d_start_pos, dg, head, rid,
                           d_run_ptr, d_cache_ptr, d_val, d_word32, *d_run_tab, *d_group_run, *d_val_cache);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(st));
    return SMVP_OK;
}

// the entries [e_b, tile_next_b) every tile b needs from beyond its end, kept in row order: ovf_ptr[ntiles + 1] (device)
int build_tile_overflow(const int *d_pos, const int *d_ovf_ptr, int total, int ntiles, int tile, int nnz,
                        const int *d_start_pos, int num_diag, const double *d_val, double *d_ovf_val, int *d_ovf_k, int positions,
                        hipStream_t st)
{
    if (total <= 0)
        return SMVP_OK;
    hipLaunchKernelGGL(overflow_entries, dim3(blocks_for(total)), dim3(256), 0, st, d_pos, d_ovf_ptr, ntiles, tile, nnz,
                       d_start_pos, num_diag, d_val, d_ovf_val, d_ovf_k, positions);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return SMVP_OK;
}

}  // namespace smvp

namespace smvp {

// Row-gather plan of a TJDS matrix (the one-kernel TJDS product, csr_stream_owner<., kFlavorTjds*>): the entries
// regrouped by row -- seg_ptr[rows + 1] bounds, pos[nnz] TJDS positions ascending inside a row -- and, if asked for,
// each stream entry's permuted column kcol.  val / row_ind / start_pos / perm themselves are only read.
int build_row_gather_plan(const int *d_row_ind, const int *d_start_pos, int num_diag, int nnz, int rows,
                          int *d_seg_ptr, int *d_pos, int *d_kcol, hipStream_t st)
{
    if (int rc = build_row_inverse(d_row_ind, nnz, rows, d_seg_ptr, d_pos, st))
        return rc;
    if (nnz > 0 && d_kcol) {
        hipLaunchKernelGGL(positions_to_columns, dim3(blocks_for(nnz)), dim3(256), 0, st, d_pos, nnz, d_start_pos, num_diag, d_kcol);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(st));
    return SMVP_OK;
}

}  // namespace smvp

extern "C" int smvp_csr_from_coo_device(const smvp_coo_t *d_coo, int rows, int cols, int nnz,
                                        int *d_row_ptr, int *d_col_ind, double *d_val, void *stream)
{
    if (rows < 0 || cols < 0 || nnz < 0 || !d_row_ptr || (nnz > 0 && (!d_coo || !d_col_ind || !d_val)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_from_coo_device: bad argument");
    if (nnz > 0 && (rows == 0 || cols == 0))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_from_coo_device: entries in an empty matrix");
    if (int rc = check_device(d_coo, "smvp_csr_from_coo_device"))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    Scratch sc;
    unsigned *order = nullptr;
    int *row_start = nullptr;
    if (int rc = sort_entries(sc, d_coo, rows, std::max(cols, 1), nnz, true, st, &order, &row_start))
        return rc;
    HIP_TRY(hipMemcpyAsync(d_row_ptr, row_start, sizeof(int) * ((size_t)rows + 1), hipMemcpyDeviceToDevice, st));
    if (nnz > 0) {
        hipLaunchKernelGGL(gather_csr, dim3(blocks_for(nnz)), dim3(256), 0, st, d_coo, order, nnz, d_col_ind, d_val);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(st));
    return SMVP_OK;
}

extern "C" int smvp_tjds_from_coo_device(const smvp_coo_t *d_coo, int rows, int cols, int nnz,
                                         int *d_perm, int *d_start_pos, int start_pos_capacity,
                                         int *d_row_ind, double *d_val,
                                         int *num_diag, int *ref_num_tjdiag, int *last_diag_single, void *stream)
{
    if (rows < 0 || cols < 0 || nnz < 0 || !d_start_pos || !num_diag || (cols > 0 && !d_perm) ||
        (nnz > 0 && (!d_coo || !d_row_ind || !d_val)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_from_coo_device: bad argument");
    if (nnz > 0 && (rows == 0 || cols == 0))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_from_coo_device: entries in an empty matrix");
    if (int rc = check_device(d_coo, "smvp_tjds_from_coo_device"))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    Scratch sc;
    unsigned *order = nullptr;
    int *col_start = nullptr;  // cols + 1 entries; col_start[c+1] - col_start[c] = length of column c
    if (int rc = sort_entries(sc, d_coo, std::max(rows, 1), cols, nnz, false, st, &order, &col_start))
        return rc;

    // column lengths, then the permutation: one stable sort on (longest-first) length keys
    int *col_len, *where, *width;
    unsigned *lk0, *lk1, *lc0, *lc1;
    HIP_TRY(sc.get(&col_len, (size_t)cols + 1));
    HIP_TRY(sc.get(&where, (size_t)cols + 1));
    HIP_TRY(sc.get(&lk0, (size_t)cols));
    HIP_TRY(sc.get(&lk1, (size_t)cols));
    HIP_TRY(sc.get(&lc0, (size_t)cols));
    HIP_TRY(sc.get(&lc1, (size_t)cols));
    int longest = 0, len0 = 0;
    if (cols > 0) {
        hipLaunchKernelGGL(column_lengths, dim3(blocks_for(cols)), dim3(256), 0, st, col_start, cols, col_len);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(length_keys, dim3(blocks_for(cols)), dim3(256), 0, st, col_len, cols, lk0, lc0);
        HIP_TRY(hipGetLastError());
        size_t tmp_bytes = 0;
        HIP_TRY(smvp::prim::radix_sort_pairs(nullptr, tmp_bytes, lk0, lk1, lc0, lc1, (size_t)cols, 0u, 32u, st));
        char *tmp;
        HIP_TRY(sc.get(&tmp, tmp_bytes));
        HIP_TRY(smvp::prim::radix_sort_pairs(tmp, tmp_bytes, lk0, lk1, lc0, lc1, (size_t)cols, 0u, 32u, st));
        hipLaunchKernelGGL(invert_perm, dim3(blocks_for(cols)), dim3(256), 0, st, lc1, cols, d_perm, where);
        HIP_TRY(hipGetLastError());
        unsigned first_key = 0;
        HIP_TRY(hipMemcpyAsync(&first_key, lk1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&len0, col_len, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        longest = (int)(0x7fffffffu - first_key);
    }
    if (longest + 1 > start_pos_capacity)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_from_coo_device: %d diagonals need start_pos_capacity >= %d",
                          longest, longest + 1);

    // start_pos = exclusive scan of the diagonal widths; start_pos[longest] = nnz falls out of the scan
    HIP_TRY(sc.get(&width, (size_t)longest + 1));
    hipLaunchKernelGGL(diagonal_widths, dim3(blocks_for(longest + 1)), dim3(256), 0, st, lk1, cols, longest, width);
    HIP_TRY(hipGetLastError());
    {
        size_t tmp_bytes = 0;
        HIP_TRY(smvp::prim::exclusive_scan(nullptr, tmp_bytes, width, d_start_pos, 0, (size_t)longest + 1, st));
        char *tmp;
        HIP_TRY(sc.get(&tmp, tmp_bytes));
        HIP_TRY(smvp::prim::exclusive_scan(tmp, tmp_bytes, width, d_start_pos, 0, (size_t)longest + 1, st));
    }
    if (nnz > 0) {
        hipLaunchKernelGGL(scatter_tjds, dim3(blocks_for(nnz)), dim3(256), 0, st, d_coo, order, nnz, col_start, where,
                           d_start_pos, d_row_ind, d_val);
        HIP_TRY(hipGetLastError());
    }
    int last_width = 0;
    if (longest > 0)
        HIP_TRY(hipMemcpyAsync(&last_width, width + longest - 1, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *num_diag = longest;
    if (ref_num_tjdiag)
        *ref_num_tjdiag = cols > 0 ? len0 : 0;
    if (last_diag_single)
        *last_diag_single = (longest > 0 && last_width == 1) ? 1 : 0;
    return SMVP_OK;
}

namespace {

// row of every entry (binary search in row_ptr), packed with its column into the sort key (row strip, column)
__global__ __launch_bounds__(256) void sweep_keys(const int *__restrict__ row_ptr, const int *__restrict__ col_ind, int rows,
                                                  int nnz, int strip_rows, u64 *__restrict__ key, unsigned *__restrict__ idx,
                                                  unsigned short *__restrict__ local_row)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz)
        return;
    int lo = 0, hi = rows - 1;  // last row r with row_ptr[r] <= e (rows without entries share a start: take the last)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (row_ptr[mid] <= e)
            lo = mid;
        else
            hi = mid - 1;
    }
    key[e] = ((u64)(unsigned)(lo / strip_rows) << 32) | (unsigned)col_ind[e];
    idx[e] = (unsigned)e;
    local_row[e] = (unsigned short)(lo % strip_rows);
}

__global__ __launch_bounds__(256) void sweep_gather(const u64 *__restrict__ key, const unsigned *__restrict__ idx,
                                                    const double *__restrict__ val, const unsigned short *__restrict__ local_row,
                                                    int nnz, int *__restrict__ e_col, double *__restrict__ e_val,
                                                    unsigned short *__restrict__ row_sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz)
        return;
    const unsigned e = idx[i];
    e_col[i] = (int)(unsigned)(key[i] & 0xffffffffu);
    e_val[i] = val[e];
    row_sorted[i] = local_row[e];
}

// The product kernel takes a strip's stream `chunk` entries at a time (one wavefront, chunk / 64 entries per lane).
// turn = how many earlier entries of the same chunk belong to the same row: entries with equal turns never share a
// row, so the wavefront adds turn 0, then turn 1, ... and every row is summed in stream order -- ascending column,
// the order of main-cli.c:410-416 -- whatever else is in flight.  Turns from turn_cap on are stored as turn_cap.
__global__ __launch_bounds__(256) void sweep_turns(const u64 *__restrict__ key, const long long *__restrict__ strip_ptr,
                                                   const unsigned short *__restrict__ row_sorted, int nnz, int chunk,
                                                   int row_bits, int turn_cap, int parts, unsigned part_width,
                                                   unsigned short *__restrict__ e_row)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz)
        return;
    // the stream this entry belongs to: its strip's, or (parts > 1) that of its strip's column part
    const long long a = strip_ptr[(unsigned)(key[i] >> 32) * (unsigned)parts + (unsigned)(key[i] & 0xffffffffu) / part_width];
    const long long first = a + ((long long)i - a) / chunk * chunk;
    const unsigned short mine = row_sorted[i];
    int turn = 0;
    for (long long q = first; q < i; ++q)
        turn += row_sorted[q] == mine ? 1 : 0;
    e_row[i] = (unsigned short)(mine | ((turn < turn_cap ? turn : turn_cap) << row_bits));
}

__global__ __launch_bounds__(256) void sweep_strip_bounds(const int *__restrict__ row_ptr, int rows, int strip_rows, int nstrips,
                                                          long long *__restrict__ strip_ptr)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b > nstrips)
        return;
    const long long r = (long long)b * strip_rows;
    strip_ptr[b] = row_ptr[r < rows ? r : rows];
}

// parts > 1: stream v = strip * parts + part holds the strip's entries whose column lies in [part * part_width, (part + 1) * part_width);
// the sorted keys of a strip are one run, so a stream starts at the first key of the run whose column reaches its part
__global__ __launch_bounds__(256) void sweep_part_bounds(const u64 *__restrict__ key, const int *__restrict__ row_ptr, int rows, int nnz,
                                                         int strip_rows, int nstrips, int parts, unsigned part_width,
                                                         long long *__restrict__ stream_ptr)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v > nstrips * parts)
        return;
    if (v == nstrips * parts) {
        stream_ptr[v] = nnz;
        return;
    }
    const int strip = v / parts, part = v % parts;
    const long long r0 = (long long)strip * strip_rows, r1 = r0 + strip_rows;
    long long lo = row_ptr[r0 < rows ? r0 : rows], hi = row_ptr[r1 < rows ? r1 : rows];
    const unsigned first_col = (unsigned)part * part_width;
    while (lo < hi) {  // first entry of the strip whose column is >= first_col
        const long long mid = (lo + hi) >> 1;
        if ((unsigned)(key[mid] & 0xffffffffu) < first_col)
            lo = mid + 1;
        else
            hi = mid;
    }
    stream_ptr[v] = lo;
}

}  // namespace

namespace smvp {

// Plan of the column-swept CSR kernel (csr_colsweep): the entries a second time, every strip of strip_rows rows (one
// wavefront's share) sorted by column (ties in input order), each with its row's number inside the strip and its turn
// inside its chunk of `chunk` stream entries (see sweep_turns), packed as row | turn << row_bits;
// strip_ptr[nstrips + 1] = where each strip's stream starts.  parts > 1 (column parts, round 6): every strip's stream is cut at the
// columns part * ceil(cols / parts) into `parts` streams of their own (turns counted per stream) and d_strip_ptr has
// nstrips * parts + 1 entries -- the kernel then gives each of a strip's parts to a wavefront of its own.
int build_colsweep_plan(const int *d_row_ptr, const int *d_col_ind, const double *d_val, int rows, int cols, int nnz, int strip_rows,
                        int parts, int chunk, int row_bits, int turn_cap, long long *d_strip_ptr, int *d_e_col, double *d_e_val,
                        unsigned short *d_e_row, hipStream_t st)
{
    if (strip_rows < 1 || strip_rows > (1 << row_bits) || row_bits < 1 || row_bits > 15 || chunk < 64 ||
        turn_cap != (1 << (16 - row_bits)) - 1 || (parts != 1 && parts != 2 && parts != 4 && parts != 8))
        return smvp::fail(SMVP_ERR_INVALID, "build_colsweep_plan: bad strip shape");
    const unsigned part_width = (unsigned)std::max(1, (int)(((long long)std::max(cols, 1) + parts - 1) / parts));
    const int nstrips = (rows + strip_rows - 1) / strip_rows;
    hipLaunchKernelGGL(sweep_strip_bounds, dim3(blocks_for((long long)nstrips + 1)), dim3(256), 0, st, d_row_ptr, rows,
                       strip_rows, nstrips, d_strip_ptr);
    HIP_TRY(hipGetLastError());
    if (nnz > 0) {
        Scratch sc;
        u64 *k0, *k1;
        unsigned *i0, *i1;
        unsigned short *lr, *lr_sorted;
        HIP_TRY(sc.get(&k0, (size_t)nnz));
        HIP_TRY(sc.get(&k1, (size_t)nnz));
        HIP_TRY(sc.get(&i0, (size_t)nnz));
        HIP_TRY(sc.get(&i1, (size_t)nnz));
        HIP_TRY(sc.get(&lr, (size_t)nnz));
        HIP_TRY(sc.get(&lr_sorted, (size_t)nnz));
        hipLaunchKernelGGL(sweep_keys, dim3(blocks_for(nnz)), dim3(256), 0, st, d_row_ptr, d_col_ind, rows, nnz, strip_rows, k0, i0, lr);
        HIP_TRY(hipGetLastError());
        const unsigned bits = 32u + (unsigned)bits_for(nstrips + 1);
        size_t tmp_bytes = 0;
        HIP_TRY(smvp::prim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, i0, i1, (size_t)nnz, 0u, bits, st));
        char *tmp;
        HIP_TRY(sc.get(&tmp, tmp_bytes));
        HIP_TRY(smvp::prim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, i0, i1, (size_t)nnz, 0u, bits, st));
        hipLaunchKernelGGL(sweep_gather, dim3(blocks_for(nnz)), dim3(256), 0, st, k1, i1, d_val, lr, nnz, d_e_col, d_e_val, lr_sorted);
        HIP_TRY(hipGetLastError());
        if (parts > 1) {
            hipLaunchKernelGGL(sweep_part_bounds, dim3(blocks_for((long long)nstrips * parts + 1)), dim3(256), 0, st, k1, d_row_ptr, rows, nnz,
                               strip_rows, nstrips, parts, part_width, d_strip_ptr);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(sweep_turns, dim3(blocks_for(nnz)), dim3(256), 0, st, k1, d_strip_ptr, lr_sorted, nnz, chunk, row_bits,
                           turn_cap, parts, part_width, d_e_row);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
    } else if (parts > 1) {
        HIP_TRY(hipMemsetAsync(d_strip_ptr, 0, sizeof(long long) * ((size_t)nstrips * parts + 1), st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return SMVP_OK;
}

}  // namespace smvp

namespace {

// smallest column of every tile of `tile` consecutive entries whose columns span (largest - smallest) less than 65536,
// -1 for the wider tiles; *narrow = how many tiles are of the first kind
__global__ __launch_bounds__(256) void tile_column_spans(const int *__restrict__ col_ind, int nnz, int tile,
                                                         int *__restrict__ col_base, int *__restrict__ narrow)
{
    __shared__ int lo_s[4], hi_s[4];
    const long long s = (long long)blockIdx.x * tile;
    int lo = 0x7fffffff, hi = -1;
    for (int i = threadIdx.x; i < tile && s + i < nnz; i += 256) {
        const int c = col_ind[s + i];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int ol = __shfl_down(lo, off, 64), oh = __shfl_down(hi, off, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        lo_s[threadIdx.x >> 6] = lo;
        hi_s[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            lo = lo_s[w] < lo ? lo_s[w] : lo;
            hi = hi_s[w] > hi ? hi_s[w] : hi;
        }
        const bool fits = hi < 0 || hi - lo < 65536;
        col_base[blockIdx.x] = !fits ? -1 : (hi < 0 ? 0 : lo);
        if (fits)
            atomicAdd(narrow, 1);
    }
}

__global__ __launch_bounds__(256) void column_offsets(const int *__restrict__ col_ind, int nnz, int tile,
                                                      const int *__restrict__ col_base, unsigned short *__restrict__ col16)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < nnz) {
        const int base = col_base[j / tile];
        col16[j] = base < 0 ? (unsigned short)0 : (unsigned short)(col_ind[j] - base);  // a wide tile reads col_ind itself
    }
}

}  // namespace

namespace smvp {

// kFlavorCsr16: col_ind a second time as 16-bit offsets from every tile's smallest column, for the tiles whose columns span
// less than 65536 (d_col_base[tile] = -1 marks the others: they read col_ind itself).  *fits when at least half of the
// tiles are of the narrow kind (nothing is written to d_col16 otherwise).  d_col_base has one entry per tile.
int build_column_offsets(const int *d_col_ind, int nnz, int tile, int *d_col_base, unsigned short *d_col16, int *fits,
                         hipStream_t st)
{
    *fits = 0;
    if (nnz <= 0)
        return SMVP_OK;
    const int ntiles = (int)(((long long)nnz + tile - 1) / tile);
    Scratch sc;
    int *narrow;
    HIP_TRY(sc.get(&narrow, 1));
    HIP_TRY(hipMemsetAsync(narrow, 0, sizeof(int), st));
    hipLaunchKernelGGL(tile_column_spans, dim3((unsigned)ntiles), dim3(256), 0, st, d_col_ind, nnz, tile, d_col_base, narrow);
    HIP_TRY(hipGetLastError());
    int h_narrow = 0;
    HIP_TRY(hipMemcpyAsync(&h_narrow, narrow, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (2 * (long long)h_narrow < (long long)ntiles)
        return SMVP_OK;
    hipLaunchKernelGGL(column_offsets, dim3(blocks_for(nnz)), dim3(256), 0, st, d_col_ind, nnz, tile, d_col_base, d_col16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    *fits = 1;
    return SMVP_OK;
}

}  // namespace smvp
