// smvp_near_window.hip -- K6: the NEAR part of the binned CSR plan with a row block's window of x in LDS.
//
// The near part of SMVP_CSR_KERNEL_BINNED (entries within `band` <= 4096 of the diagonal; main-cli.c:410-416 restricted
// to them) used to run on the tile kernel.  There a third of its gathers -- the entries between 512 and 4096 off the
// diagonal -- miss the 32 KB L1 and are served by the L2, one request each: 0.315 ms for the 72.7 M near entries of the
// SURVEY 8(d) random model against 0.214 ms with the same entries pulled inside +-512 (tools/exp_near.py).  Here no
// gather leaves the CU:
//
//   * one workgroup (1024 threads) per block of 8192 rows; x[R0 - 4096, R0 + 8192 + 4096) is loaded into LDS once
//     (128 KB; every element of x is loaded by two workgroups), beside the block's row order (16 KB) and its long rows;
//   * at plan time the block's rows are sorted by length (longest first); 64 sorted rows form a SLICE, stored step by
//     step: step k holds entry k of each of the 64 rows (lane = row), 8-byte value + 16-bit word (column inside the
//     window | valid | last step of the slice).  A row longer than 16 entries is a slice of its own: lane = every 64th
//     entry, summed across the wavefront in a fixed order at its end;
//   * every wavefront owns a contiguous run of steps -- its short slices (slices wave, wave + 16, ... of the block),
//     then its long rows (long rows wave, wave + 16, ...) -- and walks it as ONE flat stream, two batches of 8 steps
//     in flight; every lane sums its own row left to right in a register (the order of main-cli.c:410-416 among the
//     row's near entries).  No products in LDS, no barrier between the window's and the end;
//   * the sums stay in registers until the block's stream has ended; then the window's LDS becomes the block's y, is
//     filled in row order and stored with coalesced 16-byte stores.  (Stored lane by lane in sorted order the 8-byte
//     writes scatter over the block's 64 KB of y: 0.317 instead of 0.246 ms, tools/near_window_bench.hip.)
//
// Every row of y is written (0 for a row without near entries), so pass B can add the far sums afterwards.  Rows that
// keep their far entries in the near part (more than kBinRowCap of them) do not fit the window: they are listed apart
// and overwritten afterwards by csr_near_outside_rows, a wavefront per row with gathers from memory.  A block with more
// than kNwLongCap long rows, or a band wider than 4096, makes the plan unsuitable: the near part then stays on the tile
// kernel.  No atomics: the same bits from run to run.
//
// The plan is built on the device (the sort and scans of smvp_prim.h: set-up work).
#include "smvp_common.h"
#include "smvp_prim.h"
#include "smvp_kernels.h"


#include <algorithm>
#include <atomic>
#include <cstdint>
#include <vector>

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return smvp::fail(SMVP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace smvp {

namespace {

constexpr int kRB = kNwRowBlock, kBand = kNwBand, kWin = kRB + 2 * kBand, kThreads = 1024, kWaves = kThreads / 64;
#ifndef SMVP_NW_U
#define SMVP_NW_U 8
#endif
constexpr int kSlices = kRB / 64, kPerWave = kSlices / kWaves, kU = SMVP_NW_U;  // steps per batch (4, 12, 16 measured: no better)
constexpr int kValid = 0x8000, kEnd = 0x4000, kColMask = 0x3fff;
constexpr size_t kLds = sizeof(double) * kWin + 2 * kRB + 2 * kNwLongCap + 8 * kNwLongCap;
static_assert(kWin <= kColMask + 1, "a window column fits 14 bits");
static_assert(kLds <= 160 * 1024, "LDS of one CU");

typedef double double2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(kThreads) void csr_near_window(const double *__restrict__ x, double *__restrict__ y, int rows, int cols,
                                                           long long row0, int nblocks, const int *__restrict__ wave_ptr, const int *__restrict__ wave_n1,
                                                           const int *__restrict__ wave_n2, const unsigned short *__restrict__ perm16,
                                                           const double *__restrict__ sval, const unsigned short *__restrict__ sword,
                                                           const int *__restrict__ blk_long_ptr,
                                                           const unsigned short *__restrict__ long_row16)
{
    extern __shared__ double lds[];
    double *xw = lds;
    unsigned short *perm = reinterpret_cast<unsigned short *>(lds + kWin);
    unsigned short *longs = perm + kRB;
    double *lsum = reinterpret_cast<double *>(longs + kNwLongCap);  // the long rows' sums until the window is free
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // Workgroups go to the XCDs round robin: XCD i (workgroups i, i + 8, ...) takes the i-th eighth of the row blocks, in
    // order, so that the half windows neighbouring blocks share meet in one L2 (-0.8 % on the product; alone: nothing)
    const int per_xcd = (nblocks + 7) >> 3;
    const int b = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (b >= nblocks)
        return;
    const long long R0 = (long long)b * kRB;  // the block's first LOCAL row; the diagonal it sits on is column row0 + R0
    const long long wbase = row0 + R0 - kBand > 0 ? row0 + R0 - kBand : 0;
    const long long wend = row0 + R0 + kRB + kBand < (long long)cols ? row0 + R0 + kRB + kBand : (long long)cols;
    const int wlen = wend > wbase ? (int)(wend - wbase) : 0;

    const int gw = b * kWaves + wave;
    const long long off = wave_ptr[gw];
    const int n1 = wave_n1[gw], n = n1 + wave_n2[gw];
    const int lq0 = blk_long_ptr[b], nlong = blk_long_ptr[b + 1] - lq0;  // <= kNwLongCap (plan)
    const double *pv = sval + off * 64 + lane;
    const unsigned short *pw = sword + off * 64 + lane;

    // the run's first batch goes out before anything else; two batches are in flight from then on
    double v[2][kU];
    int c[2][kU];
    auto request = [&](int j0, int buf) {
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (j0 + u < n) {
                v[buf][u] = __builtin_nontemporal_load(pv + (size_t)(j0 + u) * 64);
                c[buf][u] = __builtin_nontemporal_load(pw + (size_t)(j0 + u) * 64);
            }
    };
    request(0, 0);
    // the window of x, the block's row order, its long rows
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        for (int i = 2 * t; i + 1 < wlen; i += 2 * kThreads)
            *reinterpret_cast<double2 *>(&xw[i]) = *reinterpret_cast<const double2 *>(x + wbase + i);
        if ((wlen & 1) && t == 0)
            xw[wlen - 1] = x[wbase + wlen - 1];
    } else {
        for (int i = t; i < wlen; i += kThreads)
            xw[i] = x[wbase + i];
    }
    for (int i = 4 * t; i < kRB; i += 4 * kThreads)
        *reinterpret_cast<uint2 *>(&perm[i]) = *reinterpret_cast<const uint2 *>(perm16 + (size_t)b * kRB + i);
    for (int i = t; i < nlong; i += kThreads)
        longs[i] = long_row16[lq0 + i];
    __syncthreads();

    double acc = 0.0;
    double accs[kPerWave];  // this lane's finished rows, one per short slice of the wavefront
#pragma unroll
    for (int k = 0; k < kPerWave; ++k)
        accs[k] = 0.0;
    int ks = 0, kl = 0;  // short slices / long rows of this wavefront finished so far
    auto batch = [&](int j0, int buf) {
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (j0 + u < n) {
                const int word = c[buf][u];
                if (word & kValid)
                    acc += v[buf][u] * xw[word & kColMask];
                if (__builtin_amdgcn_readfirstlane(word) & kEnd) {  // every lane's word of a slice's last step carries the flag
                    if (j0 + u < n1) {
#pragma unroll
                        for (int k = 0; k < kPerWave; ++k)
                            if (ks == k)
                                accs[k] = acc;
                        ++ks;
                    } else {
                        const double s = wave_sum_dpp(acc);
                        if (lane == 0)
                            lsum[wave + kWaves * kl] = s;
                        ++kl;
                    }
                    acc = 0.0;
                }
            }
    };
    for (int j = 0; j < n; j += 2 * kU) {
        request(j + kU, 1);
        batch(j, 0);
        request(j + 2 * kU, 0);
        batch(j + kU, 1);
    }
    // everybody has finished with the window: it becomes the block's y, filled in row order and stored coalesced
    __syncthreads();
    double *yb = xw;
#pragma unroll
    for (int k = 0; k < kPerWave; ++k) {
        const int r = perm[(wave + kWaves * k) * 64 + lane];
        if (r != 0xffff)
            yb[r] = accs[k];  // (a slice without entries never set its slot: 0)
    }
    for (int q = t; q < nlong; q += kThreads)
        yb[longs[q]] = lsum[q];
    __syncthreads();
    const int nrow = (long long)rows - R0 < kRB ? (int)(rows - R0) : kRB;
    if ((reinterpret_cast<uintptr_t>(y) & 15) == 0) {
        for (int i = 2 * t; i + 1 < nrow; i += 2 * kThreads)
            __builtin_nontemporal_store(*reinterpret_cast<const double2v *>(&yb[i]), reinterpret_cast<double2v *>(y + R0 + i));
        if ((nrow & 1) && t == 0)
            y[R0 + nrow - 1] = yb[nrow - 1];
    } else {
        for (int i = t; i < nrow; i += kThreads)
            y[R0 + i] = yb[i];
    }
}

// rows that keep entries outside the window (see the head of the file): a wavefront per row, gathers from memory
__global__ __launch_bounds__(256) void csr_near_outside_rows(int n_out, const int *__restrict__ out_row, const int *__restrict__ out_ptr,
                                                            const int *__restrict__ out_col, const double *__restrict__ out_val,
                                                            const double *__restrict__ x, double *__restrict__ y)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= n_out)
        return;
    double acc = 0.0;
    for (int j = out_ptr[q] + lane; j < out_ptr[q + 1]; j += 64)
        acc += out_val[j] * x[out_col[j]];
    acc = wave_sum_dpp(acc);
    if (lane == 0)
        y[out_row[q]] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// plan construction (device)
// ---------------------------------------------------------------------------------------------------------------------
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

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

// low 5 bits of a row's sort key: 16 - length for a short row (longest first), 17 for every other row
constexpr unsigned kLowOther = 17;

// per row: the sort key (block, low), long / outside flags, steps of a long row, entries of an outside row
__global__ __launch_bounds__(256) void nw_classify(const int *__restrict__ near_ptr, const int *__restrict__ capped, int rows,
                                                   unsigned *__restrict__ key, unsigned *__restrict__ row_id, int *__restrict__ is_long,
                                                   int *__restrict__ is_out, int *__restrict__ out_len)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r > rows)
        return;
    if (r == rows) {  // so that the scans' last elements are the totals
        is_long[r] = is_out[r] = out_len[r] = 0;
        return;
    }
    const int len = near_ptr[r + 1] - near_ptr[r];
    const bool out = capped[r] != 0, lng = !out && len > kNwShortCap;
    key[r] = (unsigned)(r / kRB) * 32u + (out || lng ? kLowOther : (unsigned)(kNwShortCap - len));
    row_id[r] = (unsigned)r;
    is_long[r] = lng ? 1 : 0;
    is_out[r] = out ? 1 : 0;
    out_len[r] = out ? len : 0;
}

__global__ __launch_bounds__(256) void nw_lists(const int *__restrict__ near_ptr, const int *__restrict__ is_long,
                                                const int *__restrict__ lscan, const int *__restrict__ is_out,
                                                const int *__restrict__ oscan, const int *__restrict__ out_scan, int rows, int nblocks,
                                                int *__restrict__ long_row, unsigned short *__restrict__ long_row16,
                                                int *__restrict__ blk_long_ptr, int *__restrict__ out_row, int *__restrict__ out_ptr)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r <= nblocks) {
        const long long first = (long long)r * kRB;
        blk_long_ptr[r] = lscan[first < rows ? first : rows];
    }
    if (r > rows)
        return;
    if (r == rows) {
        out_ptr[oscan[rows]] = out_scan[rows];
        return;
    }
    if (is_long[r]) {
        long_row[lscan[r]] = r;
        long_row16[lscan[r]] = (unsigned short)(r % kRB);
    }
    if (is_out[r]) {
        out_row[oscan[r]] = r;
        out_ptr[oscan[r]] = out_scan[r];
    }
}

__device__ __forceinline__ int slice_width(const unsigned *__restrict__ skey, long long s, int rows)
{
    if (s * 64 >= rows)
        return 0;
    const unsigned low = skey[s * 64] & 31u;  // sorted: the slice's first row is its longest
    return low <= (unsigned)kNwShortCap ? kNwShortCap - (int)low : 0;
}

// steps of every wavefront's run: its short slices, its long rows
__global__ __launch_bounds__(256) void nw_wave_steps(const unsigned *__restrict__ skey, const int *__restrict__ near_ptr,
                                                     const int *__restrict__ long_row, const int *__restrict__ blk_long_ptr, int rows,
                                                     int nwaves, int *__restrict__ wave_n1, int *__restrict__ wave_n2,
                                                     int *__restrict__ steps)
{
    const int gw = blockIdx.x * 256 + threadIdx.x;
    if (gw > nwaves)
        return;
    if (gw == nwaves) {
        steps[gw] = 0;
        return;
    }
    const int b = gw / kWaves, w = gw % kWaves;
    int n1 = 0, n2 = 0;
    for (int i = 0; i < kPerWave; ++i)
        n1 += slice_width(skey, (long long)b * kSlices + w + kWaves * i, rows);
    for (int q = blk_long_ptr[b] + w; q < blk_long_ptr[b + 1]; q += kWaves) {
        const int r = long_row[q];
        n2 += (near_ptr[r + 1] - near_ptr[r] + 63) / 64;
    }
    wave_n1[gw] = n1, wave_n2[gw] = n2, steps[gw] = n1 + n2;
}

// the short rows' entries into their slices; the block's row order
__global__ __launch_bounds__(256) void nw_emit_short(const unsigned *__restrict__ skey, const unsigned *__restrict__ order,
                                                     const int *__restrict__ near_ptr, const int *__restrict__ near_col,
                                                     const double *__restrict__ near_val, const int *__restrict__ wave_ptr, int rows,
                                                     long long row0, unsigned short *__restrict__ perm16, double *__restrict__ sval,
                                                     unsigned short *__restrict__ sword)
{
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= rows)
        return;
    const int r = (int)order[p];
    const unsigned low = skey[p] & 31u;
    const bool is_short = low <= (unsigned)kNwShortCap;
    const long long s = p / 64;
    const int lane = (int)(p % 64), b = (int)(p / kRB), k = (int)(s % kSlices), w = k % kWaves, i = k / kWaves;
    perm16[p] = is_short ? (unsigned short)(r - b * kRB) : (unsigned short)0xffff;
    const int width = slice_width(skey, s, rows);
    if (width == 0)
        return;
    long long off = wave_ptr[b * kWaves + w];
    for (int ii = 0; ii < i; ++ii)
        off += slice_width(skey, (long long)b * kSlices + w + kWaves * ii, rows);
    const int a = near_ptr[r], len = is_short ? near_ptr[r + 1] - a : 0;
    const long long wbase = row0 + (long long)b * kRB - kBand > 0 ? row0 + (long long)b * kRB - kBand : 0;
    for (int e = 0; e < len; ++e) {
        const long long d = (off + e) * 64 + lane;
        sval[d] = near_val[a + e];
        sword[d] = (unsigned short)((int)(near_col[a + e] - wbase) | kValid | (e == width - 1 ? kEnd : 0));
    }
    if (len < width)
        sword[(off + width - 1) * 64 + lane] = (unsigned short)kEnd;
}

// a long row's entries into its slice: one wavefront per long row
__global__ __launch_bounds__(256) void nw_emit_long(const int *__restrict__ long_row, const int *__restrict__ blk_long_ptr,
                                                    const int *__restrict__ near_ptr, const int *__restrict__ near_col,
                                                    const double *__restrict__ near_val, const int *__restrict__ wave_ptr,
                                                    const int *__restrict__ wave_n1, int nlong, long long row0, double *__restrict__ sval,
                                                    unsigned short *__restrict__ sword)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nlong)
        return;
    const int r = long_row[q], b = r / kRB, qi = q - blk_long_ptr[b], w = qi % kWaves, m = qi / kWaves;
    long long off = (long long)wave_ptr[b * kWaves + w] + wave_n1[b * kWaves + w];
    for (int mm = 0; mm < m; ++mm) {
        const int rr = long_row[blk_long_ptr[b] + w + kWaves * mm];
        off += (near_ptr[rr + 1] - near_ptr[rr] + 63) / 64;
    }
    const int a = near_ptr[r], len = near_ptr[r + 1] - a, steps = (len + 63) / 64;
    const long long wbase = row0 + (long long)b * kRB - kBand > 0 ? row0 + (long long)b * kRB - kBand : 0;
    for (int j = lane; j < steps * 64; j += 64) {
        const long long d = (off + j / 64) * 64 + lane;
        int word = j / 64 == steps - 1 ? kEnd : 0;
        if (j < len) {
            sval[d] = near_val[a + j];
            word |= (int)(near_col[a + j] - wbase) | kValid;
        }
        sword[d] = (unsigned short)word;
    }
}

// the outside rows' entries, as they are
__global__ __launch_bounds__(256) void nw_emit_outside(const int *__restrict__ out_row, const int *__restrict__ out_ptr,
                                                       const int *__restrict__ near_ptr, const int *__restrict__ near_col,
                                                       const double *__restrict__ near_val, int n_out, int *__restrict__ out_col,
                                                       double *__restrict__ out_val)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= n_out)
        return;
    const int r = out_row[q], a = near_ptr[r], len = near_ptr[r + 1] - a, o = out_ptr[q];
    for (int j = lane; j < len; j += 64) {
        out_col[o + j] = near_col[a + j];
        out_val[o + j] = near_val[a + j];
    }
}

int scan_exclusive(const int *in, int *out, size_t n, Scratch &sc, hipStream_t st)
{
    size_t bytes = 0;
    HIP_TRY(smvp::prim::exclusive_scan(nullptr, bytes, in, out, 0, n, st));
    char *tmp;
    HIP_TRY(sc.get(&tmp, bytes));
    HIP_TRY(smvp::prim::exclusive_scan(tmp, bytes, in, out, 0, n, st));
    return SMVP_OK;
}

template <class T>
int own(T **out, size_t count, size_t *bytes)
{
    if (hipMalloc((void **)out, std::max<size_t>(count, 4) * sizeof(T)) != hipSuccess) {
        *out = nullptr;
        return smvp::fail(SMVP_ERR_ALLOC, "cannot allocate the near-window plan (%zu bytes)", count * sizeof(T));
    }
    *bytes += count * sizeof(T);
    return SMVP_OK;
}

}  // namespace

void free_near_window(NearWindow *p)
{
    if (!p)
        return;
    for (void *q : {(void *)p->wave_ptr, (void *)p->wave_n1, (void *)p->wave_n2, (void *)p->perm16, (void *)p->sval, (void *)p->sword,
                    (void *)p->blk_long_ptr, (void *)p->long_row16, (void *)p->out_row, (void *)p->out_ptr, (void *)p->out_col,
                    (void *)p->out_val})
        if (q)
            (void)hipFree(q);
    *p = NearWindow();
}

// `capped[r]` != 0: row r keeps entries outside the band (it goes to the outside list).  *out stays off (and SMVP_OK is
// returned) where the plan does not suit the matrix.
int build_near_window(const int *near_ptr, const int *near_col, const double *near_val, const int *capped, int rows, int cols,
                      int nnz_near, int band, long long row0, NearWindow *out, hipStream_t st)
{
    free_near_window(out);
    if (rows <= 0 || nnz_near <= 0 || band > kBand)
        return SMVP_OK;
    NearWindow P;
    P.rows = rows, P.cols = cols, P.row0 = row0;
    P.nblocks = (int)(((long long)rows + kRB - 1) / kRB);
    const int nwaves = P.nblocks * kWaves;
    const size_t padded_rows = (size_t)P.nblocks * kRB;
    Scratch sc;
    struct Guard {  // whatever the plan owns so far goes with an early return
        NearWindow *p;
        ~Guard()
        {
            if (p)
                free_near_window(p);
        }
    } guard{&P};
    auto give_up = [&](int rc) { return rc; };
    // ---- classes, the sort inside every block, the two lists
    unsigned *k0, *k1, *i0, *i1;
    int *is_long, *lscan, *is_out, *oscan, *out_len, *out_scan, *long_row, *steps;
    HIP_TRY(sc.get(&k0, (size_t)rows));
    HIP_TRY(sc.get(&k1, (size_t)rows));
    HIP_TRY(sc.get(&i0, (size_t)rows));
    HIP_TRY(sc.get(&i1, (size_t)rows));
    HIP_TRY(sc.get(&is_long, (size_t)rows + 1));
    HIP_TRY(sc.get(&lscan, (size_t)rows + 1));
    HIP_TRY(sc.get(&is_out, (size_t)rows + 1));
    HIP_TRY(sc.get(&oscan, (size_t)rows + 1));
    HIP_TRY(sc.get(&out_len, (size_t)rows + 1));
    HIP_TRY(sc.get(&out_scan, (size_t)rows + 1));
    hipLaunchKernelGGL(nw_classify, dim3(blocks_for((long long)rows + 1)), dim3(256), 0, st, near_ptr, capped, rows, k0, i0, is_long,
                       is_out, out_len);
    HIP_TRY(hipGetLastError());
    {
        unsigned bits = 6;
        while (bits < 32 && (1ull << bits) < (unsigned long long)P.nblocks * 32ull)
            ++bits;
        size_t bytes = 0;
        HIP_TRY(smvp::prim::radix_sort_pairs(nullptr, bytes, k0, k1, i0, i1, (size_t)rows, 0u, bits, st));
        char *tmp;
        HIP_TRY(sc.get(&tmp, bytes));
        HIP_TRY(smvp::prim::radix_sort_pairs(tmp, bytes, k0, k1, i0, i1, (size_t)rows, 0u, bits, st));
    }
    if (int rc = scan_exclusive(is_long, lscan, (size_t)rows + 1, sc, st))
        return rc;
    if (int rc = scan_exclusive(is_out, oscan, (size_t)rows + 1, sc, st))
        return rc;
    if (int rc = scan_exclusive(out_len, out_scan, (size_t)rows + 1, sc, st))
        return rc;
    int nlong = 0, n_out = 0, out_nnz = 0;
    HIP_TRY(hipMemcpyAsync(&nlong, lscan + rows, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&n_out, oscan + rows, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&out_nnz, out_scan + rows, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    P.n_out = n_out;
    HIP_TRY(sc.get(&long_row, (size_t)nlong + 1));
    if (int rc = own(&P.long_row16, (size_t)nlong + 1, &P.plan_bytes))
        return give_up(rc);
    if (int rc = own(&P.blk_long_ptr, (size_t)P.nblocks + 1, &P.plan_bytes))
        return give_up(rc);
    if (int rc = own(&P.out_row, (size_t)n_out + 1, &P.plan_bytes))
        return give_up(rc);
    if (int rc = own(&P.out_ptr, (size_t)n_out + 1, &P.plan_bytes))
        return give_up(rc);
    if (int rc = own(&P.out_col, (size_t)out_nnz, &P.plan_bytes))
        return give_up(rc);
    if (int rc = own(&P.out_val, (size_t)out_nnz, &P.plan_bytes))
        return give_up(rc);
    hipLaunchKernelGGL(nw_lists, dim3(blocks_for(std::max<long long>((long long)rows + 1, (long long)P.nblocks + 1))), dim3(256), 0, st,
                       near_ptr, is_long, lscan, is_out, oscan, out_scan, rows, P.nblocks, long_row, P.long_row16, P.blk_long_ptr,
                       P.out_row, P.out_ptr);
    HIP_TRY(hipGetLastError());
    {  // the long rows of a block must fit the kernel's LDS list
        std::vector<int> blp((size_t)P.nblocks + 1);
        HIP_TRY(hipMemcpyAsync(blp.data(), P.blk_long_ptr, blp.size() * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int b = 0; b < P.nblocks; ++b)
            if (blp[(size_t)b + 1] - blp[(size_t)b] > kNwLongCap)
                return give_up(SMVP_OK);
    }
    // ---- every wavefront's run of steps
    if (int rc = own(&P.wave_ptr, (size_t)nwaves + 1, &P.plan_bytes))
        return give_up(rc);
    if (int rc = own(&P.wave_n1, (size_t)nwaves + 1, &P.plan_bytes))
        return give_up(rc);
    if (int rc = own(&P.wave_n2, (size_t)nwaves + 1, &P.plan_bytes))
        return give_up(rc);
    HIP_TRY(sc.get(&steps, (size_t)nwaves + 1));
    hipLaunchKernelGGL(nw_wave_steps, dim3(blocks_for((long long)nwaves + 1)), dim3(256), 0, st, k1, near_ptr, long_row, P.blk_long_ptr,
                       rows, nwaves, P.wave_n1, P.wave_n2, steps);
    HIP_TRY(hipGetLastError());
    if (int rc = scan_exclusive(steps, P.wave_ptr, (size_t)nwaves + 1, sc, st))
        return give_up(rc);
    int total_steps = 0;
    HIP_TRY(hipMemcpyAsync(&total_steps, P.wave_ptr + nwaves, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    P.slots = (long long)total_steps * 64;
    // ---- the streams
    if (int rc = own(&P.sval, (size_t)P.slots, &P.plan_bytes))
        return give_up(rc);
    if (int rc = own(&P.sword, (size_t)P.slots, &P.plan_bytes))
        return give_up(rc);
    if (int rc = own(&P.perm16, padded_rows, &P.plan_bytes))
        return give_up(rc);
    HIP_TRY(hipMemsetAsync(P.sval, 0, std::max<size_t>((size_t)P.slots, 4) * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(P.sword, 0, std::max<size_t>((size_t)P.slots, 4) * sizeof(unsigned short), st));
    HIP_TRY(hipMemsetAsync(P.perm16, 0xff, std::max<size_t>(padded_rows, 4) * sizeof(unsigned short), st));
    hipLaunchKernelGGL(nw_emit_short, dim3(blocks_for(rows)), dim3(256), 0, st, k1, i1, near_ptr, near_col, near_val, P.wave_ptr, rows,
                       row0, P.perm16, P.sval, P.sword);
    if (nlong > 0)
        hipLaunchKernelGGL(nw_emit_long, dim3((unsigned)((nlong + 3) / 4)), dim3(256), 0, st, long_row, P.blk_long_ptr, near_ptr, near_col,
                           near_val, P.wave_ptr, P.wave_n1, nlong, row0, P.sval, P.sword);
    if (n_out > 0)
        hipLaunchKernelGGL(nw_emit_outside, dim3((unsigned)((n_out + 3) / 4)), dim3(256), 0, st, P.out_row, P.out_ptr, near_ptr, near_col,
                           near_val, n_out, P.out_col, P.out_val);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    P.on = true;
    *out = P;
    guard.p = nullptr;
    return SMVP_OK;
}

hipError_t near_window_reserve_lds()
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    static std::atomic<unsigned long long> asked{0};  // more than 64 KB of dynamic LDS must be asked for, once per device
    if (dev >= 64 || !(asked.load() >> dev & 1ull)) {
        e = hipFuncSetAttribute((const void *)csr_near_window, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
        if (e != hipSuccess)
            return e;
        if (dev < 64)
            asked.fetch_or(1ull << dev);
    }
    return hipSuccess;
}

hipError_t launch_near_window(const NearWindow &p, const double *x, double *y, hipStream_t stream)
{
    hipError_t e = near_window_reserve_lds();
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(csr_near_window, dim3((unsigned)((p.nblocks + 7) / 8 * 8)), dim3(kThreads), kLds, stream, x, y, p.rows, p.cols,
                       p.row0, p.nblocks, p.wave_ptr,
                       p.wave_n1, p.wave_n2, p.perm16, p.sval, p.sword, p.blk_long_ptr, p.long_row16);
    if (p.n_out > 0)
        hipLaunchKernelGGL(csr_near_outside_rows, dim3((unsigned)((p.n_out + 3) / 4)), dim3(256), 0, stream, p.n_out, p.out_row, p.out_ptr,
                           p.out_col, p.out_val, x, y);
    return hipGetLastError();
}

}  // namespace smvp
