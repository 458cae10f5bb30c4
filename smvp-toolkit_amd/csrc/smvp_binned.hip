// smvp_binned.hip -- K5: CSR product with the FAR gathers binned through LDS (SMVP_CSR_KERNEL_BINNED).
//
// Replaces main-cli.c:410-416 for matrices like SURVEY 8(d)'s memplus-shaped random model: most entries lie in a band
// around the diagonal, but a large share (39 % there) points anywhere in an operand far larger than the L2.  On the tile
// kernel every such gather pulls its own 128-byte line of x from the Infinity Cache (46.8 M entries x 128 B = 6 GB for
// 1.8 GB of matrix: the product runs at the chip's L2-miss gather rate, 21 % of HBM peak).  Here the entries are split
// at plan time by |column - row| > band:
//
//   near   summed out of a row block's window of x in LDS (K6, smvp_near_window.hip: built from the near CSR arrays, which are
//          then released) -- or, where that plan does not suit, its own CSR arrays on the tile kernel (csr_stream_owner).
//   far    two passes that never gather from memory:
//     pass A  (csr_binned_far_products) one workgroup per COLUMN BLOCK of 16384 columns: the block of x is loaded into
//             LDS once (128 KB), the block's far entries are streamed -- 8-byte value + 16-bit word (local column | first-
//             of-cell flag) -- and every product is stored into the BINS.  The bins are ordered (super block of rows,
//             column block, row, column): inside a cell = (column block, super block) the stream order and the bin order
//             agree, so an entry's product goes to (its stream position + the cell's shift), and what a wavefront stores
//             are runs of a cell's length.  Cells are what the speed hangs on -- every run ends in partly written 128-
//             byte lines, and a partly written line costs the fabric a whole one (measured, tools/far_binned_bench.hip:
//             16-entry cells 0.31-0.38 ms, 128-entry cells 0.21-0.24 ms for 46.8 M entries) -- hence the super blocks.
//     pass B  (csr_binned_far_sums) one workgroup per ROW BLOCK of at most 8192 far entries: the row block's products,
//             one sub-run per column block, are fetched as one virtual stream -- 16-bit word (LDS slot | first-of-sub-run
//             flag) read contiguously, product at (position + the sub-run's shift) -- into their row-major slot in LDS;
//             one lane per row then sums its slots left to right (ascending column: the order of main-cli.c:410-416
//             among the row's far entries) and adds the sum to y, which the near kernel has written before.  The row
//             blocks of one super block run together on one XCD, so the lines their sub-runs share are fetched once.
//
// Every x line is read once per product instead of once per far entry; per far entry the product moves 10 B (stream A)
// + 8 B (bin written) + 8 B + 2 B (bin and word read) = 28 B instead of 128 + 12.  No atomics anywhere: the result is the
// same from run to run, bit for bit.  A row's sum is (near part, left to right for a row of at most 16 near entries) + (far
// part, left to right).  Pass A needs x only: with K6 it runs beside the near part on a stream of its own (engine).
//
// The plan -- near arrays, both streams, the bins -- is built on the device (the sorts and scans of smvp_prim.h: set-up work).
#include "smvp_common.h"
#include "smvp_prim.h"
#include "smvp_kernels.h"


#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return smvp::fail(SMVP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace smvp {

namespace {

typedef unsigned long long u64;
constexpr int kBinPad = 0xffff;   // stream word of a padding entry
constexpr int kBinFlag = 0x8000;  // first entry of a cell
constexpr int kBinLongRow = 32;   // far rows longer than this are summed by a whole wavefront
constexpr int kBinPre = 3;        // far rows per lane whose bounds and y are requested ahead of the stream
constexpr int kBinLongCap = kBinSlots / kBinLongRow + 1;  // long far rows a row block can hold (at the largest block size)
constexpr size_t kLdsA = sizeof(double) * (1u << kBinColBits) + sizeof(int) * kBinShiftCap;
constexpr size_t lds_b_bytes(int slots) { return sizeof(double) * (slots + kBinLongCap) + sizeof(int) * (kBinShiftCap + 3 * kBinLongCap + 1); }

// pass B's product loads: a sub-run is a few entries of a line that the super block's other row blocks (on the same XCD)
// read the rest of, so they go through the caches like any load (non-temporal, i.e. past the L1: measured slower)
__device__ __forceinline__ double kBinLoadBins(const double *p) { return *p; }

// ---------------------------------------------------------------------------------------------------------------------
// The streams are stored in GROUPS of 256 entries, every group interleaved so that a lane's wide load holds the entries it
// works on -- logical positions l, l + 64, l + 128, l + 192 of the group for lane l: values as two 16-byte pairs
// {l, l + 64} and {l + 128, l + 192}, the 16-bit words as one 8-byte quad.  (8-byte-per-lane streaming loads reach little
// more than half the rate of 16-byte ones, MI355X_MICROARCH "Workgroup dispatch ..." table; the first build of these
// kernels, with one value and one word per load, ran pass A in 0.251 ms against 0.2xx now.)  Logical position p of the
// padded stream -- what shifts, chunk counts and group pointers speak of -- lives at
//     values: (p & ~255) + 128 * ((p >> 7) & 1) + 2 * (p & 63) + ((p >> 6) & 1)        words: (p & ~255) + 4 * (p & 63) + ((p >> 6) & 3)
// ---------------------------------------------------------------------------------------------------------------------
typedef double double2v __attribute__((ext_vector_type(2)));      // clang vectors: what the non-temporal builtins take
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ int stream_value_index(int p) { return (p & ~255) + 128 * ((p >> 7) & 1) + 2 * (p & 63) + ((p >> 6) & 1); }
__host__ __device__ __forceinline__ int stream_word_index(int p) { return (p & ~255) + 4 * (p & 63) + ((p >> 6) & 3); }

// ---------------------------------------------------------------------------------------------------------------------
// pass A.  G: groups of 256 entries a wavefront has in flight per iteration (4 * G entries per lane)
// ---------------------------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kBinThreads) void csr_binned_far_products(
    const double *__restrict__ x, int cols, const double *__restrict__ a_val, const unsigned short *__restrict__ a_word,
    const int *__restrict__ a_chunk, const int *__restrict__ a_ptr, const int *__restrict__ a_shift_ptr,
    const int *__restrict__ a_shift, double *__restrict__ bins, int splits, int items)
{
    extern __shared__ double lds[];
    double *xs = lds;
    int *shift = reinterpret_cast<int *>(lds + (1 << kBinColBits));
    const int t = threadIdx.x;
    // one work item = (column block, part of its stream); a workgroup takes items blockIdx.x, + gridDim.x, ...: the launch
    // gives every item its own workgroup (a grid of one persistent workgroup per CU was tried to share the CUs with the
    // near product: no gain, see the engine)
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    if (item != (int)blockIdx.x)
        __syncthreads();  // everybody has left the previous item's block of x
    const int cb = item / splits, part = item % splits;
    const int a = a_ptr[cb], z = a_ptr[cb + 1];  // multiples of 256
    // with few column blocks every block's stream is cut into `splits` runs of whole groups, one workgroup each
    const int groups = (z - a) >> 8, per = (groups + splits - 1) / splits;
    const int pa = a + 256 * (part * per < groups ? part * per : groups);
    const int pz = a + 256 * ((part + 1) * per < groups ? (part + 1) * per : groups);
    if (pa >= pz)
        continue;
    const long long c0 = (long long)cb << kBinColBits;
    if (c0 + (1 << kBinColBits) <= (long long)cols) {
        for (int i = 2 * t; i < (1 << kBinColBits); i += 2 * kBinThreads)
            *reinterpret_cast<double2 *>(xs + i) = *reinterpret_cast<const double2 *>(x + c0 + i);
    } else {
        for (int i = t; i < (1 << kBinColBits); i += kBinThreads)
            xs[i] = c0 + i < (long long)cols ? x[c0 + i] : 0.0;
    }
    const int sp = a_shift_ptr[cb], ncell = a_shift_ptr[cb + 1] - sp;
    for (int i = t; i < ncell && i < kBinShiftCap; i += kBinThreads)
        shift[i] = a_shift[sp + i];
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);  // (uniform: the chunk counts become scalar loads)
    const unsigned long long le = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1);
    // Software-pipelined: a wavefront's next groups are requested before its current ones are multiplied and stored, and
    // the first ones before the barrier behind which the block of x stands in LDS (they do not depend on it).
    constexpr int STEP = (kBinThreads / 64) * 256 * G;
    double2v v[2 * G], vn[2 * G];
    uint2v w[G], wn[G];
    int4 cc[G], cn[G];
    auto load = [&](int base, double2v (&vv)[2 * G], uint2v (&ww)[G], int4 (&cw)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int b = base + 256 * g;
            if (b < pz) {
                vv[2 * g] = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(a_val + b) + lane);
                vv[2 * g + 1] = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(a_val + b + 128) + lane);
                ww[g] = __builtin_nontemporal_load(reinterpret_cast<const uint2v *>(a_word + b) + lane);
                cw[g] = *reinterpret_cast<const int4 *>(a_chunk + (b >> 6));
            } else {
                vv[2 * g] = vv[2 * g + 1] = double2v{0.0, 0.0};
                ww[g] = uint2v{0xffffffffu, 0xffffffffu};
                cw[g] = make_int4(0, 0, 0, 0);
            }
        }
    };
    int base = pa + wave * 256 * G;
    load(base, v, w, cc);
    __syncthreads();
    for (; base < pz; base += STEP) {
        load(base + STEP, vn, wn, cn);  // (past the end: nothing is read)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int wk[4] = {(int)(w[g].x & 0xffffu), (int)(w[g].x >> 16), (int)(w[g].y & 0xffffu), (int)(w[g].y >> 16)};
            const double vk[4] = {v[2 * g].x, v[2 * g].y, v[2 * g + 1].x, v[2 * g + 1].y};
            const int ck[4] = {cc[g].x, cc[g].y, cc[g].z, cc[g].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = base + 256 * g + 64 * k + lane;  // logical position
                const bool real = wk[k] != kBinPad;
                const unsigned long long m = __ballot(real && (wk[k] & kBinFlag));
                const int cell = ck[k] + __popcll(m & le);
                if (real) {
                    const double p = vk[k] * xs[wk[k] & ((1 << kBinColBits) - 1)];
                    const int sh = cell < kBinShiftCap ? shift[cell] : a_shift[sp + cell];
                    __builtin_nontemporal_store(p, bins + (j + sh));
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            v[2 * g] = vn[2 * g], v[2 * g + 1] = vn[2 * g + 1];
            w[g] = wn[g];
            cc[g] = cn[g];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// pass B
// ---------------------------------------------------------------------------------------------------------------------
// Diagnostic builds only (make HIPFLAGS+=-DSMVP_PHASE_STAMPS): where a workgroup of pass B spends its time (a sample of
// the workgroups adds 100 MHz wall-clock ticks per phase to global counters; debug_binned_phases() prints them).
#ifdef SMVP_PHASE_STAMPS
__device__ unsigned long long g_binned_phase[8];
#define SMVP_STAMP(n)                                                                 \
    do {                                                                              \
        if (threadIdx.x == 0 && (blockIdx.x & 31) == 5) {                             \
            const unsigned long long now_ = wall_clock64();                           \
            if ((n) > 0)                                                              \
                atomicAdd(&g_binned_phase[(n) - 1], now_ - stamp_prev_);              \
            else                                                                      \
                atomicAdd(&g_binned_phase[7], 1ull);                                  \
            stamp_prev_ = now_;                                                       \
        }                                                                             \
    } while (0)
#else
#define SMVP_STAMP(n) do { } while (0)
#endif

template <int SLOTS, int THREADS, int G>
__global__ __launch_bounds__(THREADS) void csr_binned_far_sums(
    const double *__restrict__ bins, const unsigned short *__restrict__ b_word, const int *__restrict__ b_chunk,
    const int4 *__restrict__ b_desc, const int *__restrict__ b_shift, const int *__restrict__ fr_row,
    const int *__restrict__ fr_ptr, const unsigned *__restrict__ fr32, double *__restrict__ y, int nrb, int q)
{
    extern __shared__ double lds[];  // all of it dynamic: the products' slots come first, 8-byte aligned
#ifdef SMVP_PHASE_STAMPS
    unsigned long long stamp_prev_ = 0;
#endif
    SMVP_STAMP(0);
    double *fp = lds;
    double *long_y = lds + SLOTS;  // what y holds for the queued long rows (read when the row is queued, not when it is summed)
    int *shift = reinterpret_cast<int *>(long_y + kBinLongCap);
    int *long_row = shift + kBinShiftCap, *long_a = long_row + kBinLongCap, *long_z = long_a + kBinLongCap;
    int &long_count = long_z[kBinLongCap];
    int rb = blockIdx.x;
    {
        const int xcd = rb & 7, seq = rb >> 3;  // XCD i takes q consecutive row blocks -- one super block -- one after the other
        rb = (seq / q) * (8 * q) + xcd * q + seq % q;
    }
    if (rb >= nrb)
        return;
    const int t = threadIdx.x;
    // the block's descriptor, one scalar load (the first build chained b_ptr, blk_fr, fr_ptr[blk_fr]: two round trips)
    const int4 d0 = b_desc[2 * rb], d1 = b_desc[2 * rb + 1];
    const int a = d0.x, z = d0.y;  // its stretch of the stream: multiples of 256
    if (a >= z)
        return;  // (a row block without entries has no far rows either)
    const int k0 = d0.z, k1 = d0.w;    // its far rows
    const int f0 = d1.x;               // its first far entry: slots count from it
    const int sp = d1.y, nruns = d1.z; // its sub-runs' shifts
    const int row_base = d1.w;         // fr32: the block's first far row (the list holds 16-bit offsets from it)
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);  // (uniform: the chunk counts become scalar loads)
    const unsigned long long le = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1);
    // Everything that hangs on the scalars above goes out together, ahead of the one barrier the shifts need: the sub-runs'
    // shifts, this wavefront's first words and chunk counts, the bounds of this lane's first far rows.  (The first build
    // asked for the words only behind that barrier, and for y behind the far rows in front of it: 3.7 us to the barrier
    // and 7.2 us from there to the next, of 16 us per workgroup -- in-kernel stamps, -DSMVP_PHASE_STAMPS; it made no difference to the whole: the phases of the
    // two workgroups a CU holds do not cover each other.)
    constexpr int STEP = (THREADS / 64) * 256 * G;
    uint2v w[G];
    int4 cc[G];
    auto load_words = [&](int base) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int b = base + 256 * g;
            if (b < z) {
                w[g] = __builtin_nontemporal_load(reinterpret_cast<const uint2v *>(b_word + b) + lane);
                cc[g] = *reinterpret_cast<const int4 *>(b_chunk + (b >> 6));
            } else {
                w[g] = uint2v{0xffffffffu, 0xffffffffu};
                cc[g] = make_int4(0, 0, 0, 0);
            }
        }
    };
    int base = a + wave * 256 * G;
    load_words(base);
    int prow[kBinPre], pa[kBinPre], pz[kBinPre];
    double py[kBinPre];
#pragma unroll
    for (int u = 0; u < kBinPre; ++u) {
        const int k = k0 + u * THREADS + t;
        prow[u] = -1, pa[u] = 0, pz[u] = 0;
        if (k < k1) {
            if (fr32) {  // one word per far row: row - row_base | first slot << 16; the row ends where the next begins (a sentinel
                const unsigned w0 = fr32[k + rb], w1 = fr32[k + rb + 1];  // per block: hence the + rb)
                prow[u] = row_base + (int)(w0 & 0xffffu);
                pa[u] = (int)(w0 >> 16);
                pz[u] = (int)(w1 >> 16);
            } else {
                prow[u] = fr_row[k];
                pa[u] = fr_ptr[k] - f0;
                pz[u] = fr_ptr[k + 1] - f0;
            }
        }
    }
    for (int i = t; i < nruns && i < kBinShiftCap; i += THREADS)
        shift[i] = b_shift[sp + i];
    if (t == 0)
        long_count = 0;
    __syncthreads();
    SMVP_STAMP(1);
    // what y holds for those rows: on its way while the products are fetched
#pragma unroll
    for (int u = 0; u < kBinPre; ++u)
        py[u] = prow[u] >= 0 ? y[prow[u]] : 0.0;
    for (; base < z; base += STEP) {
        double p[4 * G];
        int slot[4 * G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int wk[4] = {(int)(w[g].x & 0xffffu), (int)(w[g].x >> 16), (int)(w[g].y & 0xffffu), (int)(w[g].y >> 16)};
            const int ck[4] = {cc[g].x, cc[g].y, cc[g].z, cc[g].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + 256 * g + 64 * k + lane;  // logical position
                const bool real = wk[k] != kBinPad;
                const unsigned long long m = __ballot(real && (wk[k] & kBinFlag));
                const int run = ck[k] + __popcll(m & le);
                int sh = 0;
                if (real)
                    sh = run < kBinShiftCap ? shift[run] : b_shift[sp + run];
                p[4 * g + k] = real ? kBinLoadBins(bins + (i + sh)) : 0.0;
                slot[4 * g + k] = real ? (wk[k] & (SLOTS - 1)) : -1;
            }
        }
        if (base + STEP < z)
            load_words(base + STEP);  // (only a block size / workgroup pair with more than one pass comes here)
#pragma unroll
        for (int e = 0; e < 4 * G; ++e)
            if (slot[e] >= 0)
                fp[slot[e]] = p[e];
    }
    SMVP_STAMP(2);
    __syncthreads();
    SMVP_STAMP(3);
    // one lane per far row, left to right (the LDS reads go out four at a time, the adds stay in order); long rows are
    // queued, with what y holds for them, for a whole wavefront
    auto row_sum = [&](int ra, int rz) {
        double acc = 0.0;
        for (int i = ra; i < rz; i += 4) {
            const double v0 = fp[i], v1 = fp[i + 1 < rz ? i + 1 : i], v2 = fp[i + 2 < rz ? i + 2 : i], v3 = fp[i + 3 < rz ? i + 3 : i];
            acc += v0;
            if (i + 1 < rz)
                acc += v1;
            if (i + 2 < rz)
                acc += v2;
            if (i + 3 < rz)
                acc += v3;
        }
        return acc;
    };
#pragma unroll
    for (int u = 0; u < kBinPre; ++u) {
        if (prow[u] < 0)
            continue;
        if (pz[u] - pa[u] <= kBinLongRow) {
            y[prow[u]] = py[u] + row_sum(pa[u], pz[u]);
        } else {
            const int qi = atomicAdd(&long_count, 1);
            long_row[qi] = prow[u], long_a[qi] = pa[u], long_z[qi] = pz[u], long_y[qi] = py[u];
        }
    }
    for (int k = k0 + kBinPre * THREADS + t; k < k1; k += THREADS) {
        int r, ra, rz;
        if (fr32) {
            const unsigned w0 = fr32[k + rb], w1 = fr32[k + rb + 1];
            r = row_base + (int)(w0 & 0xffffu), ra = (int)(w0 >> 16), rz = (int)(w1 >> 16);
        } else {
            r = fr_row[k], ra = fr_ptr[k] - f0, rz = fr_ptr[k + 1] - f0;
        }
        const double yr = y[r];
        if (rz - ra <= kBinLongRow) {
            y[r] = yr + row_sum(ra, rz);
        } else {
            const int qi = atomicAdd(&long_count, 1);
            long_row[qi] = r, long_a[qi] = ra, long_z[qi] = rz, long_y[qi] = yr;
        }
    }
    SMVP_STAMP(4);
    __syncthreads();
    SMVP_STAMP(5);
    const int nlong = long_count;
    for (int qi = wave; qi < nlong; qi += THREADS / 64) {
        double acc = 0.0;
        for (int i = long_a[qi] + lane; i < long_z[qi]; i += 64)
            acc += fp[i];
        acc = wave_sum_dpp(acc);
        if (lane == 0)
            y[long_row[qi]] = long_y[qi] + acc;
    }
    SMVP_STAMP(6);
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

int bits_for(u64 n)  // bits needed for values 0 .. n-1
{
    int b = 1;
    while (b < 63 && (1ull << b) < n)
        ++b;
    return b;
}

__device__ __forceinline__ int row_of_entry(const int *__restrict__ row_ptr, int rows, int e)
{
    int lo = 0, hi = rows - 1;  // last row r with row_ptr[r] <= e (rows without entries share a start: take the last)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (row_ptr[mid] <= e)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

// flag[e] = 1 where |column - row| > band; row_of[e] (optional)
__global__ __launch_bounds__(256) void bin_classify(const int *__restrict__ row_ptr, const int *__restrict__ col_ind, int rows,
                                                    int nnz, int band, long long row0, int *__restrict__ flag, int *__restrict__ row_of)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e > nnz)
        return;
    if (e == nnz) {
        flag[e] = 0;  // so that the scan's last element is the total
        return;
    }
    const int r = row_of_entry(row_ptr, rows, e);
    const long long d = (long long)col_ind[e] - (row0 + r);  // the diagonal of the WHOLE matrix: row0 = this block's first global row
    flag[e] = (d > band || -d > band) ? 1 : 0;
    if (row_of)
        row_of[e] = r;
}

// a row with more than `cap` far entries keeps all of them in the near part
__global__ __launch_bounds__(256) void bin_cap_rows(const int *__restrict__ row_ptr, const int *__restrict__ fpos, int rows, int cap,
                                                    const int *__restrict__ row_of, int nnz, int *__restrict__ flag)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz || !flag[e])
        return;
    const int r = row_of[e];
    if (fpos[row_ptr[r + 1]] - fpos[row_ptr[r]] > cap)
        flag[e] = 0;
}

__global__ __launch_bounds__(256) void bin_split(const int *__restrict__ col_ind, const double *__restrict__ val,
                                                 const int *__restrict__ flag, const int *__restrict__ fpos,
                                                 const int *__restrict__ row_of, int nnz, int *__restrict__ near_col,
                                                 double *__restrict__ near_val, int *__restrict__ f_col, double *__restrict__ f_val,
                                                 int *__restrict__ f_row)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz)
        return;
    const int f = fpos[e];
    if (flag[e]) {
        f_col[f] = col_ind[e];
        f_val[f] = val[e];
        f_row[f] = row_of[e];
    } else {
        near_col[e - f] = col_ind[e];
        near_val[e - f] = val[e];
    }
}

// frp[r] = far entries in front of row r; near_ptr[r]; has[r] = the row has far entries
__global__ __launch_bounds__(256) void bin_row_ptrs(const int *__restrict__ row_ptr, const int *__restrict__ fpos, int rows,
                                                    int *__restrict__ frp, int *__restrict__ near_ptr, int *__restrict__ has)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r > rows)
        return;
    const int f = fpos[row_ptr[r]];
    frp[r] = f;
    near_ptr[r] = row_ptr[r] - f;
    has[r] = r < rows && fpos[row_ptr[r + 1]] > f ? 1 : 0;
}

__global__ __launch_bounds__(256) void bin_far_rows(const int *__restrict__ frp, const int *__restrict__ has,
                                                    const int *__restrict__ kpos, int rows, int nf, int *__restrict__ fr_row,
                                                    int *__restrict__ fr_ptr)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r > rows)
        return;
    if (r == rows) {
        fr_ptr[kpos[rows]] = nf;  // kpos[rows] = number of far rows
        return;
    }
    if (has[r]) {
        fr_row[kpos[r]] = r;
        fr_ptr[kpos[r]] = frp[r];
    }
}

// blk_fr[b] = first far row whose first far entry lies at or after b * bucket
__global__ __launch_bounds__(256) void bin_block_rows(const int *__restrict__ fr_ptr, int nfr, int nrb, int bucket,
                                                      int *__restrict__ blk_fr)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b > nrb)
        return;
    const long long want = (long long)b * bucket;
    int lo = 0, hi = nfr;  // first k in [0, nfr] with fr_ptr[k] >= want (fr_ptr[nfr] = nf)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (fr_ptr[mid] >= want)
            hi = mid;
        else
            lo = mid + 1;
    }
    blk_fr[b] = b == nrb ? nfr : lo;
}

// which: 0 = the bins' order (super block, column block), 1 = stream A (column block, super block), 2 = stream B (row block, column block)
__global__ __launch_bounds__(256) void bin_keys(const int *__restrict__ f_col, const int *__restrict__ f_row,
                                                const int *__restrict__ frp, int nf, int bucket, int q, int ncb, int nsb, int which,
                                                u64 *__restrict__ key, unsigned *__restrict__ idx)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= nf)
        return;
    const u64 cb = (u64)(f_col[f] >> kBinColBits);
    const u64 rb = (u64)(frp[f_row[f]] / bucket);
    const u64 sb = rb / (u64)q;
    key[f] = which == 0 ? sb * (u64)ncb + cb : which == 1 ? cb * (u64)nsb + sb : rb * (u64)ncb + cb;
    idx[f] = (unsigned)f;
}

__global__ __launch_bounds__(256) void bin_invert(const unsigned *__restrict__ order, int nf, int *__restrict__ binpos)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < nf)
        binpos[order[p]] = p;
}

__global__ __launch_bounds__(256) void bin_cell_flags(const u64 *__restrict__ key, int nf, int *__restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > nf)
        return;
    flag[i] = i < nf && (i == 0 || key[i] != key[i - 1]) ? 1 : 0;  // flag[nf] = 0: the scan's last element is the total
}

// ustart[g] = first sorted position whose group (key / div) is g or later; psz[g] = the group's length rounded up to 256
__global__ __launch_bounds__(256) void bin_group_starts(const u64 *__restrict__ key, int nf, u64 div, int groups,
                                                        int *__restrict__ ustart)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g > groups)
        return;
    const u64 want = (u64)g * div;
    int lo = 0, hi = nf;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key[mid] >= want)
            hi = mid;
        else
            lo = mid + 1;
    }
    ustart[g] = lo;
}

__global__ __launch_bounds__(256) void bin_group_sizes(const int *__restrict__ ustart, const int *__restrict__ cellno, int groups,
                                                       int *__restrict__ psz, int *__restrict__ shift_ptr)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g > groups)
        return;
    psz[g] = g < groups ? ((ustart[g + 1] - ustart[g] + 255) / 256) * 256 : 0;  // whole groups of 256 (see stream_value_index)
    shift_ptr[g] = cellno[ustart[g]];  // cells in front of the group (cellno[nf] = all cells)
}

// writes the stream: word, chunk, shift (and pass A's values)
__global__ __launch_bounds__(256) void bin_emit(const u64 *__restrict__ key, const unsigned *__restrict__ order,
                                                const int *__restrict__ flag, const int *__restrict__ cellno,
                                                const int *__restrict__ ustart, const int *__restrict__ ptr,
                                                const int *__restrict__ shift_ptr, const int *__restrict__ binpos, int nf, u64 div,
                                                bool pass_a, const int *__restrict__ f_col, const double *__restrict__ f_val,
                                                const int *__restrict__ blk_fr, const int *__restrict__ fr_ptr,
                                                unsigned short *__restrict__ word, int *__restrict__ chunk,
                                                int *__restrict__ shift, double *__restrict__ a_val)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nf)
        return;
    const int f = (int)order[i];
    const int g = (int)(key[i] / div);
    const int spos = ptr[g] + (i - ustart[g]);
    const int fl = flag[i];
    if (fl)
        shift[cellno[i]] = binpos[f] - spos;
    if ((spos & 63) == 0)
        chunk[spos >> 6] = cellno[i] - shift_ptr[g] - 1;
    if (pass_a) {
        word[stream_word_index(spos)] = (unsigned short)((f_col[f] & ((1 << kBinColBits) - 1)) | (fl ? kBinFlag : 0));
        a_val[stream_value_index(spos)] = f_val[f];
    } else {
        word[stream_word_index(spos)] = (unsigned short)((f - fr_ptr[blk_fr[g]]) | (fl ? kBinFlag : 0));
    }
}

// b_desc[2 * rb] = {a, z, k0, k1}, b_desc[2 * rb + 1] = {f0, sp, nruns, 0}: what a workgroup of pass B reads first
__global__ __launch_bounds__(256) void bin_block_desc(const int *__restrict__ b_ptr, const int *__restrict__ b_shift_ptr,
                                                      const int *__restrict__ blk_fr, const int *__restrict__ fr_ptr,
                                                      const int *__restrict__ fr_row, int nrb, int4 *__restrict__ desc)
{
    const int rb = blockIdx.x * 256 + threadIdx.x;
    if (rb >= nrb)
        return;
    const int k0 = blk_fr[rb], k1 = blk_fr[rb + 1], sp = b_shift_ptr[rb];
    desc[2 * rb] = make_int4(b_ptr[rb], b_ptr[rb + 1], k0, k1);
    desc[2 * rb + 1] = make_int4(fr_ptr[k0], sp, b_shift_ptr[rb + 1] - sp, k1 > k0 ? fr_row[k0] : 0);
}

// The far rows' list as pass B reads it: one 32-bit word per far row -- (row - the block's first far row) | (first slot << 16) --
// and one sentinel per block carrying the block's entry count, so that a row ends where the next word begins: entry k of
// block rb sits at k + rb.  *bad is set when a block's rows span 65536 or more (the 32-bit lists stay in use then).
__global__ __launch_bounds__(256) void bin_far_rows_packed(const int *__restrict__ fr_row, const int *__restrict__ fr_ptr,
                                                           const int *__restrict__ blk_fr, int nfr, int nrb,
                                                           unsigned *__restrict__ fr32, int *__restrict__ bad)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nfr)
        return;
    int lo = 0, hi = nrb - 1;  // the block of far row k: last rb with blk_fr[rb] <= k
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk_fr[mid] <= k)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int rb = lo, k0 = blk_fr[rb], k1 = blk_fr[rb + 1];
    const int rel = fr_row[k] - fr_row[k0], slot = fr_ptr[k] - fr_ptr[k0];
    if (rel > 0xffff || slot > 0xffff)
        atomicExch(bad, 1);
    fr32[k + rb] = (unsigned)(rel & 0xffff) | ((unsigned)(slot & 0xffff) << 16);
    if (k == k1 - 1)
        fr32[k1 + rb] = (unsigned)((fr_ptr[k1] - fr_ptr[k0]) & 0xffff) << 16;
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

int sort_pairs(u64 *k0, u64 *k1, unsigned *i0, unsigned *i1, int n, unsigned bits, Scratch &sc, hipStream_t st)
{
    size_t bytes = 0;
    HIP_TRY(smvp::prim::radix_sort_pairs(nullptr, bytes, k0, k1, i0, i1, (size_t)n, 0u, bits, st));
    char *tmp;
    HIP_TRY(sc.get(&tmp, bytes));
    HIP_TRY(smvp::prim::radix_sort_pairs(tmp, bytes, k0, k1, i0, i1, (size_t)n, 0u, bits, st));
    return SMVP_OK;
}

template <class T>
int own(T **out, size_t count, size_t *bytes)
{
    if (hipMalloc((void **)out, std::max<size_t>(count, 4) * sizeof(T)) != hipSuccess) {
        *out = nullptr;
        return smvp::fail(SMVP_ERR_ALLOC, "cannot allocate the binned plan (%zu bytes)", count * sizeof(T));
    }
    *bytes += count * sizeof(T);
    return SMVP_OK;
}

// One grouped stream from the sorted (key, far entry) pairs; group = key / div, a cell = a run of equal keys.
int build_stream(const u64 *key, const unsigned *order, int nf, u64 div, int groups, const int *binpos, bool pass_a,
                 const int *f_col, const double *f_val, const int *blk_fr, const int *fr_ptr, BinnedStream *s, double **a_val,
                 size_t *plan_bytes, hipStream_t st)
{
    Scratch sc;
    int *flag, *cellno, *ustart, *psz;
    HIP_TRY(sc.get(&flag, (size_t)nf + 1));
    HIP_TRY(sc.get(&cellno, (size_t)nf + 1));
    HIP_TRY(sc.get(&ustart, (size_t)groups + 2));
    HIP_TRY(sc.get(&psz, (size_t)groups + 2));
    s->groups = groups;
    if (int rc = own(&s->ptr, (size_t)groups + 2, plan_bytes))
        return rc;
    if (int rc = own(&s->shift_ptr, (size_t)groups + 2, plan_bytes))
        return rc;
    hipLaunchKernelGGL(bin_cell_flags, dim3(blocks_for((long long)nf + 1)), dim3(256), 0, st, key, nf, flag);
    HIP_TRY(hipGetLastError());
    if (int rc = scan_exclusive(flag, cellno, (size_t)nf + 1, sc, st))
        return rc;
    hipLaunchKernelGGL(bin_group_starts, dim3(blocks_for((long long)groups + 1)), dim3(256), 0, st, key, nf, div, groups, ustart);
    hipLaunchKernelGGL(bin_group_sizes, dim3(blocks_for((long long)groups + 1)), dim3(256), 0, st, ustart, cellno, groups, psz,
                       s->shift_ptr);
    HIP_TRY(hipGetLastError());
    if (int rc = scan_exclusive(psz, s->ptr, (size_t)groups + 1, sc, st))
        return rc;
    int padded = 0, cells = 0;
    HIP_TRY(hipMemcpyAsync(&padded, s->ptr + groups, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&cells, cellno + nf, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    s->padded = padded, s->cells = cells;
    if (int rc = own(&s->word, (size_t)padded, plan_bytes))
        return rc;
    if (int rc = own(&s->chunk, (size_t)padded / 64 + 1, plan_bytes))
        return rc;
    if (int rc = own(&s->shift, (size_t)cells, plan_bytes))
        return rc;
    HIP_TRY(hipMemsetAsync(s->word, 0xff, std::max<size_t>((size_t)padded, 4) * sizeof(unsigned short), st));
    HIP_TRY(hipMemsetAsync(s->chunk, 0, ((size_t)padded / 64 + 1) * sizeof(int), st));  // chunks of padding only are never counted from
    if (pass_a) {
        if (int rc = own(a_val, (size_t)padded, plan_bytes))
            return rc;
        HIP_TRY(hipMemsetAsync(*a_val, 0, std::max<size_t>((size_t)padded, 4) * sizeof(double), st));
    }
    hipLaunchKernelGGL(bin_emit, dim3(blocks_for(nf)), dim3(256), 0, st, key, order, flag, cellno, ustart, s->ptr, s->shift_ptr,
                       binpos, nf, div, pass_a, f_col, f_val, blk_fr, fr_ptr, s->word, s->chunk, s->shift, pass_a ? *a_val : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return SMVP_OK;
}

void free_stream(BinnedStream *s)
{
    for (void *p : {(void *)s->word, (void *)s->chunk, (void *)s->ptr, (void *)s->shift_ptr, (void *)s->shift})
        if (p)
            (void)hipFree(p);
    *s = BinnedStream();
}

}  // namespace

void free_binned_plan(BinnedPlan *p)
{
    if (!p)
        return;
    for (void *q : {(void *)p->near_ptr, (void *)p->near_col, (void *)p->near_val, (void *)p->a_val, (void *)p->bins, (void *)p->fr_row,
                    (void *)p->fr_ptr, (void *)p->blk_fr, (void *)p->b_desc, (void *)p->fr32})
        if (q)
            (void)hipFree(q);
    free_stream(&p->a);
    free_stream(&p->b);
    free_near_window(&p->nw);
    *p = BinnedPlan();
}

int csr_far_share(const int *d_row_ptr, const int *d_col_ind, int rows, int nnz, int band, long long row0, double *share, hipStream_t st)
{
    *share = 0.0;
    if (nnz <= 0 || rows <= 0)
        return SMVP_OK;
    Scratch sc;
    int *flag, *fpos;
    HIP_TRY(sc.get(&flag, (size_t)nnz + 1));
    HIP_TRY(sc.get(&fpos, (size_t)nnz + 1));
    hipLaunchKernelGGL(bin_classify, dim3(blocks_for((long long)nnz + 1)), dim3(256), 0, st, d_row_ptr, d_col_ind, rows, nnz, band,
                       row0, flag, (int *)nullptr);
    HIP_TRY(hipGetLastError());
    if (int rc = scan_exclusive(flag, fpos, (size_t)nnz + 1, sc, st))
        return rc;
    int nf = 0;
    HIP_TRY(hipMemcpyAsync(&nf, fpos + nnz, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *share = (double)nf / nnz;
    return SMVP_OK;
}

// capped[r] = row r has more than `cap` far entries (before the cap is applied: fpos from the first scan)
__global__ __launch_bounds__(256) void bin_capped_rows(const int *__restrict__ row_ptr, const int *__restrict__ fpos, int rows, int cap,
                                                       int *__restrict__ capped)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < rows)
        capped[r] = fpos[row_ptr[r + 1]] - fpos[row_ptr[r]] > cap ? 1 : 0;
}

int build_binned_plan(const int *d_row_ptr, const int *d_col_ind, const double *d_val, int rows, int cols, int nnz, int band,
                      long long row0, bool near_window, BinnedPlan *out, hipStream_t st)
{
    free_binned_plan(out);
    BinnedPlan &P = *out;
    P.band = band > 0 ? band : kBinNearBand;
    P.row0 = row0;
    P.slots = kBinSlots, P.threads_b = kBinThreads;
    const int bucket = P.slots - P.slots / 8;
    P.rows = rows, P.cols = cols, P.nnz = nnz;
    P.ncb = (int)(((long long)cols + (1 << kBinColBits) - 1) >> kBinColBits);
    P.plan_bytes = 0;
    Scratch sc;
    if (int rc = own(&P.near_ptr, (size_t)rows + 2, &P.plan_bytes))
        return rc;
    if (rows <= 0 || nnz <= 0) {
        HIP_TRY(hipMemsetAsync(P.near_ptr, 0, ((size_t)rows + 2) * sizeof(int), st));
        if (int rc = own(&P.near_col, 4, &P.plan_bytes))
            return rc;
        if (int rc = own(&P.near_val, 4, &P.plan_bytes))
            return rc;
        HIP_TRY(hipStreamSynchronize(st));
        return SMVP_OK;
    }
    // ---- classify, cap, count
    int *flag, *fpos, *row_of;
    HIP_TRY(sc.get(&flag, (size_t)nnz + 1));
    HIP_TRY(sc.get(&fpos, (size_t)nnz + 1));
    HIP_TRY(sc.get(&row_of, (size_t)nnz));
    hipLaunchKernelGGL(bin_classify, dim3(blocks_for((long long)nnz + 1)), dim3(256), 0, st, d_row_ptr, d_col_ind, rows, nnz, P.band,
                       row0, flag, row_of);
    HIP_TRY(hipGetLastError());
    if (int rc = scan_exclusive(flag, fpos, (size_t)nnz + 1, sc, st))
        return rc;
    int *capped;
    HIP_TRY(sc.get(&capped, (size_t)rows + 1));
    hipLaunchKernelGGL(bin_capped_rows, dim3(blocks_for(rows)), dim3(256), 0, st, d_row_ptr, fpos, rows, P.slots / 8, capped);
    hipLaunchKernelGGL(bin_cap_rows, dim3(blocks_for(nnz)), dim3(256), 0, st, d_row_ptr, fpos, rows, P.slots / 8, row_of, nnz, flag);
    HIP_TRY(hipGetLastError());
    if (int rc = scan_exclusive(flag, fpos, (size_t)nnz + 1, sc, st))
        return rc;
    int nf = 0;
    HIP_TRY(hipMemcpyAsync(&nf, fpos + nnz, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    P.nf = nf, P.nnz_near = nnz - nf;
    if ((long long)nf + 256ll * (P.ncb + (long long)nf / bucket + 2) > 2147483000ll)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "binned plan: %d far entries over %d column blocks do not fit 32-bit stream positions", nf, P.ncb);
    // ---- split
    if (int rc = own(&P.near_col, (size_t)P.nnz_near, &P.plan_bytes))
        return rc;
    if (int rc = own(&P.near_val, (size_t)P.nnz_near, &P.plan_bytes))
        return rc;
    int *f_col, *f_row, *frp, *has, *kpos;
    double *f_val;
    HIP_TRY(sc.get(&f_col, (size_t)nf));
    HIP_TRY(sc.get(&f_row, (size_t)nf));
    HIP_TRY(sc.get(&f_val, (size_t)nf));
    HIP_TRY(sc.get(&frp, (size_t)rows + 2));
    HIP_TRY(sc.get(&has, (size_t)rows + 2));
    HIP_TRY(sc.get(&kpos, (size_t)rows + 2));
    hipLaunchKernelGGL(bin_split, dim3(blocks_for(nnz)), dim3(256), 0, st, d_col_ind, d_val, flag, fpos, row_of, nnz, P.near_col,
                       P.near_val, f_col, f_val, f_row);
    hipLaunchKernelGGL(bin_row_ptrs, dim3(blocks_for((long long)rows + 1)), dim3(256), 0, st, d_row_ptr, fpos, rows, frp, P.near_ptr, has);
    HIP_TRY(hipGetLastError());
    // the near part's window plan, where it suits: the near arrays are then not needed any more
    auto window = [&]() -> int {
        if (!near_window)
            return SMVP_OK;
        if (int rc = build_near_window(P.near_ptr, P.near_col, P.near_val, capped, rows, cols, P.nnz_near, P.band, row0, &P.nw, st))
            return rc;
        if (P.nw.on) {
            (void)hipFree(P.near_col);
            (void)hipFree(P.near_val);
            P.near_col = nullptr, P.near_val = nullptr;
            P.plan_bytes -= (size_t)P.nnz_near * (sizeof(int) + sizeof(double));
            P.plan_bytes += P.nw.plan_bytes;
        }
        return SMVP_OK;
    };
    if (nf == 0) {
        HIP_TRY(hipStreamSynchronize(st));
        return window();  // nothing is far: the near part is the whole matrix
    }
    // ---- far rows and row blocks
    if (int rc = scan_exclusive(has, kpos, (size_t)rows + 1, sc, st))
        return rc;
    int nfr = 0;
    HIP_TRY(hipMemcpyAsync(&nfr, kpos + rows, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    P.nfr = nfr;
    P.nrb = (nf + bucket - 1) / bucket;
    // cells of about 112 entries and more: q row blocks per super block (a cell holds q * bucket / ncb entries on average)
    P.q = (int)std::min<long long>(64, std::max<long long>(1, (112ll * P.ncb + bucket - 1) / bucket));
    const int nsb = (P.nrb + P.q - 1) / P.q;
    P.splits = std::max(1, std::min(16, 512 / std::max(P.ncb, 1)));

    if (int rc = own(&P.fr_row, (size_t)nfr + 1, &P.plan_bytes))
        return rc;
    if (int rc = own(&P.fr_ptr, (size_t)nfr + 2, &P.plan_bytes))
        return rc;
    if (int rc = own(&P.blk_fr, (size_t)P.nrb + 2, &P.plan_bytes))
        return rc;
    if (int rc = own(&P.bins, (size_t)nf + 64, &P.plan_bytes))
        return rc;
    hipLaunchKernelGGL(bin_far_rows, dim3(blocks_for((long long)rows + 1)), dim3(256), 0, st, frp, has, kpos, rows, nf, P.fr_row, P.fr_ptr);
    hipLaunchKernelGGL(bin_block_rows, dim3(blocks_for((long long)P.nrb + 1)), dim3(256), 0, st, P.fr_ptr, nfr, P.nrb, bucket, P.blk_fr);
    HIP_TRY(hipGetLastError());
    // ---- the three orders of the far entries
    u64 *k0, *k1;
    unsigned *i0, *i1;
    int *binpos;
    HIP_TRY(sc.get(&k0, (size_t)nf));
    HIP_TRY(sc.get(&k1, (size_t)nf));
    HIP_TRY(sc.get(&i0, (size_t)nf));
    HIP_TRY(sc.get(&i1, (size_t)nf));
    HIP_TRY(sc.get(&binpos, (size_t)nf));
    auto sorted = [&](int which, u64 key_count) -> int {
        hipLaunchKernelGGL(bin_keys, dim3(blocks_for(nf)), dim3(256), 0, st, f_col, f_row, frp, nf, bucket, P.q, P.ncb, nsb, which, k0, i0);
        HIP_TRY(hipGetLastError());
        return sort_pairs(k0, k1, i0, i1, nf, (unsigned)bits_for(key_count), sc, st);
    };
    if (int rc = sorted(0, (u64)nsb * (u64)P.ncb))
        return rc;
    hipLaunchKernelGGL(bin_invert, dim3(blocks_for(nf)), dim3(256), 0, st, i1, nf, binpos);
    HIP_TRY(hipGetLastError());
    if (int rc = sorted(1, (u64)P.ncb * (u64)nsb))
        return rc;
    if (int rc = build_stream(k1, i1, nf, (u64)nsb, P.ncb, binpos, true, f_col, f_val, nullptr, nullptr, &P.a, &P.a_val, &P.plan_bytes, st))
        return rc;
    if (int rc = sorted(2, (u64)P.nrb * (u64)P.ncb))
        return rc;
    if (int rc = build_stream(k1, i1, nf, (u64)P.ncb, P.nrb, binpos, false, nullptr, nullptr, P.blk_fr, P.fr_ptr, &P.b, nullptr, &P.plan_bytes, st))
        return rc;
    if (int rc = own(&P.b_desc, (size_t)P.nrb * 2 + 2, &P.plan_bytes))
        return rc;
    hipLaunchKernelGGL(bin_block_desc, dim3(blocks_for(P.nrb)), dim3(256), 0, st, P.b.ptr, P.b.shift_ptr, P.blk_fr, P.fr_ptr, P.fr_row, P.nrb, P.b_desc);
    HIP_TRY(hipGetLastError());
    // the packed far-row list (4 instead of 8 bytes per far row for pass B to read); the 32-bit lists go unless a block's rows span too far
    {
        int *bad, h_bad = 0;
        HIP_TRY(sc.get(&bad, 1));
        HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), st));
        if (int rc = own(&P.fr32, (size_t)nfr + (size_t)P.nrb + 2, &P.plan_bytes))
            return rc;
        hipLaunchKernelGGL(bin_far_rows_packed, dim3(blocks_for(nfr)), dim3(256), 0, st, P.fr_row, P.fr_ptr, P.blk_fr, nfr, P.nrb, P.fr32, bad);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_bad) {
            (void)hipFree(P.fr32);
            P.fr32 = nullptr;
            P.plan_bytes -= ((size_t)nfr + (size_t)P.nrb + 2) * sizeof(unsigned);
        } else {
            (void)hipFree(P.fr_row);
            (void)hipFree(P.fr_ptr);
            P.fr_row = nullptr, P.fr_ptr = nullptr;
            P.plan_bytes -= ((size_t)nfr + 1) * sizeof(int) + ((size_t)nfr + 2) * sizeof(int);
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    return window();
}

// more than 64 KB of dynamic LDS must be asked for, once per device (the answer is remembered for ordinals below 64; beyond
// that the question is simply asked again)
static hipError_t ask_for_lds()
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    static std::atomic<unsigned long long> asked{0};
    if (dev >= 64 || !(asked.load() >> dev & 1ull)) {
        e = hipFuncSetAttribute((const void *)csr_binned_far_products<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsA);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)csr_binned_far_sums<8192, 1024, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b_bytes(8192));
        if (e != hipSuccess)
            return e;
        if (dev < 64)
            asked.fetch_or(1ull << dev);
    }
    return hipSuccess;
}

hipError_t binned_reserve_lds()
{
    const hipError_t e = ask_for_lds();
    return e != hipSuccess ? e : near_window_reserve_lds();
}

// pass A: every far product into the bins (reads x; independent of the near product, which may run beside it)
hipError_t launch_binned_products(const BinnedPlan &p, const double *x, hipStream_t stream)
{
    if (p.nf <= 0)
        return hipSuccess;
    hipError_t e = ask_for_lds();
    if (e != hipSuccess)
        return e;
    const int items = p.ncb * p.splits;
    hipLaunchKernelGGL(csr_binned_far_products<2>, dim3((unsigned)items), dim3(kBinThreads), kLdsA, stream, x, p.cols,
                       p.a_val, p.a.word, p.a.chunk, p.a.ptr, p.a.shift_ptr, p.a.shift, p.bins, p.splits, items);
    return hipGetLastError();
}

// pass B: y[row] += the row's far sum, for every row with far entries (after pass A and after the near product has written y)
hipError_t launch_binned_sums(const BinnedPlan &p, double *y, hipStream_t stream)
{
    if (p.nf <= 0)
        return hipSuccess;
    hipError_t e = ask_for_lds();
    if (e != hipSuccess)
        return e;
    const unsigned grid_b = (unsigned)((p.nrb + 8 * p.q - 1) / (8 * p.q)) * 8u * (unsigned)p.q;
#define SMVP_BINNED_B(S, T, GG)                                                                                                  \
    if (p.slots == S && p.threads_b == T) {                                                                                    \
        hipLaunchKernelGGL((csr_binned_far_sums<S, T, GG>), dim3(grid_b), dim3(T), lds_b_bytes(S), stream, p.bins, p.b.word, p.b.chunk, \
                           p.b_desc, p.b.shift, p.fr_row, p.fr_ptr, p.fr32, y, p.nrb, p.q);                                              \
        return hipGetLastError();                                                                                              \
    }
    SMVP_BINNED_B(8192, 1024, 2)   // (blocks of 4096 / 2048 slots with 512 / 256 threads were measured and are slower: 240 / 300 us against 215)
#undef SMVP_BINNED_B
    return hipErrorInvalidValue;
}

#ifdef SMVP_PHASE_STAMPS
void debug_binned_phases()
{
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_binned_phase), sizeof h) != hipSuccess || h[7] == 0)
        return;
    const double n = (double)h[7];
    fprintf(stderr, "[binned dbg] pass B, mean us per workgroup over %.0f: to barrier 1 %.2f | products into LDS %.2f | barrier 2 %.2f | "
                    "row sums %.2f | barrier 3 %.2f | long rows %.2f\n", n, h[0] * 0.01 / n, h[1] * 0.01 / n, h[2] * 0.01 / n, h[3] * 0.01 / n,
            h[4] * 0.01 / n, h[5] * 0.01 / n);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_binned_phase), z, sizeof z);
}
#else
void debug_binned_phases() {}
#endif

}  // namespace smvp
