// smvp_prim.h -- the two device-wide primitives the plan builders and the device-side converters need, hand-written for
// gfx950 (wave64): a STABLE least-significant-digit radix sort of (key, 32-bit value) pairs and prefix sums of ints.
//
// Set-up work only (COO -> CSR / TJDS on the device, main-cli.c:340-365 / :766-967; the launch plans of the binned product, the
// near-window kernel and the column sweep): nothing here runs inside a timed product.  Until round 5 these were rocPRIM calls;
// the call shape is kept (a first call with tmp == nullptr returns the bytes of temporary storage, the second does the work).
//
// Sort: 8 bits per pass.  A workgroup of four wavefronts takes a tile of 2048 consecutive elements, wavefront w the 512 from
// w * 512 on, 64 at a time -- so "wavefront, round, lane" IS the element order, and a pass is stable by construction:
//   pass = histogram (digit counts per tile, LDS atomics) -> exclusive scan of the counts in (digit, tile) order -> scatter:
//   a lane finds the lanes of its wavefront that hold the same digit with eight ballots (its rank among them = the popcount
//   of the lower ones), the lowest such lane keeps the wavefront's running count of the digit in LDS, the four wavefronts'
//   totals are prefixed per digit, and every element goes to base(digit, tile) + earlier wavefronts + earlier rounds + rank.
// The input arrays are never written (passes alternate between the output and a second buffer in the temporary storage, laid
// out so that the last pass lands in the output).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace smvp {
namespace prim {

constexpr int kScanBlock = 256, kScanItems = 4, kScanTile = kScanBlock * kScanItems;
#ifndef SMVP_SORT_ITEMS
#define SMVP_SORT_ITEMS 8  // elements per thread and pass (tests/prim_check.hip times other values)
#endif
constexpr int kSortBlock = 256, kSortItems = SMVP_SORT_ITEMS, kSortTile = kSortBlock * kSortItems, kSortWaves = kSortBlock / 64;
constexpr int kRadixBits = 8, kRadix = 1 << kRadixBits;

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------------------------------------ prefix sums
// one tile of kScanTile ints: out (unless null) = exclusive (or inclusive) prefix inside the tile (+ carry[tile] where given),
// sums[tile] (unless null) = the tile's total
template <bool INCLUSIVE>
__global__ __launch_bounds__(kScanBlock) void scan_tiles(const int *in, int *out, size_t n, int init,  // (in may be out: no __restrict__)
                                                         int *__restrict__ sums, const int *__restrict__ carry)
{
    __shared__ int wave_total[kScanBlock / 64];
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
    int v[kScanItems], mine = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = base + k < n ? in[base + k] : 0;
        mine += v[k];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = mine;  // inclusive prefix of the lanes' sums inside the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off)
            incl += up;
    }
    if (lane == 63)
        wave_total[wave] = incl;
    __syncthreads();
    int before = carry ? carry[blockIdx.x] : init;
    for (int w = 0; w < wave; ++w)
        before += wave_total[w];
    int run = before + incl - mine;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (INCLUSIVE)
            run += v[k];
        if (out && base + k < n)
            out[base + k] = run;
        if (!INCLUSIVE)
            run += v[k];
    }
    if (sums && threadIdx.x == kScanBlock - 1) {
        int total = 0;
        for (int w = 0; w < kScanBlock / 64; ++w)
            total += wave_total[w];
        sums[blockIdx.x] = total;
    }
}

inline size_t scan_tmp_bytes(size_t n)
{
    size_t bytes = 0;
    for (size_t m = (n + kScanTile - 1) / kScanTile; m > 1; m = (m + kScanTile - 1) / kScanTile)
        bytes += 2 * align_up(m * sizeof(int));  // this level's tile totals and their exclusive prefix
    return bytes + 256;
}

// out[i] = init + in[0] + ... + in[i - 1] (exclusive) or ... + in[i] (inclusive).  in == out is allowed.
template <bool INCLUSIVE>
inline hipError_t scan(void *tmp, size_t &bytes, const int *in, int *out, int init, size_t n, hipStream_t st)
{
    if (!tmp) {
        bytes = scan_tmp_bytes(n);
        return hipSuccess;
    }
    if (n == 0)
        return hipSuccess;
    const size_t tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles == 1) {
        hipLaunchKernelGGL(scan_tiles<INCLUSIVE>, dim3(1), dim3(kScanBlock), 0, st, in, out, n, init, (int *)nullptr, (const int *)nullptr);
        return hipGetLastError();
    }
    // tile totals -> their exclusive prefix (recursively, starting from `init`) -> the tiles again with their carries
    int *totals = static_cast<int *>(tmp);
    int *carries = reinterpret_cast<int *>(static_cast<char *>(tmp) + align_up(tiles * sizeof(int)));
    char *rest = static_cast<char *>(tmp) + 2 * align_up(tiles * sizeof(int));
    // (first sweep: the tiles' totals only -- out == nullptr writes no prefix; `in` may be `out`, the last sweep reads every
    // element into a register before the same thread overwrites it)
    hipLaunchKernelGGL(scan_tiles<false>, dim3((unsigned)tiles), dim3(kScanBlock), 0, st, in, (int *)nullptr, n, 0, totals,
                       (const int *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    size_t sub = 0;
    e = scan<false>(rest, sub, totals, carries, init, tiles, st);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(scan_tiles<INCLUSIVE>, dim3((unsigned)tiles), dim3(kScanBlock), 0, st, in, out, n, 0, (int *)nullptr, (const int *)carries);
    return hipGetLastError();
}

inline hipError_t exclusive_scan(void *tmp, size_t &bytes, const int *in, int *out, int init, size_t n, hipStream_t st)
{
    return scan<false>(tmp, bytes, in, out, init, n, st);
}
inline hipError_t inclusive_scan(void *tmp, size_t &bytes, const int *in, int *out, size_t n, hipStream_t st)
{
    return scan<true>(tmp, bytes, in, out, 0, n, st);
}

// ------------------------------------------------------------------------------------------------------------ radix sort
template <class K>
__global__ __launch_bounds__(kSortBlock) void sort_histogram(const K *__restrict__ keys, size_t n, unsigned shift, unsigned mask,
                                                             int *__restrict__ counts, unsigned ntiles)
{
    __shared__ int hist[kRadix];
    hist[threadIdx.x] = 0;  // (kSortBlock == kRadix)
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kSortTile;
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const size_t i = base + (size_t)k * kSortBlock + threadIdx.x;  // (order does not matter for counting)
        if (i < n)
            atomicAdd(&hist[(unsigned)(keys[i] >> shift) & mask], 1);
    }
    __syncthreads();
    counts[(size_t)threadIdx.x * ntiles + blockIdx.x] = hist[threadIdx.x];
}

template <class K, class V>
__global__ __launch_bounds__(kSortBlock) void sort_scatter(const K *__restrict__ kin, const V *__restrict__ vin, K *__restrict__ kout,
                                                           V *__restrict__ vout, size_t n, unsigned shift, unsigned mask,
                                                           const int *__restrict__ bases, unsigned ntiles)
{
    __shared__ int wcount[kSortWaves][kRadix];  // per wavefront and digit: running count, then the wavefronts' exclusive prefix
    __shared__ int gbase[kRadix];
    for (int i = threadIdx.x; i < kSortWaves * kRadix; i += kSortBlock)
        (&wcount[0][0])[i] = 0;
    gbase[threadIdx.x] = bases[(size_t)threadIdx.x * ntiles + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const size_t first = (size_t)blockIdx.x * kSortTile + (size_t)wave * (kSortTile / kSortWaves);
    K key[kSortItems];
    int rank[kSortItems];
    unsigned digit[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const size_t i = first + (size_t)r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? kin[i] : (K)0;
        const unsigned d = valid ? (unsigned)(key[r] >> shift) & mask : 0u;
        digit[r] = d;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const unsigned long long has = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? has : ~has;
        }
        // (an invalid lane's `same` is empty: it takes no part)
        const int leader = same ? __ffsll((long long)same) - 1 : lane;
        int run = 0;
        if (valid && lane == leader) {
            run = wcount[wave][d];
            wcount[wave][d] = run + __popcll(same);
        }
        run = __shfl(run, leader, 64);
        rank[r] = run + __popcll(same & below);
    }
    __syncthreads();
    {   // per digit: the wavefronts' totals -> exclusive prefix over the wavefronts (thread d = digit d)
        int acc = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const int c = wcount[w][threadIdx.x];
            wcount[w][threadIdx.x] = acc;
            acc += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const size_t i = first + (size_t)r * 64 + lane;
        if (i < n) {
            const size_t pos = (size_t)gbase[digit[r]] + wcount[wave][digit[r]] + rank[r];
            kout[pos] = key[r];
            vout[pos] = vin[i];
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void copy_array(const T *__restrict__ in, T *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = in[i];
}

// Sorts the n pairs by bits [begin_bit, end_bit) of the key, ascending, stable; results in kout / vout; kin / vin untouched.
template <class K, class V>
inline hipError_t radix_sort_pairs(void *tmp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, size_t n, unsigned begin_bit,
                                   unsigned end_bit, hipStream_t st)
{
    static_assert(sizeof(V) == 4 && (sizeof(K) == 4 || sizeof(K) == 8), "32-bit values, 32- or 64-bit unsigned keys");
    static_assert(kSortBlock == kRadix, "one thread per digit");
    const size_t ntiles = (n + kSortTile - 1) / kSortTile;
    const size_t counts_n = ntiles * kRadix;
    size_t scan_bytes = 0;
    (void)scan<false>(nullptr, scan_bytes, nullptr, nullptr, 0, counts_n, st);
    const size_t off_counts = 0, off_scan = off_counts + align_up(counts_n * sizeof(int)), off_keys = off_scan + align_up(scan_bytes),
                 off_vals = off_keys + align_up(n * sizeof(K)), total = off_vals + align_up(n * sizeof(V));
    if (!tmp) {
        bytes = total + 256;
        return hipSuccess;
    }
    if (n == 0)
        return hipSuccess;
    if (ntiles > 0x7fffffffull / kRadix)
        return hipErrorInvalidValue;
    char *t = static_cast<char *>(tmp);
    int *counts = reinterpret_cast<int *>(t + off_counts);
    K *kalt = reinterpret_cast<K *>(t + off_keys);
    V *valt = reinterpret_cast<V *>(t + off_vals);
    const unsigned width = end_bit > begin_bit ? end_bit - begin_bit : 0;
    const int passes = (int)((width + kRadixBits - 1) / kRadixBits);
    if (passes == 0) {
        hipLaunchKernelGGL(copy_array<K>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, kin, kout, n);
        hipLaunchKernelGGL(copy_array<V>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, vin, vout, n);
        return hipGetLastError();
    }
    const K *ksrc = kin;
    const V *vsrc = vin;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;  // the last pass writes the output, the ones before alternate
        K *kdst = to_out ? kout : kalt;
        V *vdst = to_out ? vout : valt;
        const unsigned shift = begin_bit + (unsigned)p * kRadixBits;
        const unsigned bits_here = width - (unsigned)p * kRadixBits < (unsigned)kRadixBits ? width - (unsigned)p * kRadixBits : (unsigned)kRadixBits;
        const unsigned mask = (1u << bits_here) - 1u;
        hipLaunchKernelGGL(sort_histogram<K>, dim3((unsigned)ntiles), dim3(kSortBlock), 0, st, ksrc, n, shift, mask, counts, (unsigned)ntiles);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
        size_t sb = scan_bytes;
        e = scan<false>(t + off_scan, sb, counts, counts, 0, counts_n, st);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL((sort_scatter<K, V>), dim3((unsigned)ntiles), dim3(kSortBlock), 0, st, ksrc, vsrc, kdst, vdst, n, shift, mask, counts,
                           (unsigned)ntiles);
        e = hipGetLastError();
        if (e != hipSuccess)
            return e;
        ksrc = kdst;
        vsrc = vdst;
    }
    return hipSuccess;
}

}  // namespace prim
}  // namespace smvp
