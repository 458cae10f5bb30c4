// smvp_kernels.hip -- the CDNA4 (gfx950) SpMV kernels.
//
// These replace the two timed loops of the reference:
//   CSR   main-cli.c:410-416    for row: for j: y[row] += val[j] * x[col_ind[j]]
//   TJDS  main-cli.c:1013-1020  for diag: for j: y[row_ind[j]] += val[j] * x_perm[...]
// Both are HBM-bound streaming + gather/scatter work (2 flop per 12 B), so there
// is no MFMA here: what matters is that val/col_ind/row_ind are read exactly
// once with wide coalesced loads, that x is gathered through L2 / Infinity Cache,
// and that y is written once.
//
// Compiled with -ffp-contract=off: a product is rounded before it is added,
// like the reference's x86-64 build, so every row that is summed left to right
// by one lane is bit-identical to the serial CPU result.
#include "smvp_kernels.h"

#include <cstdio>
#include <cstdlib>

namespace smvp {

// ---------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------
template <int WIDTH>
__device__ __forceinline__ double shfl_down_sum(double v)
{
#pragma unroll
    for (int off = WIDTH / 2; off > 0; off >>= 1)
        v += __shfl_down(v, off, WIDTH);
    return v;
}

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Tiles are
// handed out in groups: XCD i takes `group` consecutive tiles out of every run of
// 8*group, so an XCD's L2 sees neighbouring rows (and re-uses their x lines) while
// all eight XCDs stay within 8*group tiles of each other in memory.  Measured on
// MI355X (profiles/r01_tile_group_sweep.txt; % of HBM peak on memplus x944 / on the
// random model, L2->fabric read bytes per launch):
//   group 1     64.9 / 18.7   2.08 GB      group 128   64.4 / 21.8   1.68 GB
//   group 8     65.5 / 19.3   1.91 GB      group 512   62.2 / 21.7   1.61 GB
//   group 32    65.2 / 21.2   1.84 GB      group 2048  55.9 / 20.8   1.60 GB
// (one XCD per contiguous eighth of the matrix was 60.1 %).  64 keeps the speed of
// the small groups and most of the traffic saving of the large ones.
// Speed only: any placement gives the same result.
__device__ __forceinline__ int tile_of_block(int block, int group)
{
    const int xcd = block & 7, seq = block >> 3;
    return (seq / group) * (8 * group) + xcd * group + seq % group;
}

#ifndef SMVP_TU_ILP
// ---------------------------------------------------------------------------
// K1: CSR, one sub-wavefront of T lanes per row, __shfl_down sums.
// T = 64 is the classic wavefront-per-row kernel; smaller T packs 64/T rows
// into a wave for short rows.  Lane l of a row reads entries a+l, a+l+T, ...
// so a wave's loads are contiguous runs of col_ind / val.
// ---------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(kVectorBlock) void csr_vector_rows(
    const int *__restrict__ row_ptr, const int *__restrict__ col_ind, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, int rows)
{
    const long long gid = (long long)blockIdx.x * kVectorBlock + threadIdx.x;
    const long long row = gid / T;
    const int lane = threadIdx.x & (T - 1);
    int a = 0, z = 0;
    if (row < rows) {
        a = row_ptr[row];
        z = row_ptr[row + 1];
    }
    double acc = 0.0;
    for (int j = a + lane; j < z; j += T)
        acc += val[j] * x[col_ind[j]];
    acc = shfl_down_sum<T>(acc);
    if (lane == 0 && row < rows)
        y[row] = acc;
}

// ---------------------------------------------------------------------------
// K2: CSR, fixed-nnz tiles with an LDS-staged segmented reduction.
//
// Tile b owns entries [b*TILE, (b+1)*TILE): every block streams the same number
// of bytes whatever the row lengths are (memplus: 86 % of rows <= 8 entries,
// 28 % of entries in rows > 64).  Phase 1 loads col_ind / val with 16-byte
// per-lane loads, gathers x and parks the products in LDS.  Phase 2 walks the
// rows that START inside the tile (tile_row[b] .. tile_row[b+1]): one lane per
// short row sums its segment left to right out of LDS; rows longer than
// kLongRow are queued and summed by a whole wavefront with __shfl_down.
// Entries in front of the first owned row belong to a row that started in an
// earlier tile: their sum goes to carry[b] and csr_stream_carry_fixup adds the
// carries to y in tile order, so the result does not depend on timing and y
// needs no zeroing.
// ---------------------------------------------------------------------------
template <int VPT>
__global__ __launch_bounds__(kStreamBlock) void csr_stream_tiles(
    const int *__restrict__ row_ptr, const int *__restrict__ col_ind, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ tile_row,
    double *__restrict__ carry, int rows, int nnz, int ntiles, int tile_group)
{
    constexpr int TILE = kStreamBlock * VPT;
    constexpr int QCAP = TILE / kLongRow + 1;
    __shared__ double prod[TILE];
    __shared__ int long_rows[QCAP];
    __shared__ int long_count;

    const int b = tile_of_block(blockIdx.x, tile_group);
    if (b >= ntiles)
        return;
    const int t = threadIdx.x;
    const long long s = (long long)b * TILE;
    const int e = (int)(s + TILE < (long long)nnz ? s + TILE : (long long)nnz);
    if (t == 0)
        long_count = 0;
    const int rlo = tile_row[b];
    const int rhi = tile_row[b + 1];

    // ---- phase 1: stream + gather + multiply.  The full-tile path is straight-line code so that
    // all VPT gathers of a lane are in flight together; a guarded version that branched per entry
    // ran 12-25 % slower (MI355X, memplus x944).
    const long long j0 = s + (long long)t * VPT;
    double p[VPT];
    if (j0 + VPT <= (long long)nnz) {
        int c[VPT];
        double v[VPT];
#pragma unroll
        for (int k = 0; k < VPT; k += 4)
            *reinterpret_cast<int4 *>(&c[k]) = *reinterpret_cast<const int4 *>(col_ind + j0 + k);
#pragma unroll
        for (int k = 0; k < VPT; k += 2)
            *reinterpret_cast<double2 *>(&v[k]) = *reinterpret_cast<const double2 *>(val + j0 + k);
#pragma unroll
        for (int k = 0; k < VPT; ++k)
            p[k] = v[k] * x[c[k]];
    } else {
#pragma unroll
        for (int k = 0; k < VPT; ++k)
            p[k] = (j0 + k < (long long)nnz) ? val[j0 + k] * x[col_ind[j0 + k]] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < VPT; k += 2)
        *reinterpret_cast<double2 *>(&prod[t * VPT + k]) = make_double2(p[k], p[k + 1]);
    __syncthreads();

    // ---- phase 2a: the head of the tile continues an earlier row
    const int lo = (int)s;  // nnz < 2^31
    if (t < 64) {
        const int first = rlo < rows ? row_ptr[rlo] : nnz;
        const int head = (first < e ? first : e) - lo;
        double acc = 0.0;
        for (int i = t; i < head; i += 64)
            acc += prod[i];
        acc = shfl_down_sum<64>(acc);
        if (t == 0)
            carry[b] = acc;
    }

    // ---- phase 2b: one lane per owned row, long rows deferred
    for (int r = rlo + t; r < rhi; r += kStreamBlock) {
        const int a = row_ptr[r];
        const int nxt = row_ptr[r + 1];
        const int z = nxt < e ? nxt : e;
        if (z - a <= kLongRow) {
            double acc = 0.0;
            for (int i = a - lo; i < z - lo; ++i)
                acc += prod[i];
            y[r] = acc;
        } else {
            long_rows[atomicAdd(&long_count, 1)] = r;
        }
    }
    __syncthreads();

    // ---- phase 2c: one wavefront per long row
    const int nlong = long_count;
    const int lane = t & 63;
    for (int q = t >> 6; q < nlong; q += kStreamBlock / 64) {
        const int r = long_rows[q];
        const int a = row_ptr[r];
        const int nxt = row_ptr[r + 1];
        const int z = nxt < e ? nxt : e;
        double acc = 0.0;
        for (int i = a - lo + lane; i < z - lo; i += 64)
            acc += prod[i];
        acc = shfl_down_sum<64>(acc);
        if (lane == 0)
            y[r] = acc;
    }
}

#endif  // !SMVP_TU_ILP
// ---------------------------------------------------------------------------
// K2': the same tiles, but the tile in which a row STARTS finishes that row
// itself: it also loads the few entries past its end that belong to its last
// row (tile_next[b] = row_ptr[first row of the next tile] says how many) into an
// LDS overflow area.  No carries, no second launch, and every row of up to
// kLongRow entries is summed left to right by one lane -- bit-identical to the
// serial loop -- wherever it lies.  A tile in which no row starts exits at once.
// A last row that runs more than kStreamOver entries past the tile is finished
// by the whole workgroup straight from global memory; matrices with rows far
// longer than that are planned onto the carry kernel above instead, which
// spreads such a row over many workgroups.
//
// FLAVOR selects what an "entry" is (owner_entry below):
//   kFlavorCsr     CSR: value val[j], operand x[col_ind[j]]                       (main-cli.c:410-416)
//   kFlavorCsr16   the same product with 10 instead of 12 bytes per entry: for the tiles whose columns span less than 65536
//                  (banded and block-structured matrices: all of them) the plan keeps col_ind a second time as 16-bit
//                  offsets from the tile's smallest column; the tile adds its base (one scalar) back.  Wider tiles
//                  (col_base < 0), overflow entries and the slow paths read col_ind itself.
//   kFlavorTjdsK   TJDS by rows: the stream lists, row by row, the TJDS positions p of the row's entries (`pos`) and
//                  their permuted columns k (`col_ind`): value val[p] gathered from the jagged-diagonal array,
//                  operand x_perm[k] (the one-kernel TJDS product, see ensure_row_gather in the engine)
//   kFlavorTjdsS   the same rows, but inside every tile the entries are listed in TJDS order (ascending position):
//                  the tile walks its piece of the jagged diagonals the way the format stores them -- neighbouring
//                  lanes read neighbouring val / x_perm entries, lane t takes entries t, t + 256, ... of the tile --
//                  and each product goes to the LDS slot of its row-major place (`col_ind` holds slot | diagonal << 11);
//                  the entries a tile needs from beyond its end are kept a second time in row order (ovf_*)
//   kFlavorTjdsH   kFlavorTjdsS with 4 bytes of index per entry instead of 8: two 16-bit halves of one word.  One is the LDS slot | the
//                  number, modulo 32, of the entry's RUN: a stretch of the tile's sorted entries inside one jagged diagonal
//                  and one aligned block of 2^16 positions (the tile's cached entries, which need only their column, are
//                  sorted by column and form runs by blocks of 2^16 columns).  The other is the low half of the position
//                  (cached: of the column).  The tile's run table holds {base, sub} per run: position = base + low half,
//                  operand index = position - sub (sub = start_pos of the run's diagonal; cached: 0).  An entry finds its
//                  run from the run of its group of 32 entries (`group_run`, one 16-bit word per group) and the 5-bit hint;
//                  the table sits in a wavefront's registers (a handful of runs per tile on banded matrices), so the two
//                  gathers follow the streams after two cross-lane reads.
// ---------------------------------------------------------------------------
typedef int int4v __attribute__((ext_vector_type(4)));               // clang vectors: what the non-temporal builtins take

// The streams every flavour reads are plain __restrict__ kernel parameters (the compiler then keeps the uniform plan
// reads on the scalar unit and is free to order the loads); what only some flavours need travels in this struct.
struct OwnerExtra {
    const int *pos = nullptr;                 // Tjds*: TJDS position of each stream entry
    const int *start_pos = nullptr;           // TjdsS
    const int *ovf_ptr = nullptr;             // TjdsS: ntiles + 1 bounds of the tiles' overflow entries in ovf_val / ovf_k
    const double *ovf_val = nullptr;          // TjdsS: value ...
    const int *ovf_k = nullptr;               // TjdsS: ... and permuted column of the entries [e, tile_next) of each tile, row order
    const int *cache_ptr = nullptr;           // TjdsS: ntiles + 1 bounds of the tiles' runs in val_cache (a tile's last entries)
    const double *val_cache = nullptr;        // TjdsS: values of the entries whose val lines scatter over many tiles, tile by tile
    const unsigned short *col16 = nullptr;    // Csr16: column - col_base[tile] per entry
    const int *col_base = nullptr;            // Csr16: smallest column of every tile
    const unsigned *word32 = nullptr;         // TjdsH, per entry: low 16 bits of its position (cached entries: of its permuted column) | (slot | run hint << kSlotBits) << 16
    const unsigned short *group_run = nullptr;  // TjdsH: per tile and group of 32 entries, the run (inside the tile) of its first entry
    const int *run_ptr = nullptr;             // TjdsH: ntiles + 1 bounds of the tiles' runs in run_tab
    const int *run_tab = nullptr;             // TjdsH: per run {base of its block of 2^16 positions, start_pos of its diagonal} (cached: {column block, 0})
    int unit_x = 0;                     // TjdsH: the operand is the unit vector -- no x gather; overflow entries are POSITIONS in ovf_k (their
                                    // values are read from val like everybody's: val changes from launch to launch, see the engine's TjdsSource)
    const unsigned short *row_rel = nullptr;  // every row's first entry relative to the first entry of the tile it
                                    // starts in (2 B per row read by the product instead of row_ptr's 4); nullptr: row_ptr itself
    unsigned long long *stamps = nullptr;     // STAMPED: per-wave {first, last} wall-clock ticks of this launch
};

// The ONE place an OwnerExtra is filled from a launch description (the plain launcher, the ILP unit's launcher through it, the
// repeating launcher): a field added to either struct is carried -- or left at its null default -- here and nowhere else.
[[maybe_unused]] static inline OwnerExtra owner_extra_of(const OwnerLaunch &l, unsigned long long *stamps)
{
    OwnerExtra ex{};
    ex.pos = l.pos, ex.start_pos = l.start_pos, ex.ovf_ptr = l.ovf_ptr, ex.ovf_val = l.ovf_val, ex.ovf_k = l.ovf_k;
    ex.cache_ptr = l.cache_ptr, ex.val_cache = l.val_cache;
    ex.col16 = l.col16, ex.col_base = l.col_base;
    ex.unit_x = l.unit_x, ex.word32 = l.word32, ex.group_run = l.group_run, ex.run_ptr = l.run_ptr, ex.run_tab = l.run_tab;
    ex.row_rel = l.row_rel;
    ex.stamps = stamps;
    return ex;
}

// what the helpers below need of the kernel's arguments
struct OwnerArgs {
    const int *__restrict__ col_ind;
    const double *__restrict__ val;
    const double *__restrict__ x;
    const int *__restrict__ pos;
    const int *__restrict__ start_pos;
    const double *__restrict__ ovf_val;
    const int *__restrict__ ovf_k;
    int unit_x;   // TjdsH: overflow entries are positions, the operand is 1 (OwnerExtra::unit_x)
};

// one entry's product the slow way (tile tails, overflow beyond one block width, giant rows)
template <int FLAVOR>
__device__ __forceinline__ double owner_product_slow(const OwnerArgs &a, long long j)
{
    if constexpr (FLAVOR == kFlavorCsr || FLAVOR == kFlavorCsr16)
        return a.val[j] * a.x[a.col_ind[j]];
    else if constexpr (FLAVOR == kFlavorTjdsK)
        return a.val[a.pos[j]] * a.x[a.col_ind[j]];
    else {
        const int p = a.pos[j];
        return a.val[p] * a.x[p - a.start_pos[(unsigned)a.col_ind[j] >> kSlotBits]];
    }
}

// product of overflow entry `i` of tile b (stream entry e + i, the i-th entry past the tile's end)
template <int FLAVOR>
__device__ __forceinline__ double owner_overflow_product(const OwnerArgs &a, int ovf_base, int e, int i)
{
    if constexpr (FLAVOR == kFlavorTjdsS || FLAVOR == kFlavorTjdsH)
        return a.unit_x ? a.val[a.ovf_k[ovf_base + i]] : a.ovf_val[ovf_base + i] * a.x[a.ovf_k[ovf_base + i]];
    else
        return owner_product_slow<FLAVOR>(a, (long long)e + i);
}

// Diagnostic builds only (make HIPFLAGS+=-DSMVP_PHASE_STAMPS): where a workgroup of the owner kernel spends its time.
// Thread 0 of every workgroup adds the 100 MHz wall-clock ticks between its phase boundaries to six global counters;
// smvp_debug_phase_stamps() (engine) prints and clears them.  The normal build contains none of this.
#if defined(SMVP_PHASE_STAMPS) && defined(SMVP_TU_ILP)
#undef SMVP_PHASE_STAMPS   // (the diagnostic build runs every flavour in the other unit ...
#define SMVP_ILP_UNIT_UNUSED  // ... and this unit instantiates no owner kernel: one device body per kernel name in the library)
#endif
#ifdef SMVP_PHASE_STAMPS
__device__ unsigned long long g_owner_phase[8];
#define SMVP_PHASE(n)                                                                        \
    do {                                                                                     \
        if (threadIdx.x == 0 && (blockIdx.x & 127) == 5) {                                   \
            const unsigned long long now_ = wall_clock64();                                  \
            if ((n) > 0)                                                                     \
                atomicAdd(&g_owner_phase[(n)-1], now_ - phase_prev_);                        \
            else                                                                             \
                atomicAdd(&g_owner_phase[7], 1ull);                                          \
            phase_prev_ = now_;                                                              \
        }                                                                                    \
    } while (0)
#else
#define SMVP_PHASE(n) do { } while (0)
#endif

// Diagnostic builds only (make ... HIPFLAGS+=-DSMVP_TJDS_NEUTRALISE=mask; tools/tjds_traffic_by_stream.sh): the tile-ordered TJDS
// product with one of its streams taken out -- WRONG results, the PMC counters then show what that stream moves.
// 1: the x_perm gathers (operand 1.0)   2: the in-place val gathers (val[position])   4: the value cache's reads
// 8: the overflow entries (their index, value and operand loads).  The normal build contains none of this.
#ifndef SMVP_TJDS_NEUTRALISE
#define SMVP_TJDS_NEUTRALISE 0
#endif

// The body of the owner kernel for workgroup number `block` of the launch's grid (the plain kernel passes blockIdx.x; the
// repeating kernel below walks the same grid several times).  Every thread of the workgroup leaves through the same exit.
template <int VPT, int FLAVOR, bool STAMPED>
__device__ __forceinline__ void owner_body(
    const int *__restrict__ row_ptr, const int *__restrict__ col_ind, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ tile_row,
    const int *__restrict__ tile_next, int rows, int nnz_arg, int ntiles, int tile_group_arg, const OwnerExtra &ex, const int block)
{
    const OwnerArgs a = {col_ind, val, x, ex.pos, ex.start_pos, ex.ovf_val, ex.ovf_k, FLAVOR == kFlavorTjdsH ? ex.unit_x : 0};
    constexpr int TILE = kStreamBlock * VPT;
    constexpr int QCAP = (TILE + kStreamOver) / kLongRow + 1;
    constexpr bool CSR = FLAVOR == kFlavorCsr || FLAVOR == kFlavorCsr16;
    constexpr bool COL16 = FLAVOR == kFlavorCsr16;
    constexpr bool HALF = FLAVOR == kFlavorTjdsH;
    constexpr bool TJDS = FLAVOR == kFlavorTjdsK || FLAVOR == kFlavorTjdsS || HALF;
    constexpr bool SORTED = FLAVOR == kFlavorTjdsS || HALF;  // entries in TJDS order inside the tile, lane-strided
    static_assert(TILE <= (1 << kSlotBits), "slot bits");
    __shared__ double prod[TILE + kStreamOver];
    __shared__ int long_rows[QCAP];  // rows longer than kLongRow and their LDS segments, filled in phase 2b
    __shared__ int long_a[QCAP];
    __shared__ int long_z[QCAP];
    __shared__ int long_count;
    __shared__ double wave_sum[kStreamBlock / 64];

    const int t = threadIdx.x;
#ifdef SMVP_PHASE_STAMPS
    unsigned long long phase_prev_ = 0;
#endif
    // optional device-side timing: every wave notes when it started and (after its last store has been
    // acknowledged) when it finished; max(last) - min(first) over the launch is the product's own duration,
    // free of launch and event overhead (the engine reduces the slots, see stamp_reduce)
    unsigned long long *stamp = nullptr;
    if constexpr (STAMPED) {
        stamp = ex.stamps + 2 * ((size_t)block * (kStreamBlock / 64) + (t >> 6));
        if ((t & 63) == 0)
            stamp[0] = wall_clock64();
    }
#define SMVP_OWNER_EXIT()                                         \
    do {                                                          \
        if constexpr (STAMPED) {                                  \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
            if ((t & 63) == 0)                                    \
                stamp[1] = wall_clock64();                        \
        }                                                         \
        return;                                                   \
    } while (0)

    const int b = tile_of_block(block, tile_group_arg);
    if (b >= ntiles)
        SMVP_OWNER_EXIT();
    const int nnz = nnz_arg;
    const long long s = (long long)b * TILE;
    const long long j0 = s + (long long)t * VPT;

    // ---- phase 1: stream + gather + multiply.  Straight-line code, ordered so that nothing waits on more
    // than one dependent round trip: the tile's own index / value loads go out first (they depend on nothing
    // but the block index), then the plan words, then row_ptr for phase 2 and the overflow entries, then ALL
    // gathers (index -> operand is the one dependence that cannot be avoided).
    const bool whole = j0 + VPT <= (long long)nnz;  // this lane's entries all exist (always, except in the last tile)
    int c[VPT];     // operand index
    int pj[VPT];    // Tjds*: position in val
    double v[VPT];
    const bool full_tile = s + TILE <= (long long)nnz;
    int grp[HALF ? VPT : 1];  // TjdsH: run (inside the tile) of the first entry of this entry's group of 32
    // TjdsH: the tile's run table ({base, sub} per run), one run per lane, requested with the tile's own streams: an entry
    // then takes its run's words from its wavefront's copy by cross-lane reads instead of a second, dependent trip to memory
    // in front of the gathers (tiles with more than 64 runs read the rest from memory)
    int run0 = 0;
    int2 run_tbl = make_int2(0, 0);
    if constexpr (HALF) {
        if (full_tile) {
            run0 = ex.run_ptr[b];
            const int nruns = ex.run_ptr[b + 1] - run0;
            const int ln = t & 63;
            if (ln < nruns)
                run_tbl = reinterpret_cast<const int2 *>(ex.run_tab)[run0 + ln];
        }
    }
    if constexpr (HALF) {
        if (full_tile) {
#pragma unroll
            for (int k = 0; k < VPT; ++k) {
                if constexpr (VPT == 1) {  // 256-entry tiles: a matrix that lives in the caches and is multiplied again and again
                    pj[k] = (int)ex.word32[s + k * kStreamBlock + t];
                } else {
                    pj[k] = (int)__builtin_nontemporal_load(ex.word32 + s + k * kStreamBlock + t);  // low half of the position | (slot | run hint) << 16
                }
                // the two group words of this wavefront's 64 entries are one aligned 32-bit word at a wave-uniform address: a
                // scalar load instead of a vector one per lane
                const unsigned gw = reinterpret_cast<const unsigned *>(ex.group_run)[((size_t)b * (TILE / 32) + k * (kStreamBlock / 32) +
                                                                                     2 * __builtin_amdgcn_readfirstlane(t >> 6)) >> 1];
                grp[k] = (t & 32) ? (int)(gw >> 16) : (int)(gw & 0xffffu);
            }
        }
    } else if constexpr (SORTED) {
        if (full_tile) {
#pragma unroll
            for (int k = 0; k < VPT; ++k) {
                if constexpr (VPT > 1) {  // read once: non-temporal loads leave the L2 to the val / x_perm lines neighbouring tiles share
                    pj[k] = __builtin_nontemporal_load(a.pos + s + k * kStreamBlock + t);
                    c[k] = __builtin_nontemporal_load(a.col_ind + s + k * kStreamBlock + t);
                } else {
                    pj[k] = a.pos[s + k * kStreamBlock + t];
                    c[k] = a.col_ind[s + k * kStreamBlock + t];  // slot | diagonal << kSlotBits
                }
            }
        }
    } else if (whole) {
        if constexpr (VPT >= 4) {
            if constexpr (COL16) {
                const int base = ex.col_base[b];
                if (base >= 0) {
#pragma unroll
                    for (int k = 0; k < VPT; k += 4) {  // four 16-bit offsets per 8-byte load
                        const uint2 w = *reinterpret_cast<const uint2 *>(ex.col16 + j0 + k);
                        c[k] = base + (int)(w.x & 0xffffu), c[k + 1] = base + (int)(w.x >> 16);
                        c[k + 2] = base + (int)(w.y & 0xffffu), c[k + 3] = base + (int)(w.y >> 16);
                    }
                } else {  // a tile whose columns span 65536 or more
#pragma unroll
                    for (int k = 0; k < VPT; k += 4)
                        *reinterpret_cast<int4 *>(&c[k]) = *reinterpret_cast<const int4 *>(a.col_ind + j0 + k);
                }
            } else {
#pragma unroll
                for (int k = 0; k < VPT; k += 4)
                    *reinterpret_cast<int4 *>(&c[k]) = *reinterpret_cast<const int4 *>(a.col_ind + j0 + k);
            }
            if constexpr (TJDS) {  // (read once: non-temporal)
#pragma unroll
                for (int k = 0; k < VPT; k += 4)
                    *reinterpret_cast<int4v *>(&pj[k]) = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(a.pos + j0 + k));
            }
            if constexpr (CSR) {
#pragma unroll
                for (int k = 0; k < VPT; k += 2)
                    *reinterpret_cast<double2 *>(&v[k]) = *reinterpret_cast<const double2 *>(a.val + j0 + k);
            }
        } else {  // VPT == 1: 256-entry tiles for matrices too small to fill the chip with 1024-entry ones
            c[0] = a.col_ind[j0];
            if constexpr (TJDS)
                pj[0] = a.pos[j0];
            if constexpr (CSR)
                v[0] = a.val[j0];
        }
    }
    const int rlo = tile_row[b];
    const int rhi = tile_row[b + 1];
    if (rlo == rhi)
        SMVP_OWNER_EXIT();  // all of this tile continues a row owned by an earlier tile
    SMVP_PHASE(0);
    const int lo = (int)s;  // nnz < 2^31
    const int e = (int)(s + TILE < (long long)nnz ? s + TILE : (long long)nnz);
    const int zend = tile_next[b];  // row_ptr[rhi]: end of the last owned row, >= e
    const int ext = zend - e;
    // bounds of owned row r as offsets from the tile's first entry: from the plan's 16-bit offsets where it keeps them (the
    // last owned row ends at zend), else from row_ptr
    const unsigned short *__restrict__ row_rel = ex.row_rel;
    auto row_bounds = [&](int r, int &ra, int &rz) {
        if (row_rel) {
            ra = row_rel[r];
            rz = r + 1 < rhi ? (int)row_rel[r + 1] : zend - lo;
        } else {
            ra = row_ptr[r] - lo;
            rz = row_ptr[r + 1] - lo;
        }
    };
    const bool giant = ext > kStreamOver;
    if (t == 0)
        long_count = 0;
    const bool over0 = !giant && t < ext;  // this lane fetches overflow entry e + t
    double p[VPT];
    double po = 0.0;
    int rp_a = 0, rp_b = 0, rp_a2 = 0, rp_b2 = 0;  // bounds of this lane's first two rows of phase 2 (rlo + t, rlo + t + 256), from the tile's first entry
    int ovf_base = 0, cache0 = 0, in_place = 0;
    if constexpr (SORTED) {
        ovf_base = ex.ovf_ptr[b];
        cache0 = ex.cache_ptr[b];
        in_place = (e - lo) - (ex.cache_ptr[b + 1] - cache0);  // entries of this tile whose value is read from val itself
    }
    if constexpr (SORTED) {
        if (full_tile && rlo + t < rhi)
            row_bounds(rlo + t, rp_a, rp_b);
        if (full_tile && rlo + t + kStreamBlock < rhi)
            row_bounds(rlo + t + kStreamBlock, rp_a2, rp_b2);
        int co = 0;
        double vo = 0.0;
        const bool unit_x = HALF && ex.unit_x;
        if (over0 && !(SMVP_TJDS_NEUTRALISE & 8)) {
            co = a.ovf_k[ovf_base + t];
            if (!unit_x)
                vo = a.ovf_val[ovf_base + t];
        }
        if (full_tile) {
            int slot[VPT];
#pragma unroll
            for (int k = 0; k < VPT; ++k) {
                if constexpr (HALF)
                    c[k] = (int)((unsigned)pj[k] >> 16), pj[k] &= 0xffff;  // the entry's word: low half of the position | (slot | run hint) << 16
                slot[k] = c[k] & ((1 << kSlotBits) - 1);
                if constexpr (HALF) {
                    const int r = grp[k] + (((c[k] >> kSlotBits) - grp[k]) & 31);
                    int base = __shfl(run_tbl.x, r & 63), sub = __shfl(run_tbl.y, r & 63);
                    if (r >= 64) {
                        const int2 w = reinterpret_cast<const int2 *>(ex.run_tab)[run0 + r];
                        base = w.x, sub = w.y;
                    }
                    pj[k] += base;
                    c[k] = pj[k] - sub;
                } else {
                    c[k] = pj[k] - a.start_pos[(unsigned)c[k] >> kSlotBits];
                }
            }
            double xk[VPT];
            // the tile's last `cached` entries take their value from the tile's own run of the cache (coalesced) instead
            // of val[position]: their val lines are shared with many other tiles (see mark_scattered_lines)
#pragma unroll
            for (int k = 0; k < VPT; ++k) {
                const int idx = k * kStreamBlock + t;
                const double *src = idx < in_place ? a.val + pj[k] : ex.val_cache + (cache0 + (idx - in_place));
                if ((SMVP_TJDS_NEUTRALISE & 2) && idx < in_place)
                    v[k] = 1.0;
                else if ((SMVP_TJDS_NEUTRALISE & 4) && idx >= in_place)
                    v[k] = 1.0;
                else
                    v[k] = *src;
            }
            double xo = 0.0;
            if (unit_x) {  // second phase of the two-phase product: val holds products, the overflow entries' too (by position)
#pragma unroll
                for (int k = 0; k < VPT; ++k)
                    xk[k] = 1.0;
                if (over0)
                    vo = a.val[co], xo = 1.0;
            } else {
#pragma unroll
                for (int k = 0; k < VPT; ++k)
                    xk[k] = (SMVP_TJDS_NEUTRALISE & 1) ? 1.0 : a.x[c[k]];
                xo = over0 && !(SMVP_TJDS_NEUTRALISE & 8) ? a.x[co] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < VPT; ++k)
                prod[slot[k]] = v[k] * xk[k];
            po = vo * xo;
        } else {  // the last, partial tile
            for (int k = 0; k < VPT; ++k) {
                const int idx = k * kStreamBlock + t;
                if (s + idx < (long long)nnz) {
                    if constexpr (HALF) {
                        const unsigned w32 = ex.word32[s + idx];
                        const int m = (int)(w32 >> 16), g = ex.group_run[(size_t)b * (TILE / 32) + (idx >> 5)];
                        const int r = g + (((m >> kSlotBits) - g) & 31);
                        const int2 w = reinterpret_cast<const int2 *>(ex.run_tab)[ex.run_ptr[b] + r];
                        const int pw = w.x + (int)(w32 & 0xffffu);
                        const double vv = idx < in_place ? a.val[pw] : ex.val_cache[cache0 + (idx - in_place)];
                        prod[m & ((1 << kSlotBits) - 1)] = unit_x ? vv : vv * a.x[pw - w.y];
                    } else {
                        const int pw = a.pos[s + idx];
                        const double vv = idx < in_place ? a.val[pw] : ex.val_cache[cache0 + (idx - in_place)];
                        const int m = a.col_ind[s + idx];
                        prod[m & ((1 << kSlotBits) - 1)] = vv * a.x[pw - a.start_pos[(unsigned)m >> kSlotBits]];
                    }
                }
            }
            if (over0)
                po = unit_x ? a.val[co] : vo * a.x[co];
        }
    } else if (whole) {
        if (rlo + t < rhi)  // this lane's first row in phase 2
            row_bounds(rlo + t, rp_a, rp_b);
        if (rlo + t + kStreamBlock < rhi)  // ... and its second: with short rows a tile holds more rows than lanes, and a row_ptr read
            row_bounds(rlo + t + kStreamBlock, rp_a2, rp_b2);  // behind the barrier is a whole memory round trip in phase 2 (in-kernel
                                                               // stamps on the near part of the random model, 4.3 entries per row:
                                                               // phase 2 took 4.1 of the workgroup's 9.9 us)
        int co = 0, pjo = 0;
        double vo = 0.0;
        if (over0) {
            co = a.col_ind[e + t];
            if constexpr (TJDS)
                pjo = a.pos[e + t];
            if constexpr (CSR)
                vo = a.val[e + t];
        }
        double xk[VPT];
        if constexpr (TJDS) {
#pragma unroll
            for (int k = 0; k < VPT; ++k)
                v[k] = a.val[pj[k]];
            if (over0)
                vo = a.val[pjo];
        }
#pragma unroll
        for (int k = 0; k < VPT; ++k)
            xk[k] = a.x[c[k]];
        const double xo = over0 ? a.x[co] : 0.0;
#pragma unroll
        for (int k = 0; k < VPT; ++k)
            p[k] = v[k] * xk[k];
        po = vo * xo;
    } else {
#pragma unroll
        for (int k = 0; k < VPT; ++k)
            p[k] = (j0 + k < (long long)nnz) ? owner_product_slow<FLAVOR>(a, j0 + k) : 0.0;
        if (over0)
            po = owner_product_slow<FLAVOR>(a, e + t);
    }
    if constexpr (!SORTED) {
        if constexpr (VPT >= 2) {
#pragma unroll
            for (int k = 0; k < VPT; k += 2)
                *reinterpret_cast<double2 *>(&prod[t * VPT + k]) = make_double2(p[k], p[k + 1]);
        } else {
            prod[t] = p[0];
        }
    }
    if (over0)
        prod[e - lo + t] = po;
    if (!giant)  // rare: the last row runs more than one block width past the tile
        for (int i = t + kStreamBlock; i < ext; i += kStreamBlock)
            prod[e - lo + i] = owner_overflow_product<FLAVOR>(a, ovf_base, e, i);
    SMVP_PHASE(1);
    __syncthreads();
    SMVP_PHASE(2);

    // ---- phase 2b: one lane per owned row; its bounds were fetched with the tile (no global read after
    // the barrier for the first kStreamBlock rows); long rows are queued in LDS with their bounds
    // Each lane's first two rows (their bounds came with the tile) are finished without touching memory in between: a
    // row_ptr read in the same loop as the y stores made the compiler wait for vmcnt(0) every round, i.e. for the
    // previous round's STORES to be acknowledged (CDNA4 counts stores in vmcnt) -- 3 of a workgroup's 10 us on short
    // rows (in-kernel stamps).  Only a tile with more than 512 rows goes on to the loop that reads row_ptr.
    const int last = rhi - 1;
    auto finish_row = [&](int r, int ra, int rz) {
        if (rz - ra <= kLongRow) {
            // left to right, like the serial loop; the LDS reads go out four at a time, the adds stay in order
            double acc = 0.0;
            for (int i = ra; i < rz; i += 4) {
                const double v0 = prod[i], v1 = prod[i + 1 < rz ? i + 1 : i], v2 = prod[i + 2 < rz ? i + 2 : i],
                             v3 = prod[i + 3 < rz ? i + 3 : i];
                acc += v0;
                if (i + 1 < rz)
                    acc += v1;
                if (i + 2 < rz)
                    acc += v2;
                if (i + 3 < rz)
                    acc += v3;
            }
            // (256-entry tiles are what matrices that live in the caches get: there a plain store is acknowledged by the L2,
            // sooner than a non-temporal one by memory -- and the product's window ends with that acknowledgement)
            if constexpr (VPT == 1)
                y[r] = acc;
            else
                __builtin_nontemporal_store(acc, &y[r]);
        } else {
            const int q = atomicAdd(&long_count, 1);
            long_rows[q] = r;
            long_a[q] = ra;
            long_z[q] = rz;
        }
    };
    int r_next = rlo + t;
    if (full_tile) {
        if (r_next < rhi && !(giant && r_next == last))
            finish_row(r_next, rp_a, rp_b);
        r_next += kStreamBlock;
        if (r_next < rhi && !(giant && r_next == last))
            finish_row(r_next, rp_a2, rp_b2);
        r_next += kStreamBlock;
    }
    for (int r = r_next; r < rhi; r += kStreamBlock) {
        if (giant && r == last)
            continue;
        int ra, rz;
        row_bounds(r, ra, rz);
        finish_row(r, ra, rz);
    }
    SMVP_PHASE(3);
    __syncthreads();
    SMVP_PHASE(4);

    // ---- phase 2c: one wavefront per long row (measured: handing a long row to the finder's own wavefront
    // by ballot, without this queue and barrier, was 0-2 % slower -- the long rows of a tile then share one wave)
    const int nlong = long_count;
    const int lane = t & 63;
    for (int q = t >> 6; q < nlong; q += kStreamBlock / 64) {
        const int zq = long_z[q];
        double acc = 0.0;
        for (int i = long_a[q] + lane; i < zq; i += 64)
            acc += prod[i];
        acc = wave_sum_dpp(acc);
        if (lane == 0)
            y[long_rows[q]] = acc;
    }
    SMVP_PHASE(5);

    // ---- phase 2d: a last row that runs far past the tile: LDS part + the rest from global memory
    if (giant) {
        const int ra = row_ptr[last];
        double acc = 0.0;
        for (int i = ra - lo + t; i < e - lo; i += kStreamBlock)
            acc += prod[i];
        for (int j = e + t; j < zend; j += 4 * kStreamBlock) {
            double pg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j + u * kStreamBlock;
                pg[u] = jj < zend ? owner_overflow_product<FLAVOR>(a, ovf_base, e, jj - e) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc += pg[u];
        }
        acc = shfl_down_sum<64>(acc);
        if (lane == 0)
            wave_sum[t >> 6] = acc;
        __syncthreads();
        if (t == 0) {
            double total = 0.0;
#pragma unroll
            for (int w = 0; w < kStreamBlock / 64; ++w)
                total += wave_sum[w];
            y[last] = total;
        }
    }
    SMVP_OWNER_EXIT();
#undef SMVP_OWNER_EXIT
}

template <int VPT, int FLAVOR, bool STAMPED>
__global__ __launch_bounds__(kStreamBlock) void csr_stream_owner(
    const int *__restrict__ row_ptr, const int *__restrict__ col_ind, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ tile_row,
    const int *__restrict__ tile_next, int rows, int nnz_arg, int ntiles, int tile_group_arg, const OwnerExtra ex)
{
    owner_body<VPT, FLAVOR, STAMPED>(row_ptr, col_ind, val, x, y, tile_row, tile_next, rows, nnz_arg, ntiles, tile_group_arg, ex,
                                     (int)blockIdx.x);
}

#ifndef SMVP_TU_ILP
// ---------------------------------------------------------------------------
// K2-repeat: `reps` products in ONE launch, each with a window of its own -- for the reference's own use case, -n 1000 on a
// matrix that lives in the caches (main-cli.c:402-420: the same product, the same x, n times).  A launch boundary costs such
// a product as much as the product (memplus.mtx: 3.0 us in the kernel, 4.9 us from dispatch to dispatch in a hipGraph, 6.4 us of
// loop wall per product: profiles/r04_cli_n1000.txt).  Here the workgroups stay: every one walks its share of the launch's
// grid once per product, and between two products all of them meet at a barrier that orders TIME only -- the products
// are independent (x is never changed, y is overwritten with the same values), so nothing has to become visible to anybody
// and the barrier needs no release / acquire fences (which are what a grid barrier mostly costs: MI355X_MICROARCH price
// list, barrier-xcd): arrivals are counted on about sqrt(workgroups) sharded counters (device-scope atomics run at the memory
// side, ~12 ns each on one address: different addresses take them in parallel), the last arrival of a shard counts on a top counter,
// everybody polls that one with sc1 loads.  Every wave stamps the wall clock when it passes the barrier into product i and
// when its last store of product i has been acknowledged: the same {first, last} slots the stamped single launches write,
// reduced by stamp_reduce -- windows that cannot overlap, one per product, inside one launch.
//
// The grid must be resident as a whole (the launcher sizes it from the occupancy query, with a margin); every poll is
// bounded all the same: a workgroup that waits longer than the launch's patience (50 ms unless the caller says otherwise --
// a product of a grid that is resident at once takes microseconds) sets the top counter's abort bit and the run's sticky
// give-up word, everybody leaves, launches of the same run queued behind it leave as they start, and the host -- which waits
// for the FIRST launch of a run before it queues the others -- falls back to single launches.
// ---------------------------------------------------------------------------
constexpr int kRepeatMaxShards = 32;
constexpr unsigned kRepeatAbort = 0x80000000u;

// ctl_words (kRepeatCtlWords unsigned, every word on a 128-byte line of its own): line 0 the STICKY give-up word of a run -- set by the
// launch that gives up, never cleared between the launches of one run, read by every workgroup of a later launch before it
// does anything: launches already enqueued behind one that gave up leave at once instead of waiting out their own patience;
// lines 1..32 the shard counters, 33..64 the go words, the last line the top counter (bit 31: this launch gave up).
struct RepeatCtl {
    unsigned *sticky = nullptr;  // the run's give-up word (see above)
    unsigned *shard = nullptr;   // `shards` counters, 32 words (128 B) apart
    unsigned *top = nullptr;     // one counter; bit 31 = abort
    unsigned *go = nullptr;      // `shards` generation words, 128 B apart: what the workgroups of a shard poll (sharded grids)
    unsigned long long *stamps = nullptr;  // reps * slots_per_product * 2
    int reps = 0, grid_virtual = 0, slots_per_product = 0;
    int shards = 1;              // 1: every workgroup counts on `top` itself (small grids); else about sqrt(workgroups), a power of two
    unsigned long long patience = 0;  // ticks of the 100 MHz wall clock a workgroup waits at one barrier before the launch gives up
};

template <int VPT, int FLAVOR>
__global__ __launch_bounds__(kStreamBlock) void csr_stream_owner_repeat(
    const int *__restrict__ row_ptr, const int *__restrict__ col_ind, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ tile_row,
    const int *__restrict__ tile_next, int rows, int nnz_arg, int ntiles, int tile_group_arg, const OwnerExtra ex, const RepeatCtl ctl)
{
    __shared__ unsigned go;
    const int t = threadIdx.x;
    const unsigned nwg = gridDim.x, me = blockIdx.x;
    const unsigned S = (unsigned)ctl.shards;
    const unsigned sh = me % S;
    const unsigned members = (nwg - sh + S - 1) / S;   // workgroups of this shard
    const unsigned arrivals = S > 1 ? (nwg < S ? nwg : S) : nwg;  // what `top` counts per product: shards with members, or workgroups
    unsigned long long *stamp = ctl.stamps + 2 * ((size_t)me * (kStreamBlock / 64) + (t >> 6));
    // an earlier launch of this run gave up: so does this one, at once (and says so on its own top word, which the host reads)
    if (t == 0) {
        go = __hip_atomic_load(ctl.sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0u : 1u;
        if (!go)
            atomicOr(ctl.top, kRepeatAbort);
    }
    __syncthreads();
    if (!go)
        return;
    for (int rep = 0; rep < ctl.reps; ++rep) {
        // ---- every workgroup has finished product rep - 1 (its stores acknowledged) before anybody starts product rep
        __syncthreads();
        if (t == 0) {
            const unsigned gen = (unsigned)rep + 1u;
            const unsigned long long t0 = wall_clock64();
            unsigned seen = 0;
            if (S > 1) {
                // arrivals: shard counter, its last arrival on the top counter, ITS last arrival writes the generation into
                // every shard's go word -- which is what a shard's workgroups poll: polls and arrivals never meet on one
                // address (all workgroups polling the top counter held up the arrivals on it: 32 shards 7.0 us per product
                // against 4.96 with 8, memplus.mtx; a one-level form in which every workgroup's first wavefront polled all
                // shard counters ran at 7-15 us: the polls also slow the workgroups that are still multiplying;
                // profiles/r05_cli_n1000.txt)
                const unsigned before = atomicAdd(ctl.shard + 32 * sh, 1u);
                if (before + 1u == members * gen) {
                    const unsigned tb = atomicAdd(ctl.top, 1u);
                    // (a maximum, not a store: a go word that already says kRepeatAbort -- the largest unsigned in play --
                    // keeps saying it, whoever arrives last)
                    if ((tb & ~kRepeatAbort) + 1u == arrivals * gen)
                        for (unsigned q = 0; q < S; ++q)
                            __hip_atomic_fetch_max(ctl.go + 32 * q, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                for (;;) {
                    const unsigned g = __hip_atomic_load(ctl.go + 32 * sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (g >= gen && g != kRepeatAbort)
                        break;
                    if (g == kRepeatAbort) {
                        seen = kRepeatAbort;
                        break;
                    }
                    if (wall_clock64() - t0 > ctl.patience) {
                        atomicOr(ctl.top, kRepeatAbort);
                        __hip_atomic_store(ctl.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (unsigned q = 0; q < S; ++q)
                            __hip_atomic_fetch_max(ctl.go + 32 * q, kRepeatAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        seen = kRepeatAbort;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            } else {
                atomicAdd(ctl.top, 1u);  // (result unused: no round trip in front of the poll)
                for (;;) {
                    seen = __hip_atomic_load(ctl.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((seen & kRepeatAbort) || (seen & ~kRepeatAbort) >= arrivals * gen)
                        break;
                    if (wall_clock64() - t0 > ctl.patience) {
                        atomicOr(ctl.top, kRepeatAbort);
                        __hip_atomic_store(ctl.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        seen = kRepeatAbort;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            go = seen & kRepeatAbort ? 0u : 1u;
        }
        __syncthreads();
        if (!go)
            return;  // (uniform: the whole workgroup read the same word)
        if ((t & 63) == 0)
            stamp[0] = wall_clock64();
        for (int vb = (int)me; vb < ctl.grid_virtual; vb += (int)nwg) {
            if (vb != (int)me)
                __syncthreads();  // the previous tile's LDS is free
            owner_body<VPT, FLAVOR, false>(row_ptr, col_ind, val, x, y, tile_row, tile_next, rows, nnz_arg, ntiles, tile_group_arg, ex, vb);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((t & 63) == 0)
            stamp[1] = wall_clock64();
        stamp += 2 * (size_t)ctl.slots_per_product;
    }
}

// One workgroup per product of a stamped run: min of the waves' first ticks, max of their last.
__global__ __launch_bounds__(256) void stamp_reduce(const unsigned long long *__restrict__ stamps, int slots_per_product,
                                                    unsigned long long *__restrict__ first_last)
{
    __shared__ unsigned long long lo_s[256 / 64], hi_s[256 / 64];
    const unsigned long long *s = stamps + (size_t)blockIdx.x * slots_per_product * 2;
    unsigned long long lo = ~0ull, hi = 0ull;
    for (int i = threadIdx.x; i < slots_per_product; i += 256) {
        const unsigned long long f = s[2 * i], l = s[2 * i + 1];
        lo = f < lo ? f : lo;
        hi = l > hi ? l : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long ol = __shfl_down(lo, off, 64), oh = __shfl_down(hi, off, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        lo_s[threadIdx.x >> 6] = lo;
        hi_s[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / 64; ++w) {
            lo = lo_s[w] < lo ? lo_s[w] : lo;
            hi = hi_s[w] > hi ? hi_s[w] : hi;
        }
        first_last[2 * blockIdx.x] = lo;
        first_last[2 * blockIdx.x + 1] = hi;
    }
}

// One thread per tile; the head of each run of tiles that feed the same row adds
// their carries to that row in tile order.
__global__ __launch_bounds__(256) void csr_stream_carry_fixup(
    const int *__restrict__ carry_row, const double *__restrict__ carry, double *__restrict__ y, int ntiles)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= ntiles)
        return;
    const int r = carry_row[b];
    if (r < 0 || (b > 0 && carry_row[b - 1] == r))
        return;
    double acc = y[r];
    for (int k = b; k < ntiles && carry_row[k] == r; ++k)
        acc += carry[k];
    y[r] = acc;
}

// ---------------------------------------------------------------------------
// K3: TJDS, column-major.  Thread k of a block is permuted column k0 + k: it
// keeps x_perm[k] in a register and walks down its column, one jagged diagonal
// per step, so that at every step the block reads one contiguous run of val /
// row_ind.  The products scatter into y with hardware fp64 atomics (y zeroed by
// the caller, like main-cli.c:1008).  Work items cut the (column block,
// diagonal range) plane into pieces of at most kTjdsBlock x kTjdsDiagChunk
// entries so the few long columns do not serialise the launch.
// ---------------------------------------------------------------------------
template <bool OPERAND_BY_ROW>
__global__ __launch_bounds__(kTjdsBlock) void tjds_colmajor_scatter(
    const int *__restrict__ start_pos, const int *__restrict__ row_ind, const double *__restrict__ val,
    const double *__restrict__ x_perm, double *__restrict__ y, const int4 *__restrict__ work, int cols)
{
    const int4 w = work[blockIdx.x];  // x = first column, y = first diagonal, z = one past the last
    const int k = w.x + threadIdx.x;
    const double xk = (!OPERAND_BY_ROW && k < cols) ? x_perm[k] : 0.0;
    // all loads of the chunk first (they are independent), then the atomics: a loop that
    // loaded and added one diagonal at a time paid one memory round trip per diagonal
    int r[kTjdsDiagChunk];
    double v[kTjdsDiagChunk];
#pragma unroll
    for (int i = 0; i < kTjdsDiagChunk; ++i) {
        const int d = w.y + i;
        r[i] = -1;
        v[i] = 0.0;
        if (d < w.z) {
            const int base = start_pos[d];
            if (k < start_pos[d + 1] - base) {  // diagonal lengths never grow with d
                r[i] = row_ind[base + k];
                v[i] = val[base + k];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kTjdsDiagChunk; ++i) {
        if (r[i] >= 0) {
            // main-cli.c:1018 indexes the permuted operand by the row; the corrected
            // product uses the column's own entry
            const double xv = OPERAND_BY_ROW ? x_perm[r[i]] : xk;
            unsafeAtomicAdd(&y[r[i]], v[i] * xv);
        }
    }
}

// K3': the atomic-free TJDS product, phase 1.  Same traversal as tjds_colmajor_scatter -- thread k of a
// work item is permuted column k0 + k, holds x_perm[k] in a register and walks down its column -- but the
// product of entry j is stored to prod[j] (plain, coalesced store; row_ind is not even read).  Phase 2 sums
// each row's products through the row-inverted index built at create time (csr_stream_owner<4, true>):
// "store every contribution once, then sum per destination", fixed order, so the result is bit-reproducible
// and y needs no zeroing.
__global__ __launch_bounds__(kTjdsBlock) void tjds_colmajor_products(
    const int *__restrict__ start_pos, const double *__restrict__ val, const double *__restrict__ x_perm,
    double *__restrict__ prod, const int4 *__restrict__ work, int cols)
{
    const int4 w = work[blockIdx.x];
    const int k = w.x + threadIdx.x;
    const double xk = k < cols ? x_perm[k] : 0.0;
    int j[kTjdsDiagChunk];
    double v[kTjdsDiagChunk];
#pragma unroll
    for (int i = 0; i < kTjdsDiagChunk; ++i) {
        const int d = w.y + i;
        j[i] = -1;
        v[i] = 0.0;
        if (d < w.z) {
            const int base = start_pos[d];
            if (k < start_pos[d + 1] - base) {
                j[i] = base + k;
                v[i] = val[base + k];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kTjdsDiagChunk; ++i)
        if (j[i] >= 0)
            __builtin_nontemporal_store(v[i] * xk, &prod[j[i]]);
}

__global__ __launch_bounds__(256) void tjds_permute_operand(
    const int *__restrict__ perm, const double *__restrict__ x, double *__restrict__ x_perm, int cols)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < cols)
        x_perm[k] = x[perm[k]];
}

// first index (plus one) whose value lies outside [0, limit): adopted device arrays are checked
// before any kernel indexes with them
__global__ __launch_bounds__(256) void find_out_of_range(const int *__restrict__ a, long long n, int limit,
                                                          int *__restrict__ bad)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int v = a[i];
        if (v < 0 || v >= limit)
            atomicMax(bad, (int)(i < 2147483646ll ? i + 1 : 2147483647ll));
    }
}

// max |v[i]| as the bit pattern of a non-negative double (those order like unsigned integers); *bits = 0 first
__global__ __launch_bounds__(256) void vector_absmax(const double *__restrict__ v, long long n,
                                                      unsigned long long *__restrict__ bits)
{
    double m = 0.0;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double a = fabs(v[i]);
        m = a > m ? a : m;  // NaN entries are skipped
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0)
        atomicMax(bits, (unsigned long long)__double_as_longlong(m));
}

__global__ __launch_bounds__(256) void vector_divide(double *__restrict__ v, long long n,
                                                      const unsigned long long *__restrict__ bits)
{
    const double m = __longlong_as_double((long long)*bits);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && m > 0.0)
        v[i] = v[i] / m;
}

__global__ __launch_bounds__(256) void fill_value(double *__restrict__ p, double v, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        p[i] = v;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_csr_vector(int lanes_per_row, const int *row_ptr, const int *col_ind, const double *val,
                             const double *x, double *y, int rows, hipStream_t stream)
{
    if (rows <= 0)
        return hipSuccess;
    const long long threads = (long long)rows * lanes_per_row;
    const unsigned grid = (unsigned)((threads + kVectorBlock - 1) / kVectorBlock);
#define SMVP_VEC_CASE(T)                                                                              \
    case T:                                                                                           \
        hipLaunchKernelGGL(csr_vector_rows<T>, dim3(grid), dim3(kVectorBlock), 0, stream, row_ptr,   \
                           col_ind, val, x, y, rows);                                                 \
        break;
    switch (lanes_per_row) {
        SMVP_VEC_CASE(2)
        SMVP_VEC_CASE(4)
        SMVP_VEC_CASE(8)
        SMVP_VEC_CASE(16)
        SMVP_VEC_CASE(32)
        SMVP_VEC_CASE(64)
    default:
        return hipErrorInvalidValue;
    }
#undef SMVP_VEC_CASE
    return hipGetLastError();
}

// Tiles per XCD turn for a launch of `ntiles` tiles: kStreamTileGroup, smaller for small matrices so that
// the grid (rounded up to a multiple of 8 * group) is not mostly empty blocks.
// (Groups of 1 ... 2048 were swept in rounds 2 and 5 -- within 2 % from 16 on; profiles/HISTORY_r01_r04.md.)
static int tile_group(int ntiles, int wanted = kStreamTileGroup)
{
    const int g = wanted;
    const int fit = ntiles / 64;
    return fit < 1 ? 1 : (fit < g ? fit : g);
}

hipError_t launch_csr_stream(int vpt, const int *row_ptr, const int *col_ind, const double *val,
                             const double *x, double *y, const int *tile_row, const int *carry_row,
                             double *carry, int rows, int nnz, int ntiles, hipStream_t stream)
{
    if (rows <= 0)
        return hipSuccess;
    const int group = tile_group(ntiles);
    const dim3 grid((unsigned)((ntiles + 8 * group - 1) / (8 * group)) * 8u * group);
    switch (vpt) {
    case 4:
        hipLaunchKernelGGL(csr_stream_tiles<4>, grid, dim3(kStreamBlock), 0, stream, row_ptr, col_ind, val, x, y,
                           tile_row, carry, rows, nnz, ntiles, group);
        break;
    case 8:
        hipLaunchKernelGGL(csr_stream_tiles<8>, grid, dim3(kStreamBlock), 0, stream, row_ptr, col_ind, val, x, y,
                           tile_row, carry, rows, nnz, ntiles, group);
        break;
    default:
        return hipErrorInvalidValue;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    if (ntiles > 1) {
        hipLaunchKernelGGL(csr_stream_carry_fixup, dim3((ntiles + 255) / 256), dim3(256), 0, stream, carry_row,
                           carry, y, ntiles);
        e = hipGetLastError();
    }
    return e;
}

// grid of a launch of `ntiles` tiles (a multiple of 8 * group, see tile_of_block)
static unsigned owner_grid(int ntiles, int group) { return (unsigned)((ntiles + 8 * group - 1) / (8 * group)) * 8u * group; }

// tiles per XCD turn by flavour.  CSR measured best at 64 (profiles/r01_tile_group_sweep.txt).  The tile-ordered TJDS stream: the
// group decides how often x_perm crosses the fabric -- a copy of memplus is 61 tiles, every XCD that gets some of them pulls
// the copy's part of x_perm -- but not the time (round 5, memplus x944, PMC FETCH per product / ms; x_perm's share in brackets):
// group 4: 2125 MB (388)   16: 2059 (333) 0.390   32: 1983 (258) 0.391   48: 1934 (211) 0.392   64: 1907 (185) 0.395   256: 1841 (126)
// pwt x459: 0.263-0.267 / 0.265-0.267 / 0.267-0.268 / 0.269-0.270 ms for 16 / 32 / 48 / 64.  32 moves 76 MB less at the same speed.
static int flavor_group(int flavor) { return flavor == kFlavorTjdsS || flavor == kFlavorTjdsH ? kTjdsTileGroup : kStreamTileGroup; }

int owner_stamp_slots(int ntiles, int flavor)
{
    return (int)owner_grid(ntiles, tile_group(ntiles, flavor_group(flavor))) * (kStreamBlock / 64);
}

hipError_t launch_csr_stream_owner(int vpt, int flavor, const OwnerLaunch &l, hipStream_t stream)
{
    if (l.rows <= 0)
        return hipSuccess;
    const int group = tile_group(l.ntiles, flavor_group(flavor));
    const dim3 grid(owner_grid(l.ntiles, group));
    const OwnerExtra ex = owner_extra_of(l, l.stamps);
#define SMVP_OWNER_ST(V, F, S)                                                                                     \
    hipLaunchKernelGGL((csr_stream_owner<V, F, S>), grid, dim3(kStreamBlock), 0, stream, l.row_ptr, l.col_ind, l.val, \
                       l.x, l.y, l.tile_row, l.tile_next, l.rows, l.nnz, l.ntiles, group, ex)
#define SMVP_OWNER(V, F)                  \
    if (vpt == V && flavor == F) {        \
        if (l.stamps)                     \
            SMVP_OWNER_ST(V, F, true);    \
        else                              \
            SMVP_OWNER_ST(V, F, false);   \
        return hipGetLastError();         \
    }
#ifdef SMVP_PHASE_STAMPS   // (diagnostic builds keep every flavour here: one set of phase counters)
    SMVP_OWNER(1, kFlavorCsr)
    SMVP_OWNER(4, kFlavorCsr)
    SMVP_OWNER(8, kFlavorCsr)
    SMVP_OWNER(4, kFlavorCsr16)
    SMVP_OWNER(8, kFlavorCsr16)
#else
    // the CSR flavours are compiled in the second translation unit of this file (SMVP_TU_ILP: the max-ILP scheduling strategy
    // suits them -- memplus x944 0.2866 -> 0.2800 ms -- and costs the TJDS flavours 2-3 %)
    if (flavor == kFlavorCsr || flavor == kFlavorCsr16)
        return launch_owner_csr_ilp(vpt, flavor, l, &ex, grid.x, group, stream);
#endif
    SMVP_OWNER(1, kFlavorTjdsK)
    SMVP_OWNER(4, kFlavorTjdsK)
    SMVP_OWNER(8, kFlavorTjdsK)
    SMVP_OWNER(1, kFlavorTjdsS)
    SMVP_OWNER(4, kFlavorTjdsS)
    SMVP_OWNER(8, kFlavorTjdsS)
    SMVP_OWNER(1, kFlavorTjdsH)
    SMVP_OWNER(4, kFlavorTjdsH)
    SMVP_OWNER(8, kFlavorTjdsH)
#undef SMVP_OWNER
#undef SMVP_OWNER_ST
    return hipErrorInvalidValue;
}

// ---- the repeating form (csr_stream_owner_repeat): `reps` products in one launch
// Workgroups of the repeating launch for a plan of `ntiles` tiles: the plain launch's grid -- every workgroup its one tile per
// product, as in a launch of its own -- if that grid is resident at once with room to spare (the occupancy query can read one
// workgroup per CU high, MI355X_MICROARCH "Residency"); 0: the grid is larger (a capped grid whose workgroups walk several tiles
// per product runs each product slower than a launch of its own does -- memplus x2 ... x32: windows of 4.8 ... 17.9 us against
// 3.3 ... 12.8, profiles/r05_cli_n1000.txt -- so those matrices keep one launch per product), this (vpt, flavor) has no repeating
// form, or the device could not be asked.
template <int V, int F>
static int repeat_capacity()
{
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, csr_stream_owner_repeat<V, F>, kStreamBlock, 0) != hipSuccess ||
        hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return per_cu > 1 ? (per_cu - 1) * cus * 3 / 4 : 0;  // three quarters of (one workgroup per CU fewer than the query's answer)
}

#define SMVP_REPEAT_FORMS(X) \
    X(1, kFlavorCsr) X(4, kFlavorCsr) X(8, kFlavorCsr) X(4, kFlavorCsr16) X(8, kFlavorCsr16) \
    X(1, kFlavorTjdsS) X(4, kFlavorTjdsS) X(8, kFlavorTjdsS) X(1, kFlavorTjdsH) X(4, kFlavorTjdsH) X(8, kFlavorTjdsH)

int owner_repeat_grid(int vpt, int flavor, int ntiles)
{
    if (ntiles <= 0)
        return 0;
    int cap = 0;
#define X(V, F)                     \
    if (vpt == V && flavor == F)    \
        cap = repeat_capacity<V, F>();
    SMVP_REPEAT_FORMS(X)
#undef X
    if (cap <= 0)
        return 0;
    const int full = (int)owner_grid(ntiles, tile_group(ntiles, flavor_group(flavor)));
    return full <= cap ? full : 0;
}

// `reps` products, each stamped: stamps[reps][grid * 4][2]; ctl_words: kRepeatCtlWords unsigned -- the counters are cleared here
// for every launch, the sticky give-up word only for the first launch of a run.  Refuses (hipErrorInvalidValue, nothing is
// launched) a launch whose control block would be incomplete: no stamps, no control words, no products, no tiles to walk.
hipError_t launch_csr_stream_owner_repeat(int vpt, int flavor, const OwnerLaunch &l, int reps, int grid, unsigned *ctl_words,
                                          bool first_of_run, unsigned long long patience_ticks, hipStream_t stream)
{
    if (l.rows <= 0)
        return hipSuccess;
    const int group = tile_group(l.ntiles, flavor_group(flavor));
    RepeatCtl ctl{};
    ctl.sticky = ctl_words;
    ctl.shard = ctl_words + 32, ctl.go = ctl_words + 32 * (1 + kRepeatMaxShards), ctl.top = ctl_words + kRepeatCtlWords - 32;
    // arrivals cost about 12 ns each on one address: n / S on a shard, then S on the top counter
    // measured (profiles/r05_cli_n1000.txt): one level up to two dozen workgroups, 8 shards up to ~600, 16 beyond
    ctl.shards = grid <= 24 ? 1 : grid <= 640 ? 8 : 16;
    ctl.stamps = l.stamps;
    ctl.reps = reps, ctl.grid_virtual = (int)owner_grid(l.ntiles, group), ctl.slots_per_product = grid * (kStreamBlock / 64);
    ctl.patience = patience_ticks;
    if (reps <= 0 || grid <= 0 || ctl.grid_virtual <= 0 || !ctl.stamps || !ctl_words)
        return hipErrorInvalidValue;
    static_assert(kRepeatCtlWords == 32 * (2 + 2 * kRepeatMaxShards), "sticky + shards + go words + top, one 128-byte line each");
    hipError_t e = first_of_run ? hipMemsetAsync(ctl_words, 0, sizeof(unsigned) * kRepeatCtlWords, stream)
                                : hipMemsetAsync(ctl_words + 32, 0, sizeof(unsigned) * (kRepeatCtlWords - 32), stream);
    if (e != hipSuccess)
        return e;
    const OwnerExtra ex = owner_extra_of(l, nullptr);
#define X(V, F)                                                                                                                    \
    if (vpt == V && flavor == F) {                                                                                                 \
        hipLaunchKernelGGL((csr_stream_owner_repeat<V, F>), dim3((unsigned)grid), dim3(kStreamBlock), 0, stream, l.row_ptr, l.col_ind, \
                           l.val, l.x, l.y, l.tile_row, l.tile_next, l.rows, l.nnz, l.ntiles, group, ex, ctl);                     \
        return hipGetLastError();                                                                                                  \
    }
    SMVP_REPEAT_FORMS(X)
#undef X
    return hipErrorInvalidValue;
}

#ifdef SMVP_PHASE_STAMPS
void debug_owner_phases()
{
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_owner_phase), sizeof h) != hipSuccess || h[7] == 0)
        return;
    const double n = (double)h[7];
    fprintf(stderr, "[owner dbg] mean us per workgroup over %.0f: loads + gathers + LDS writes %.2f | barrier %.2f | lane-per-row sums %.2f | "
                    "barrier %.2f | long rows %.2f\n", n, h[0] * 0.01 / n, h[1] * 0.01 / n, h[2] * 0.01 / n, h[3] * 0.01 / n, h[4] * 0.01 / n);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_owner_phase), z, sizeof z);
}
#else
void debug_owner_phases() {}
#endif

hipError_t launch_stamp_reduce(const unsigned long long *stamps, int slots_per_product, int products,
                               unsigned long long *first_last, hipStream_t stream)
{
    if (products <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(stamp_reduce, dim3(products), dim3(256), 0, stream, stamps, slots_per_product, first_last);
    return hipGetLastError();
}

hipError_t launch_tjds_products(const int *start_pos, const double *val, const double *x_perm, double *prod,
                                const int4 *work, int nwork, int cols, hipStream_t stream)
{
    if (nwork <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(tjds_colmajor_products, dim3(nwork), dim3(kTjdsBlock), 0, stream, start_pos, val, x_perm, prod,
                       work, cols);
    return hipGetLastError();
}

hipError_t launch_tjds_scatter(bool operand_by_row, const int *start_pos, const int *row_ind, const double *val,
                               const double *x_perm, double *y, const int4 *work, int nwork, int cols,
                               hipStream_t stream)
{
    if (nwork <= 0)
        return hipSuccess;
    if (operand_by_row)
        hipLaunchKernelGGL(tjds_colmajor_scatter<true>, dim3(nwork), dim3(kTjdsBlock), 0, stream, start_pos,
                           row_ind, val, x_perm, y, work, cols);
    else
        hipLaunchKernelGGL(tjds_colmajor_scatter<false>, dim3(nwork), dim3(kTjdsBlock), 0, stream, start_pos,
                           row_ind, val, x_perm, y, work, cols);
    return hipGetLastError();
}

hipError_t launch_tjds_permute(const int *perm, const double *x, double *x_perm, int cols, hipStream_t stream)
{
    if (cols <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(tjds_permute_operand, dim3((cols + 255) / 256), dim3(256), 0, stream, perm, x, x_perm, cols);
    return hipGetLastError();
}

hipError_t launch_find_out_of_range(const int *a, long long n, int limit, int *bad, hipStream_t stream)
{
    if (n <= 0)
        return hipSuccess;
    const long long want = (n + 255) / 256;
    hipLaunchKernelGGL(find_out_of_range, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, stream, a, n,
                       limit, bad);
    return hipGetLastError();
}

// v <- v / max|v| (left alone when the maximum is 0); scratch = one unsigned long long of device memory
hipError_t launch_normalize_max(double *v, long long n, unsigned long long *scratch, hipStream_t stream)
{
    if (n <= 0)
        return hipSuccess;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess)
        return e;
    const long long want = (n + 255) / 256;
    hipLaunchKernelGGL(vector_absmax, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, stream, v, n, scratch);
    hipLaunchKernelGGL(vector_divide, dim3((unsigned)want), dim3(256), 0, stream, v, n, scratch);
    return hipGetLastError();
}

hipError_t launch_fill(double *p, double v, long long n, hipStream_t stream)
{
    if (n <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(fill_value, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, v, n);
    return hipGetLastError();
}

#else  // SMVP_TU_ILP: the plain launches of the CSR flavours (launch_csr_stream_owner in the other unit prepares `extra`, grid and group)
#ifdef SMVP_ILP_UNIT_UNUSED   // a -DSMVP_PHASE_STAMPS diagnostic build: every flavour is launched from the other unit, nothing is instantiated here
hipError_t launch_owner_csr_ilp(int, int, const OwnerLaunch &, const void *, unsigned, int, hipStream_t) { return hipErrorInvalidValue; }
#else
hipError_t launch_owner_csr_ilp(int vpt, int flavor, const OwnerLaunch &l, const void *extra, unsigned grid_x, int group, hipStream_t stream)
{
    const OwnerExtra &ex = *static_cast<const OwnerExtra *>(extra);
    const dim3 grid(grid_x);
#define SMVP_OWNER_ST(V, F, S)                                                                                     \
    hipLaunchKernelGGL((csr_stream_owner<V, F, S>), grid, dim3(kStreamBlock), 0, stream, l.row_ptr, l.col_ind, l.val, \
                       l.x, l.y, l.tile_row, l.tile_next, l.rows, l.nnz, l.ntiles, group, ex)
#define SMVP_OWNER(V, F)                  \
    if (vpt == V && flavor == F) {        \
        if (l.stamps)                     \
            SMVP_OWNER_ST(V, F, true);    \
        else                              \
            SMVP_OWNER_ST(V, F, false);   \
        return hipGetLastError();         \
    }
    SMVP_OWNER(1, kFlavorCsr)
    SMVP_OWNER(4, kFlavorCsr)
    SMVP_OWNER(8, kFlavorCsr)
    SMVP_OWNER(4, kFlavorCsr16)
    SMVP_OWNER(8, kFlavorCsr16)
#undef SMVP_OWNER
#undef SMVP_OWNER_ST
    return hipErrorInvalidValue;
}
#endif  // SMVP_ILP_UNIT_UNUSED
#endif  // SMVP_TU_ILP

}  // namespace smvp

// ---------------------------------------------------------------------------
// K4: CSR, column-swept row strips -- for matrices whose columns scatter over an operand far larger than L2
// (BASELINE config 4: 32 uniform columns per row over an 80 MB x).  The tile kernels above then run at the chip's
// L2-miss gather rate (about 54 G gathers/s, 8.4 % of HBM peak on config 4) whatever they do, because every gather
// fetches its own line from the Infinity Cache.  Here the entries are kept a second time (the plan; row_ptr /
// col_ind / val stay as they are), ordered by (row strip, column) with a 16-bit word per entry: the row's number inside
// the strip and the entry's turn (below).  One WAVEFRONT owns a strip of up to 2048 rows: it keeps the strip's sums in
// its own quarter of the workgroup's LDS and streams the strip's entries in ascending column order, 256 at a time, so
// all the wavefronts that run together gather from one window of x that slides over the operand once per product and
// fits the XCD's 4 MB L2 (measured 135 G gathers/s on config 4: 2.4 ms against 5.98).  The launches are cut into
// generations of workgroups that are resident together and start together (per_launch), which is what keeps their
// windows aligned; a strip's sums are complete when its wavefront ends.
//
// Every row is summed in ascending column order -- the order of the serial loop, main-cli.c:410-416 -- so the result
// is bit-identical to it and the same from run to run.  (Round 2 added the products with LDS atomics in arrival order.)
// A strip belongs to one wavefront, whose LDS instructions execute in program order, so only entries of one row that
// meet in the same chunk of 256 need ordering: the plan gives each entry its turn, the number of earlier entries of
// its row in its chunk, entries of equal turn never share a row, and the wavefront adds turn 0 (nearly everything),
// then turn 1, ...  Turns from kSweepTurnCap on (a dense row in a narrow band of columns: not what this kernel is for,
// but it must stay right) are added one lane at a time in stream order.  No atomics, no barriers.
// ---------------------------------------------------------------------------
namespace smvp {

#ifdef SMVP_TU_ILP   // (max-ILP scheduling: config 4's structure 1.767 -> 1.60 ms)
// One iteration of a wavefront takes G chunks of its strip (G * 256 entries, G * 4 per lane): all their gathers are in
// flight together, the plan words of the next iteration are requested before the sums are touched, and the chunks are
// then added one after the other (a chunk's turns are counted inside the chunk, so this keeps every row in order).
template <int G>
__global__ __launch_bounds__(kSweepBlock) void csr_colsweep(
    const long long *__restrict__ strip_ptr, const int *__restrict__ e_col, const double *__restrict__ e_val,
    const unsigned short *__restrict__ e_row, const double *__restrict__ x, double *__restrict__ y, int rows, int strip_rows,
    int nstrips, int wg_first, int parts, double *__restrict__ ypart, long long ypart_stride)
{
    constexpr int U = kSweepUnroll, E = G * U;
    constexpr long long STEP = (long long)G * kSweepChunk;
    extern __shared__ double sums_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // `strip` numbers the STREAMS: strip * 1 (parts == 1: a stream is a strip's) or real strip * parts + column part
    int strip = (wg_first + (int)blockIdx.x) * kSweepWaves + wave;
    bool live = strip < nstrips * parts;
    long long part_rows0 = -1;  // parts == kSweepXcdParts: where this wavefront's partial sums go in ypart
    if (parts == kSweepXcdParts) {
        // XCD-private column parts: workgroup b of a launch runs on XCD b % 8 and takes column part b % 8 of the four strips of row
        // group b / 8 -- every XCD gathers from its own eighth of x only; the wavefronts keep whole strips, the eight partial sums
        // of a row meet in ypart and are added part by part by sweep_combine (reproducible, not the serial loop's bits)
        const int b = wg_first + (int)blockIdx.x, part = b % kSweepXcdParts, real = (b / kSweepXcdParts) * kSweepWaves + wave;
        live = real < nstrips;
        strip = real * kSweepXcdParts + part;
        part_rows0 = (long long)part * ypart_stride + (long long)real * strip_rows;
        if (!live)
            return;
    }
    if (!live && parts == 1)
        return;  // (no barrier in the one-part form)
    // volatile: every access is issued where it stands -- another lane's earlier store must be seen; the LDS address
    // space is spelled out so that these stay ds_read / ds_write
    typedef __attribute__((address_space(3))) volatile double lds_double;
    lds_double *sums = (lds_double *)sums_all + (size_t)wave * strip_rows;
    const long long a = live ? strip_ptr[strip] : 0, z = live ? strip_ptr[strip + 1] : 0;

    // entry e of an iteration starting at `base`: stream position base + e * 64 + lane; w = row | turn << kSweepRowBits,
    // -1 where the stream has ended
    auto fetch = [&](long long base, int (&c)[E], double (&v)[E], int (&w)[E]) {
        if (base + STEP <= z) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const long long j = base + e * 64 + lane;
                c[e] = __builtin_nontemporal_load(e_col + j);
                v[e] = __builtin_nontemporal_load(e_val + j);
                w[e] = __builtin_nontemporal_load(e_row + j);
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const long long j = base + e * 64 + lane;
                const bool in = j < z;
                c[e] = in ? e_col[j] : 0;
                v[e] = in ? e_val[j] : 0.0;
                w[e] = in ? (int)e_row[j] : -1;
            }
        }
    };
    int c[E], w[E];
    double v[E];
    if (a < z)
        fetch(a, c, v, w);
    for (int i = lane; i < strip_rows; i += 64)
        sums[i] = 0.0;
    for (long long base = a; base < z; base += STEP) {
        double p[E];
#pragma unroll
        for (int e = 0; e < E; ++e)
            p[e] = x[c[e]];
        int cn[E], wn[E];
        double vn[E];
        if (base + STEP < z)
            fetch(base + STEP, cn, vn, wn);
#pragma unroll
        for (int e = 0; e < E; ++e)
            p[e] = v[e] * p[e];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int row[U], turn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                row[u] = w[g * U + u] < 0 ? 0 : w[g * U + u] & ((1 << kSweepRowBits) - 1);
                turn[u] = w[g * U + u] >> kSweepRowBits;  // -1 stays -1: no entry
            }
            for (int k = 0;; ++k) {
                double cur[U];
                bool later = false;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    cur[u] = sums[row[u]];  // every lane reads (row 0 where it has no entry): four reads in flight, no branches
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (turn[u] == k)
                        sums[row[u]] = cur[u] + p[g * U + u];
                    later = later || (turn[u] > k && turn[u] < kSweepTurnCap);
                }
                if (!__any(later))
                    break;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                unsigned long long m = __ballot(turn[u] == kSweepTurnCap);
                while (m) {  // lane by lane, i.e. in stream order
                    const int l = __ffsll((long long)m) - 1;
                    if (lane == l)
                        sums[row[u]] = sums[row[u]] + p[g * U + u];
                    m &= m - 1;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            c[e] = cn[e];
            v[e] = vn[e];
            w[e] = wn[e];
        }
    }
    if (parts == kSweepXcdParts) {
        const long long r0 = part_rows0 - (part_rows0 / ypart_stride) * ypart_stride;  // the strip's first row
        for (int i = lane; i < strip_rows && r0 + i < rows; i += 64)
            ypart[part_rows0 + i] = sums[i];  // (read again by sweep_combine: through the caches)
        return;
    }
    if (parts == 1) {
        const long long r0 = (long long)strip * strip_rows;
        for (int i = lane; i < strip_rows && r0 + i < rows; i += 64)
            __builtin_nontemporal_store(sums[i], &y[r0 + i]);
        return;
    }
    // Column parts: the wavefronts w * parts ... w * parts + parts - 1 of this workgroup hold the partial sums of ONE strip over
    // their parts of the columns (each summed in ascending column order); a row's sum is its partial sums added in the
    // order of the parts -- ascending columns still, but associated part by part: reproducible from run to run, within the
    // rounding bound, NOT the serial loop's bits.
    __syncthreads();
    const int strips_here = kSweepWaves / parts;
    lds_double *all = (lds_double *)sums_all;
    for (int s = 0; s < strips_here; ++s) {
        const long long r0 = ((long long)(wg_first + (int)blockIdx.x) * strips_here + s) * strip_rows;
        for (int i = (int)threadIdx.x; i < strip_rows && r0 + i < rows; i += kSweepBlock) {
            double acc = all[(size_t)(s * parts) * strip_rows + i];
            for (int q = 1; q < parts; ++q)
                acc += all[(size_t)(s * parts + q) * strip_rows + i];
            __builtin_nontemporal_store(acc, &y[r0 + i]);
        }
    }
}

// XCD-private column parts: y[r] = the eight partial sums of row r, added in the order of the parts
__global__ __launch_bounds__(256) void sweep_combine(const double *__restrict__ ypart, long long stride, double *__restrict__ y, int rows)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows)
        return;
    double acc = ypart[r];
#pragma unroll
    for (int q = 1; q < kSweepXcdParts; ++q)
        acc += ypart[(long long)q * stride + r];
    __builtin_nontemporal_store(acc, &y[r]);
}

// Chunks in flight per wavefront.  A launch generation is one workgroup per CU (four wavefronts): with tall strips two chunks
// in flight make up for the missing wavefronts, with short ones they only make the window race (the strips' streams are
// short).  Measured (MI355X, config 4's columns; ms for G = 1 / 2; profiles/r05_colsweep_heights.txt): strips of 2048 rows
// 2.55 / 2.25 (10 M rows), 0.497 / 0.444 (1.25 M rows); 1221 rows 0.315 / 0.287 (1.25 M), 0.632 / 0.575 (2.5 M); 814 rows
// 0.648 / 0.636; 611 rows 0.328 / 0.373; 512 rows 0.413 / 0.468; 407 rows 0.348 / 0.459 -- two from about 800 rows on.
// `asked` = 1 | 2 | 4 overrides (SMVP_CSR_SWEEP_PARAM's third field: experiments); 0 = the rule.  Asked once per plan build
// (the engine keeps the answer in the handle), never on a launch path.
int sweep_chunks_in_flight(int strip_rows, int asked)
{
    return asked == 1 || asked == 2 || asked == 4 ? asked : (strip_rows >= 768 ? 2 : 1);
}

// Strips taller than 2048 rows need more dynamic LDS (up to the CU's 160 KB) than a kernel gets unasked: asked for here, for every
// form of the kernel, on the CURRENT device -- once per plan build (the engine), never on a launch path; always the same
// maximum, so that plans of different heights on one device cannot take it from one another.
constexpr size_t kSweepMaxLds = 160u * 1024;
hipError_t prepare_csr_colsweep()
{
    for (const void *f : {(const void *)csr_colsweep<1>, (const void *)csr_colsweep<2>, (const void *)csr_colsweep<4>}) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSweepMaxLds);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

hipError_t launch_csr_colsweep(const long long *strip_ptr, const int *e_col, const double *e_val, const unsigned short *e_row,
                               const double *x, double *y, int rows, int strip_rows, int parts, int per_launch, int g, double *ypart,
                               hipStream_t stream)
{
    if (rows <= 0)
        return hipSuccess;
    const bool xcd = parts == kSweepXcdParts;
    if ((parts != 1 && parts != 2 && parts != 4 && !xcd) || (xcd && (!ypart || (per_launch > 0 && per_launch % kSweepXcdParts != 0))))
        return hipErrorInvalidValue;
    const int nstrips = (rows + strip_rows - 1) / strip_rows;
    const int nwg = xcd ? (nstrips + kSweepWaves - 1) / kSweepWaves * kSweepXcdParts
                        : (int)(((long long)nstrips * parts + kSweepWaves - 1) / kSweepWaves);
    const long long ypart_stride = rows;
    const size_t lds = sizeof(double) * (size_t)strip_rows * kSweepWaves;
    if (per_launch <= 0)
        per_launch = nwg;
    if (lds > kSweepMaxLds)
        return hipErrorInvalidValue;  // (strips of more than 5120 rows; up to there prepare_csr_colsweep has asked for the LDS)
    for (int first = 0; first < nwg; first += per_launch) {
        const unsigned grid = (unsigned)(nwg - first < per_launch ? nwg - first : per_launch);
#define SMVP_SWEEP(GG)                                                                                                   \
    hipLaunchKernelGGL(csr_colsweep<GG>, dim3(grid), dim3(kSweepBlock), lds, stream, strip_ptr, e_col, e_val, e_row, x, y, \
                       rows, strip_rows, nstrips, first, parts, ypart, ypart_stride)
        if (g == 4)
            SMVP_SWEEP(4);
        else if (g == 2)
            SMVP_SWEEP(2);
        else
            SMVP_SWEEP(1);
#undef SMVP_SWEEP
    }
    if (xcd)
        hipLaunchKernelGGL(sweep_combine, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, ypart, ypart_stride, y, rows);
    return hipGetLastError();
}

#else  // !SMVP_TU_ILP
// How scattered are the gathers of a CSR matrix?  Sample s of `samples` looks at kSpreadSpan consecutive entries --
// what one XCD's turn of 64 tiles of the tile kernel gathers -- and counts the distinct 128-byte lines of x they touch,
// by linear counting: every line sets one hashed bit of a 512 Kbit LDS map; the host turns the number of set bits into
// the estimate (engine: csr_gather_spread).  A count near the number of entries means that nearly every gather pulls
// its own line through the L2: the matrix is one for the column sweep.
__global__ __launch_bounds__(1024) void csr_line_spread(const int *__restrict__ col_ind, long long nnz, int samples,
                                                        int *__restrict__ set_bits)
{
    constexpr int WORDS = 16 * 1024;  // 512 Kbit
    __shared__ unsigned map[WORDS];
    __shared__ int total;
    for (int i = threadIdx.x; i < WORDS; i += 1024)
        map[i] = 0u;
    if (threadIdx.x == 0)
        total = 0;
    __syncthreads();
    const long long first = samples > 1 ? (nnz - kSpreadSpan) / (samples - 1) * blockIdx.x : 0;
    for (int i = threadIdx.x; i < kSpreadSpan; i += 1024) {
        const long long j = first + i;
        if (j < nnz) {
            const unsigned line = (unsigned)col_ind[j] >> 4;
            const unsigned h = (line * 2654435761u) >> (32 - 19);
            atomicOr(&map[h >> 5], 1u << (h & 31));
        }
    }
    __syncthreads();
    int n = 0;
    for (int i = threadIdx.x; i < WORDS; i += 1024)
        n += __popc(map[i]);
    atomicAdd(&total, n);
    __syncthreads();
    if (threadIdx.x == 0)
        set_bits[blockIdx.x] = total;
}

hipError_t launch_csr_line_spread(const int *col_ind, long long nnz, int samples, int *set_bits, hipStream_t stream)
{
    if (samples <= 0 || nnz < kSpreadSpan)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(csr_line_spread, dim3(samples), dim3(1024), 0, stream, col_ind, nnz, samples, set_bits);
    return hipGetLastError();
}

#endif  // SMVP_TU_ILP

}  // namespace smvp
