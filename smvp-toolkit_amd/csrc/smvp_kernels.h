// smvp_kernels.h -- internal interface between the kernel TU and the engine TU.
#pragma once
#include <hip/hip_runtime.h>

namespace smvp {

constexpr int kVectorBlock = 256;   // threads per block, csr_vector_rows
constexpr int kStreamBlock = 256;   // threads per block, csr_stream_tiles
constexpr int kLongRow = 32;        // segments longer than this are summed by a wavefront
constexpr int kStreamOver = 1024;    // entries past its end a tile may finish its last row with, through LDS
constexpr int kStreamTileGroup = 64;  // consecutive tiles per XCD turn (see tile_of_block)
constexpr int kTjdsTileGroup = 32;    // ... for the tile-ordered TJDS stream (16 until round 5)
constexpr int kSweepBlock = 256;   // threads per block, csr_colsweep: four wavefronts, each with its own strip of rows
constexpr int kSweepWaves = kSweepBlock / 64;
constexpr int kSweepUnroll = 4;    // stream entries per lane and pass: a wavefront takes its strip 256 entries at a time
constexpr int kSweepChunk = 64 * kSweepUnroll;
constexpr int kSweepXcdParts = 8;  // XCD-private column parts: one per XCD (workgroup b of a launch runs on XCD b % 8)
constexpr int kSweepRowBits = 13;  // a strip holds at most 8192 rows (64 KB of sums in LDS; four strips of 5120 fill a CU's 160 KB) ...
constexpr int kSweepTurnCap = (1 << (16 - kSweepRowBits)) - 1;  // ... and the 16-bit row word carries the entry's turn, capped
constexpr int kTjdsBlock = 256;     // permuted columns per work item
constexpr int kTjdsDiagChunk = 8;   // jagged diagonals per work item

hipError_t launch_csr_vector(int lanes_per_row, const int *row_ptr, const int *col_ind, const double *val,
                             const double *x, double *y, int rows, hipStream_t stream);
hipError_t launch_csr_stream(int vpt, const int *row_ptr, const int *col_ind, const double *val,
                             const double *x, double *y, const int *tile_row, const int *carry_row,
                             double *carry, int rows, int nnz, int ntiles, hipStream_t stream);
// what an entry of the owner kernel's stream is (see csr_stream_owner)
constexpr int kFlavorCsr = 0;    // val[j] * x[col_ind[j]]
                                 // (1 was a unit-value CSR: the two-phase TJDS product's second phase until round 5)
constexpr int kFlavorTjdsK = 2;  // val[pos[j]] * x_perm[col_ind[j]]   (col_ind = permuted column k)
constexpr int kFlavorTjdsS = 3;  // the same entries, every tile's in TJDS order; col_ind = LDS slot | diagonal << kSlotBits
constexpr int kFlavorTjdsH = 4;  // kFlavorTjdsS with a 16-bit second word: slot | run hint << 11; the start_pos of the entry's
                                 // diagonal comes from the tile's run table (6 bytes of index per entry instead of 8)
constexpr int kFlavorCsr16 = 5;  // kFlavorCsr reading 16-bit column offsets: x[col_base[tile] + col16[j]] (tiles whose columns span < 65536)
constexpr int kSlotBits = 11;    // a tile holds at most 2048 entries

struct OwnerLaunch {
    const int *row_ptr = nullptr, *col_ind = nullptr;
    const double *val = nullptr, *x = nullptr;
    double *y = nullptr;
    const int *tile_row = nullptr, *tile_next = nullptr;
    const int *pos = nullptr;
    const int *start_pos = nullptr;
    unsigned long long *stamps = nullptr;  // owner_stamp_slots(ntiles) pairs, or nullptr
    const int *ovf_ptr = nullptr, *ovf_k = nullptr;  // kFlavorTjdsS / H: the tiles' overflow entries (permuted column ...
    const double *ovf_val = nullptr;                 // ... and value), row order
    const int *cache_ptr = nullptr;                                      // kFlavorTjdsS
    const double *val_cache = nullptr;
    const unsigned short *col16 = nullptr;                               // kFlavorCsr16
    const int *col_base = nullptr;
    const unsigned short *group_run = nullptr;        // kFlavorTjdsH
    const unsigned *word32 = nullptr;
    int unit_x = 0;                            // kFlavorTjdsH: the operand is the unit vector (second phase of the two-phase TJDS product)
    const int *run_ptr = nullptr, *run_tab = nullptr;
    const unsigned short *row_rel = nullptr;   // rows' first entries relative to their tile's (or nullptr: row_ptr is read)
    int rows = 0, nnz = 0, ntiles = 0;
};
hipError_t launch_csr_stream_owner(int vpt, int flavor, const OwnerLaunch &l, hipStream_t stream);
// diagnostic builds (make HIPFLAGS+=-DSMVP_PHASE_STAMPS): print and clear the mean time a workgroup of the owner kernel / of
// the binned plan's pass B spent in each of its phases; nothing in a normal build
void debug_owner_phases();
void debug_binned_phases();
int owner_stamp_slots(int ntiles, int flavor);  // {first, last} tick pairs one stamped launch writes
// the repeating form: `reps` products in one launch, every product's window stamped (csr_stream_owner_repeat).  grid =
// owner_repeat_grid(...) workgroups (0: no such form for this plan / device), stamps = reps * grid * 4 {first, last} pairs in
// l.stamps, ctl_words = kRepeatCtlWords unsigned of device memory; *top bit 31 set afterwards = the launch gave up (abort)
constexpr int kRepeatCtlWords = 32 * 66;  // the run's sticky give-up word, 32 shard counters, 32 go words, the top counter (bit 31: abort; the last line), each on a 128-byte line of its own
constexpr unsigned kRepeatPatienceDefaultUs = 50000;  // how long a workgroup waits at one barrier before the launch gives up
int owner_repeat_grid(int vpt, int flavor, int ntiles);
// (the CSR flavours' plain launches live in the second translation unit of smvp_kernels.hip, compiled -DSMVP_TU_ILP with the
// max-ILP scheduling strategy -- see the Makefile; `extra` is the launcher's OwnerExtra)
hipError_t launch_owner_csr_ilp(int vpt, int flavor, const OwnerLaunch &l, const void *extra, unsigned grid_x, int group, hipStream_t stream);
// first_of_run: clears the sticky give-up word too (later launches of the run leave at once when it is set); patience_ticks: of the
// 100 MHz wall clock (repeat_patience_ticks below)
hipError_t launch_csr_stream_owner_repeat(int vpt, int flavor, const OwnerLaunch &l, int reps, int grid, unsigned *ctl_words,
                                          bool first_of_run, unsigned long long patience_ticks, hipStream_t stream);
// smvp_run_opts_t.repeat_patience_us -> ticks: 0 = the default, negative = no patience at all (every workgroup that has to wait gives up)
inline unsigned long long repeat_patience_ticks(int us) { return us < 0 ? 0ull : 100ull * (unsigned)(us ? us : (int)kRepeatPatienceDefaultUs); }
hipError_t launch_stamp_reduce(const unsigned long long *stamps, int slots_per_product, int products,
                               unsigned long long *first_last, hipStream_t stream);
hipError_t launch_tjds_products(const int *start_pos, const double *val, const double *x_perm, double *prod,
                                const int4 *work, int nwork, int cols, hipStream_t stream);
hipError_t launch_tjds_scatter(bool operand_by_row, const int *start_pos, const int *row_ind, const double *val,
                               const double *x_perm, double *y, const int4 *work, int nwork, int cols,
                               hipStream_t stream);
hipError_t launch_tjds_permute(const int *perm, const double *x, double *x_perm, int cols, hipStream_t stream);
hipError_t launch_find_out_of_range(const int *a, long long n, int limit, int *bad, hipStream_t stream);
hipError_t launch_normalize_max(double *v, long long n, unsigned long long *scratch, hipStream_t stream);
hipError_t launch_fill(double *p, double v, long long n, hipStream_t stream);

hipError_t launch_csr_colsweep(const long long *strip_ptr, const int *e_col, const double *e_val, const unsigned short *e_row,
                               const double *x, double *y, int rows, int strip_rows, int parts, int per_launch, int g, double *ypart,
                               hipStream_t stream);  // parts == kSweepXcdParts: ypart = 8 * rows doubles of scratch
hipError_t prepare_csr_colsweep();  // once per plan build, on the plan's device: the kernel may use up to 160 KB of dynamic LDS
int sweep_chunks_in_flight(int strip_rows, int asked);  // the G of csr_colsweep<G> a strip height runs with (asked = 1 | 2 | 4 overrides; plan time only)
// entries of 64 K-entry samples of a CSR matrix that gather from distinct 128-byte lines of x (see csr_line_spread)
constexpr int kSpreadSpan = 64 * 1024;
hipError_t launch_csr_line_spread(const int *col_ind, long long nnz, int samples, int *distinct, hipStream_t stream);

// ---- K5: near / far split with a binned two-pass far product (smvp_binned.hip) ----
constexpr int kBinColBits = 14;    // pass A: columns per block, 16384 = 128 KB of x in LDS
constexpr int kBinSlots = 8192;    // pass B: far entries a row block holds at most (64 KB of products in LDS, two workgroups per CU)
constexpr int kBinRowCap = 1024;   // a row with more far entries than this keeps all of them in the near part
constexpr int kBinBucket = kBinSlots - kBinRowCap;  // a row block = the rows whose first far entry falls into one bucket of this many
constexpr int kBinThreads = 1024;  // threads per workgroup, both passes
constexpr int kBinShiftCap = 1024; // cell shifts of a group staged in LDS (the rest are read from memory); pass A's 132 KB leave
                                   // room for one workgroup of the near product's tile kernel (25.8 KB) on the same CU
constexpr int kBinNearBand = 4096; // default: an entry is far when |column - row| exceeds this

// One grouped stream of the far entries (pass A: groups = column blocks, cells = super blocks of rows; pass B: groups =
// row blocks, cells = column blocks): every group's entries one after the other, padded to a multiple of 64; a 16-bit word
// per entry whose top bit marks the first entry of a cell; per cell the distance from the entry's place in this stream to
// its product's place in the bins; per 64 entries the number (inside the group, minus one) of the cell the chunk starts in.
// Groups are padded to whole 256-entry units, each stored interleaved for wide per-lane loads (smvp_binned.hip).
struct BinnedStream {
    int groups = 0, padded = 0, cells = 0;
    unsigned short *word = nullptr;  // padded: 0xffff = padding
    int *chunk = nullptr;            // padded / 64
    int *ptr = nullptr;              // groups + 1 stream offsets (multiples of 256)
    int *shift_ptr = nullptr;        // groups + 1 offsets into shift
    int *shift = nullptr;            // cells
};

// ---- K6: the near part with a row block's window of x in LDS (smvp_near_window.hip) ----
constexpr int kNwRowBlock = 8192;  // rows per workgroup
constexpr int kNwBand = 4096;      // the window reaches this far to either side of the block's rows: 16384 columns = 128 KB of LDS
constexpr int kNwShortCap = 16;    // a row of at most this many near entries is one lane of a slice; a longer one a slice of its own
constexpr int kNwLongCap = 1024;   // long rows a block may hold (their list and their sums sit in LDS): more, and the plan does not suit
struct NearWindow {
    bool on = false;
    int rows = 0, cols = 0, nblocks = 0, n_out = 0;
    long long row0 = 0;                            // global number of the handle's first row: the window of local block b starts at row0 + b * 8192 - band
    long long slots = 0;                           // entries of the streams (64 per step)
    int *wave_ptr = nullptr;                       // nblocks * 16 + 1: first step of every wavefront's run
    int *wave_n1 = nullptr, *wave_n2 = nullptr;    // steps of its short slices / of its long rows
    unsigned short *perm16 = nullptr;              // nblocks * 8192: the block's rows, longest first (0xffff: not a short row)
    double *sval = nullptr;                        // slots
    unsigned short *sword = nullptr;               // slots: column inside the window | valid | last step of the slice
    int *blk_long_ptr = nullptr;                   // nblocks + 1
    unsigned short *long_row16 = nullptr;          // the long rows, block by block
    int *out_row = nullptr, *out_ptr = nullptr, *out_col = nullptr;  // rows with entries outside the window: a CSR of their own
    double *out_val = nullptr;
    size_t plan_bytes = 0;
};
void free_near_window(NearWindow *p);
// the near CSR arrays of a binned plan -> the window plan; `capped[r]` != 0 marks a row that keeps entries outside the band.
// out->on stays false (and SMVP_OK is returned) where the plan does not suit: band > kNwBand, a block with too many long rows
int build_near_window(const int *near_ptr, const int *near_col, const double *near_val, const int *capped, int rows, int cols,
                      int nnz_near, int band, long long row0, NearWindow *out, hipStream_t stream);
// y[r] = the near part's sum for EVERY row r (0 without near entries)
hipError_t launch_near_window(const NearWindow &p, const double *x, double *y, hipStream_t stream);

struct BinnedPlan {
    int band = kBinNearBand;
    long long row0 = 0;  // global number of the handle's first row (a row block of a sharded matrix): "far" is |column - (row0 + row)| > band
    int rows = 0, cols = 0, nnz = 0, nnz_near = 0, nf = 0;  // nf: far entries
    int ncb = 0, nrb = 0, q = 1, nfr = 0, splits = 1;       // column blocks, row blocks, row blocks per super block, far rows
    int slots = kBinSlots, threads_b = kBinThreads;          // pass B: far entries per row block at most (a row's cap is an eighth of it), threads per workgroup
    // near part: its own CSR arrays (rows + 1, nnz_near, nnz_near), run by the tile kernel -- or, where it suits, the window
    // plan `nw` built from them (the arrays are then released)
    int *near_ptr = nullptr, *near_col = nullptr;
    double *near_val = nullptr;
    NearWindow nw;
    // far part
    double *a_val = nullptr;   // pass A stream: values (a.padded)
    BinnedStream a, b;
    double *bins = nullptr;    // nf products, ordered (super block, column block, row, column)
    int *fr_row = nullptr, *fr_ptr = nullptr, *blk_fr = nullptr;  // far rows: row, first far entry (nfr + 1); first far row of each row block (nrb + 1)
    int4 *b_desc = nullptr;    // pass B: two int4 per row block {a, z, k0, k1} {f0, sp, nruns, first far row} (one scalar load instead of a chain)
    unsigned *fr32 = nullptr;  // the far rows as pass B reads them: row - the block's first | first slot << 16, a sentinel per block (nfr + nrb);
                               // fr_row / fr_ptr are released then (kept only where a block's rows span 65536 or more)
    size_t plan_bytes = 0;
};
void free_binned_plan(BinnedPlan *p);
// near / far split of a device-resident CSR matrix and the far part's two streams, built on the device
// near_window: build the near part's window plan where it suits (out->nw.on tells)
int build_binned_plan(const int *d_row_ptr, const int *d_col_ind, const double *d_val, int rows, int cols, int nnz, int band,
                      long long row0, bool near_window, BinnedPlan *out, hipStream_t stream);
// far products into the bins (pass A: needs x only); y[row] += the row's far sum for every row with far entries (pass B:
// after pass A, and after the near product has written y)
// asks the current device for the > 64 KB of dynamic LDS the three kernels of the plan use (once per device; hipSuccess: granted)
hipError_t binned_reserve_lds();
hipError_t near_window_reserve_lds();
hipError_t launch_binned_products(const BinnedPlan &p, const double *x, hipStream_t stream);
hipError_t launch_binned_sums(const BinnedPlan &p, double *y, hipStream_t stream);
// share (0 ... 1) of a CSR matrix's entries with |column - row| > band: what AUTO's choice of the binned plan rests on
int csr_far_share(const int *d_row_ptr, const int *d_col_ind, int rows, int nnz, int band, long long row0, double *share, hipStream_t stream);

// A wavefront's sum without a trip through the LDS crossbar: four data-parallel-primitive steps inside each row of 16 lanes
// (lane i += lane i + 8, + 4, + 2, + 1; lanes beyond the row read 0), then the four row sums -- lanes 0, 16, 32, 48 -- are read
// as scalars and added in a fixed order.  Every lane returns the total.  (__shfl_down on doubles is two ds_bpermute_b32 and a
// wait per step: the long rows of memplus x944 cost the tile kernel 5 % of its time that way.)
template <int N>
__device__ __forceinline__ double dpp_row_shl(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + N, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

//
// ORDER (which forms sum a long row alike): this helper adds lanes i, i+8, i+4, i+2, i+1 inside each row of 16 lanes and then
// (r0 + r1) + (r2 + r3) -- not the order of a __shfl_down tree over 64 lanes (32, 16, 8, 4, 2, 1).  It is what the tile kernel
// (csr_stream_owner, CSR and TJDS flavours), the binned plan's pass B and the near-window kernel use for rows of more than 32
// entries; csr_stream_tiles (STREAM_CARRY), csr_vector_rows and the tile kernel's giant-row path (rows beyond one tile) keep
// shfl_down_sum<64>.  Rows of up to 32 entries are summed left to right by one lane in every form (the serial loop's bits);
// longer rows agree between the two families within the row-normwise bound, not bit for bit.
// Needs wave64 and a full EXEC mask: readlane of lanes 16 / 32 / 48 reads whatever an inactive lane holds -- call it from
// wave-uniform control flow only (every caller loops over rows with a wave-uniform trip count).
__device__ __forceinline__ double wave_sum_dpp(double v)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "wave_sum_dpp: four DPP rows of 16 lanes = one wave64 (gfx9 / CDNA only)"
#endif
    v += dpp_row_shl<8>(v);
    v += dpp_row_shl<4>(v);
    v += dpp_row_shl<2>(v);
    v += dpp_row_shl<1>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * k), __builtin_amdgcn_readlane(lo, 16 * k));
    return (r[0] + r[1]) + (r[2] + r[3]);
}


}  // namespace smvp
