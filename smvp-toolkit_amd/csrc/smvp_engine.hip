// smvp_engine.hip -- device-resident matrices, launch plans and the two
// reference-shaped compute entry points.
//
//   smvp_csr_compute   replaces main-cli.c:325-469
//   smvp_tjds_compute  replaces main-cli.c:734-1162
// The conversion halves live in smvp_convert.cpp (host); this file owns what
// the reference keeps in CSRData / TJDSData (main-cli.c:61-75) once it is in
// HBM, the per-matrix launch plan, and the timed iteration loop
// (main-cli.c:402-420, :1004-1024) with hipEvents in place of clock_gettime.
#include "smvp_common.h"
#include "smvp_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return smvp::fail(SMVP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

// Runs the launches of one call on the handle's device, whatever the caller's current device is.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != device)
            switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope()
    {
        if (switched)
            (void)hipSetDevice(prev);
    }
};

// 32-bit indices like the reference's; the kernels add up to a few thousand to an entry index
constexpr long long kMaxEntries = 2147483647ll - 65536;

int usable_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return smvp::fail(SMVP_ERR_NO_DEVICE, "no HIP device is visible (this engine has no CPU path)");
    if (device < 0 || device >= n)
        return smvp::fail(SMVP_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    return SMVP_OK;
}

template <class T>
int to_device(T **dst, const T *src, size_t count, int mem_kind, bool *owned)
{
    if (mem_kind == SMVP_MEM_DEVICE) {
        *dst = const_cast<T *>(src);
        *owned = false;
        return SMVP_OK;
    }
    *owned = true;
    HIP_TRY(hipMalloc((void **)dst, std::max<size_t>(count, 4) * sizeof(T)));
    if (count)
        HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return SMVP_OK;
}

// Device-resident index arrays are range-checked on the device (host arrays are checked on the host).
int check_device_indices(const int *d_a, long long n, int limit, const char *what)
{
    if (n <= 0)
        return SMVP_OK;
    int *d_bad = nullptr, h_bad = 0;
    HIP_TRY(hipMalloc((void **)&d_bad, sizeof(int)));
    hipError_t e = hipMemset(d_bad, 0, sizeof(int));
    if (e == hipSuccess)
        e = smvp::launch_find_out_of_range(d_a, n, limit, d_bad, nullptr);
    if (e == hipSuccess)
        e = hipMemcpy(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    if (e != hipSuccess)
        return smvp::fail(SMVP_ERR_HIP, "range check of %s failed: %s", what, hipGetErrorString(e));
    if (h_bad)
        return smvp::fail(SMVP_ERR_INVALID, "%s[%d] lies outside [0, %d)", what, h_bad - 1, limit);
    return SMVP_OK;
}

template <class T>
int upload(T **dst, const std::vector<T> &src)
{
    HIP_TRY(hipMalloc((void **)dst, std::max<size_t>(src.size(), 4) * sizeof(T)));
    if (!src.empty())
        HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return SMVP_OK;
}

}  // namespace

// ===========================================================================
// CSR
// ===========================================================================
struct smvp_csr {
    int device = 0;
    int rows = 0, cols = 0, nnz = 0;
    long long row0 = 0;  // global number of this handle's first row (a row block of a sharded matrix; 0 for a whole matrix):
                         // where the diagonal lies -- the binned plan's near / far split and its windows of x go by it
    int *d_row_ptr = nullptr;
    int *d_col_ind = nullptr;
    double *d_val = nullptr;
    bool own_row_ptr = false, own_col_ind = false, own_val = false;
    std::vector<int> h_row_ptr;  // kept for re-planning

    // what an entry of the stream is (smvp_kernels.h kFlavor*): plain CSR, the unit-value form (second phase of the
    // two-phase TJDS product), or a TJDS matrix regrouped by rows (the one-kernel TJDS product); the TJDS flavours
    // borrow these arrays from their smvp_tjds owner
    int flavor = smvp::kFlavorCsr;
    const int *d_pos = nullptr;       // TjdsK: what the kernel reads; TjdsS: the row-major stream the tiles are sorted from
    const int *d_start_pos = nullptr;
    int num_diag = 0;
    // TjdsS: per-tile TJDS-ordered streams and the tiles' overflow entries, owned, rebuilt with the tile plan
    int *d_pos_sorted = nullptr, *d_meta = nullptr, *d_ovf_ptr = nullptr, *d_ovf_k = nullptr;
    double *d_ovf_val = nullptr;  // TjdsS / H: the overflow entries' values (read coalesced by the tile that finishes the row)
    // TjdsS: values of the entries whose val lines scatter over cache_min_tiles tiles or more, kept tile by tile (0: none).
    // A line split over two or three tiles is the edge between neighbouring tiles (they run together on one XCD: an L2
    // hit); from four on its entries belong to unrelated rows.  Measured on memplus x944 (profiles/r03_tjds_forms_measured.txt):
    // none 0.555 ms / 3.66 GB moved, >= 8 tiles 0.461 / 2.92 (20 % of the values cached), >= 4 tiles 0.444 / 2.73 (35 %).
    int cache_min_tiles = 2, cached_total = 0, ovf_total = 0;  // 2: every val line that is not one tile's alone (measured, r04)
    bool unit_operand = false;  // TjdsH as the second phase of the two-phase TJDS product (see TjdsSource)
    int *d_cache_ptr = nullptr;
    double *d_val_cache = nullptr;
    // TjdsH: the 16-bit second word of every entry, the tiles' runs (start_pos of each run's diagonal), each group of 32's run
    unsigned short *d_group_run = nullptr;
    unsigned *d_word32 = nullptr;
    int *d_run_ptr = nullptr, *d_run_tab = nullptr;
    int runs_total = 0;
    int kernel = SMVP_CSR_KERNEL_AUTO;  // resolved: never AUTO once a plan exists
    int lanes_per_row = 64;             // VECTOR
    int vpt = 4;                        // STREAM: entries per thread (tile = 256 * vpt)
    int ntiles = 0;
    int max_row_len = 0;
    int *d_tile_row = nullptr;
    int *d_tile_next = nullptr;   // STREAM: row_ptr[first row of the next tile]
    // STREAM, plain CSR, tiles of 1024 / 2048 entries whose columns all span < 65536: col_ind a second time as 16-bit
    // offsets from the tile's smallest column (10 instead of 12 bytes per entry; csr_stream_owner<., kFlavorCsr16, .>)
    unsigned short *d_col16 = nullptr;
    int *d_col_base = nullptr;
    // STREAM: every row's first entry as a 16-bit offset from the first entry of the tile the row
    // starts in -- what phase 2 reads instead of row_ptr (2 instead of 4 bytes per row)
    unsigned short *d_row_rel = nullptr;
    bool tile_chosen = false;     // the caller named the tile size (smvp_csr_set_kernel param): the plan keeps it
    int *d_carry_row = nullptr;   // STREAM_CARRY
    double *d_carry = nullptr;    // STREAM_CARRY
    // COLSWEEP: the entries a second time, every strip of sweep_rb / 4 rows sorted by column (built on the device);
    // sweep_rb = rows per workgroup (four wavefronts, one strip each)
    int sweep_rb = 0, sweep_per_launch = 0;
    double *d_sweep_part = nullptr;  // COLSWEEP with XCD-private column parts: 8 * rows partial sums
    int sweep_parts = 1;       // COLSWEEP: column parts per strip (1: every row summed in the serial loop's order; 2 / 4: see sweep_param)
    double spread = -2.0;  // share of gathers that pull their own line of x (csr_gather_spread); -2: not measured yet
    long long *d_sweep_ptr = nullptr;
    int *d_sweep_col = nullptr;
    double *d_sweep_val = nullptr;
    unsigned short *d_sweep_row = nullptr;
    // BINNED: the entries a second time, split by |column - row| > band: the near part as the window plan bin.nw (K6) or,
    // where that does not suit, as CSR arrays of its own run by the tile kernel through the nested handle `near`; the far
    // part as the two streams and the bins of smvp_binned.hip
    smvp::BinnedPlan bin;
    smvp_csr *near = nullptr;
    // BINNED with the window plan: pass A (it needs x only) goes onto a stream of its own, ahead of the near part, and
    // the near part's workgroups take the CUs as pass A's last ones leave them
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    double far_share = -2.0;   // share of entries with |column - row| > kBinNearBand; -2: not measured yet
    bool plain_only = false;   // a nested handle: AUTO stays on the tile kernels
    int sweep_g = 0;           // COLSWEEP: chunks in flight per wavefront (fixed when the plan is built)
    double plan_build_ms = 0.0;  // host wall time of the last plan build
};

namespace {

void free_binned(smvp_csr *h)
{
    smvp_csr_destroy(h->near);
    h->near = nullptr;
    smvp::free_binned_plan(&h->bin);
    if (h->side)
        (void)hipStreamDestroy(h->side);
    if (h->ev_fork)
        (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join)
        (void)hipEventDestroy(h->ev_join);
    h->side = nullptr;
    h->ev_fork = h->ev_join = nullptr;
}

void free_sweep_plan(smvp_csr *h)
{
    for (void *p : {(void *)h->d_sweep_ptr, (void *)h->d_sweep_col, (void *)h->d_sweep_val, (void *)h->d_sweep_row, (void *)h->d_sweep_part})
        if (p)
            (void)hipFree(p);
    h->d_sweep_part = nullptr;
    h->d_sweep_ptr = nullptr;
    h->d_sweep_col = nullptr;
    h->d_sweep_val = nullptr;
    h->d_sweep_row = nullptr;
}

// Rows per workgroup of the column sweep (four strips) and how many workgroups start together.  A block of rb rows streams
// rb * (mean row length) entries in column order, 1024 per pass of its workgroup (256 per strip), so its window moves
// 1024 * cols / (rb * mean) columns of x per pass: the taller the block, the slower the window and the better the
// XCD's L2 holds what the resident workgroups gather.  The workgroups start in even generations of at most 256, one per
// CU -- those that start late drift out of the others' window (2.5 M rows, 305 workgroups: 2 x 153 0.883 ms, all at once
// 0.926; 5 M rows: 3 x 204 1.331 against 1.540) -- so a height also decides how full the chip is: 1.25 M rows as 153
// workgroups of 8192 rows keep 60 % of the CUs busy (0.439 ms), as 3 x 204 of 2048 rows 80 % (0.411 ms).  Measured at equal
// fill, 8192 rows run at 1.0, 4096 at 0.86, 2048 at 0.80 of the rate (config 4: 2.25 / 2.62 ms); the height is the one with
// the best fill x rate, never so short that a pass moves the window by more than ~1.3 MB (tools/exp_colsweep.py,
// profiles/r03_colsweep_measured.txt).
// Round 5: the height need not be a power of two.  With generations of at most 256 workgroups the heights 8192 / 4096 / 2048
// leave a rank's block of BASELINE config 4 (1.25 M rows) either 40 % of the CUs idle (153 workgroups of 8192 rows) or three
// generations -- and every generation makes each of the eight L2s pull all of x again (8 x 80 MB: 1.9 GB for 0.56 GB of the
// block's own entries).  Candidates are now also the heights that cut the rows into EXACTLY g x 256
// workgroups, g = 1, 2, ...: 1.25 M rows as 256 workgroups of 4884 rows are one full generation.  The rate of a height in
// between is read off the measured ones on a log scale.
// Round 6: strips as tall as the LDS allows.  Every generation of workgroups makes all eight L2s pull all of x again, and a
// generation covers 256 x (rows per workgroup) rows: taller strips = fewer generations (config 4: two instead of five).  A taller
// strip's column-sorted stream is also denser (1954-row strips: a mean gap of 160 columns, 4883-row strips: 64), so adjacent lanes
// of one gather instruction share a line of x -- one L2 request -- more often: worth a few per cent by itself (the column-parts
// experiment, profiles/r06_colsweep_column_parts.txt).  The 16-bit row word spends 13 bits on the row (turns capped at 7), the
// four strips of a workgroup may fill the CU's LDS (kSweepMaxRb rows = 160 KB of sums), and config 4 whole runs
// 1.98 -> 1.73-1.86 ms as 2 generations of 256 x 19532 rows.
constexpr int kSweepMaxRb = 20480;
double sweep_rate(int rb)
{
    // (20480 rows: round 6 -- strips of up to 5120 rows, the four of a workgroup fill the CU's 160 KB of LDS; config 4 whole:
    // 7816 rows 161.7 G gathers/s, 9768 166.1, 13024 171.2, 19536 177.7 -- profiles/r06_colsweep_tall_strips.txt)
    const struct { double lg, rate; } pts[] = {{10.0, 0.70}, {11.0, 0.80}, {12.0, 0.86}, {13.0, 1.0}, {14.32, 1.10}};
    constexpr int N = (int)(sizeof pts / sizeof pts[0]);
    const double lg = std::log2((double)std::max(rb, 1));
    if (lg <= pts[0].lg)
        return pts[0].rate * std::max(0.25, lg / pts[0].lg);
    for (int i = 0; i + 1 < N; ++i)
        if (lg <= pts[i + 1].lg)
            return pts[i].rate + (pts[i + 1].rate - pts[i].rate) * (lg - pts[i].lg) / (pts[i + 1].lg - pts[i].lg);
    return pts[N - 1].rate;
}

void choose_sweep_shape(int rows, int cols, int nnz, int want_rb, int *rb, int *per_launch)
{
    const double mean = rows > 0 ? std::max(1.0, (double)nnz / rows) : 1.0;
    int floor_rb = 1024;
    while (floor_rb < 8192 && (double)floor_rb * mean * 160.0 < (double)cols)
        floor_rb <<= 1;
    auto generations = [](int nrb) { return (nrb + 255) / 256; };
    std::vector<int> heights;
    for (int hgt : {kSweepMaxRb, 16384, 8192, 4096, 2048, 1024})
        if (hgt >= floor_rb)
            heights.push_back(hgt);
    // ... and the heights of whole generations; those may be half as tall as the floor (a full chip outweighs the faster window:
    // a 312 K-row chunk of config 4 as 256 x 1224 rows 0.0975 ms, as 153 x 2048 rows 0.134; profiles/r05_colsweep_heights.txt)
    for (int g = 1; g <= 16 && rows > 0; ++g) {
        const long long want = ((long long)rows + 256ll * g - 1) / (256ll * g);  // rows per workgroup for g full generations
        const int h4 = (int)std::min<long long>(kSweepMaxRb, (want + 3) / 4 * 4);  // four strips per workgroup
        if (h4 >= std::max(1024, floor_rb / 2))
            heights.push_back(h4);
    }
    int r = floor_rb;
    double best = -1.0;
    int best_gen = 1 << 30;
    for (int hgt : heights) {
        const int nrb = (rows + hgt - 1) / hgt;
        const int gen = std::max(1, generations(nrb));
        const double score = (double)nrb / (gen * 256.0) * sweep_rate(hgt);
        // (a tie within 1 % goes to fewer generations: each costs 8 x the operand in L2 fills)
        if (score > best * 1.01 || (score > best * 0.99 && gen < best_gen)) {
            best = std::max(best, score);
            best_gen = gen;
            r = hgt;
        }
    }
    if (want_rb > 0)
        r = want_rb;
    const int nrb = (rows + r - 1) / r;
    const int gen = std::max(1, generations(nrb));  // no rows: one (empty) generation
    *rb = r;
    *per_launch = std::max(1, (nrb + gen - 1) / gen);
}

// The column sweep's kernel parameter: rows per workgroup (a multiple of 4, 0 = chosen) in the low 24 bits and, above them, log2 of
// the COLUMN PARTS per strip (round 6).  With parts = 1 a workgroup's four wavefronts own four strips; with parts = 2 / 4 they own
// two strips / one strip whose stream is cut into column halves / quarters, one wavefront and one array of partial sums each:
// the strip is 2x / 4x as tall for the same rows per workgroup -- a block of rows too short to fill the LDS with four strips
// (one rank's eighth of config 4: 256 workgroups x 4884 rows = four strips of 1221) gets the denser streams of a tall strip
// (one strip of 4884) -- and a row's sum is its partial sums added part by part: reproducible, inside the rounding bound, no
// longer the serial loop's bits.  Never chosen by AUTO (which keeps the serial order); SMVP_CSR_SWEEP_PARTS(rb, parts) asks for it.
// Above the parts (bits 26-27): chunks in flight per wavefront for experiments -- 0 = the rule (sweep_chunks_in_flight), 1 / 2 / 3 = one /
// two / four (what the environment switch SMVP_SWEEP_G used to say; tools/exp_colsweep.py --g).
constexpr int kSweepPartsShift = 24, kSweepChunksShift = 26;
inline int sweep_param_rb(int param) { return param & ((1 << kSweepPartsShift) - 1); }
inline int sweep_param_parts(int param) { return 1 << ((param >> kSweepPartsShift) & 3); }  // (8 = one part per XCD, below)
inline int sweep_param_chunks(int param) { const int c = (param >> kSweepChunksShift) & 3; return c == 3 ? 4 : c; }
bool sweep_param_ok(int param)
{
    const int rb = sweep_param_rb(param), parts = sweep_param_parts(param);
    if (param < 0 || (param >> (kSweepChunksShift + 2)) != 0)
        return false;
    if (rb == 0)
        return true;  // (height chosen; the parts are honoured where the chosen height leaves room for them)
    if (parts == smvp::kSweepXcdParts)  // XCD-private parts: the workgroup's four strips are whole strips of rb / 4 rows
        return rb >= 256 && rb % 4 == 0 && rb <= kSweepMaxRb;
    return rb >= 256 && rb % 4 == 0 && (long long)rb * parts <= kSweepMaxRb;
}

int build_sweep_plan(smvp_csr *h, int want)
{
    free_sweep_plan(h);
    const int want_rb = sweep_param_rb(want);
    h->sweep_parts = sweep_param_parts(want);
    const bool xcd = h->sweep_parts == smvp::kSweepXcdParts;
    if (xcd) {
        // XCD-private column parts (SMVP_CSR_SWEEP_PARTS(rb, 8)): a generation of 256 workgroups = 32 row groups x 8 parts; the height is the
        // caller's, or the one that cuts the rows into whole generations with the tallest strips the LDS takes
        int rb = want_rb;
        if (rb == 0) {
            const long long per_gen_max = 32ll * kSweepMaxRb;
            const long long gens = std::max<long long>(1, ((long long)h->rows + per_gen_max - 1) / per_gen_max);
            rb = (int)std::min<long long>(kSweepMaxRb, std::max<long long>(256, (((long long)h->rows + 32 * gens - 1) / (32 * gens) + 3) / 4 * 4));
        }
        h->sweep_rb = rb;
        h->sweep_per_launch = 256;
    } else {
        choose_sweep_shape(h->rows, h->cols, h->nnz, want_rb, &h->sweep_rb, &h->sweep_per_launch);
        while (h->sweep_parts > 1 && (long long)h->sweep_rb * h->sweep_parts > kSweepMaxRb)
            h->sweep_parts >>= 1;  // (a chosen height may leave room for fewer parts than asked)
    }
    const int strip_rows = xcd ? h->sweep_rb / smvp::kSweepWaves : h->sweep_rb * h->sweep_parts / smvp::kSweepWaves;
    h->sweep_g = smvp::sweep_chunks_in_flight(strip_rows, sweep_param_chunks(want));
    if (const hipError_t pe = smvp::prepare_csr_colsweep(); pe != hipSuccess)
        return smvp::fail(SMVP_ERR_HIP, "the column sweep cannot have its LDS: %s", hipGetErrorString(pe));
    const int nstrips = (h->rows + strip_rows - 1) / strip_rows;
    const size_t n = (size_t)std::max(h->nnz, 4);
    if (hipMalloc((void **)&h->d_sweep_ptr, ((size_t)nstrips * h->sweep_parts + 2) * sizeof(long long)) != hipSuccess ||
        hipMalloc((void **)&h->d_sweep_col, n * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&h->d_sweep_val, n * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&h->d_sweep_row, n * sizeof(unsigned short)) != hipSuccess ||
        (xcd && hipMalloc((void **)&h->d_sweep_part, sizeof(double) * smvp::kSweepXcdParts * (size_t)std::max(h->rows, 1)) != hipSuccess))
        return smvp::fail(SMVP_ERR_ALLOC, "cannot allocate the column-sweep plan (%d entries)", h->nnz);
    return smvp::build_colsweep_plan(h->d_row_ptr, h->d_col_ind, h->d_val, h->rows, h->cols, h->nnz, strip_rows, h->sweep_parts, smvp::kSweepChunk,
                                     smvp::kSweepRowBits, smvp::kSweepTurnCap, h->d_sweep_ptr, h->d_sweep_col, h->d_sweep_val, h->d_sweep_row, nullptr);
}

// Share of the gathers of a (large) CSR matrix that pull their own 128-byte line of x through the L2, estimated on
// kSpreadSamples runs of 64 K consecutive entries -- one XCD's turn of the tile kernel -- by linear counting
// (smvp_kernels.hip: csr_line_spread): distinct lines / entries, 0 ... 1.  -1: too small to sample, or the probe failed.
constexpr int kSpreadSamples = 64;
double csr_gather_spread(const smvp_csr *h)
{
    if ((long long)h->nnz < (long long)kSpreadSamples * smvp::kSpreadSpan || !h->d_col_ind)
        return -1.0;
    int *d_bits = nullptr;
    if (hipMalloc((void **)&d_bits, sizeof(int) * kSpreadSamples) != hipSuccess)
        return -1.0;
    int bits[kSpreadSamples];
    hipError_t e = smvp::launch_csr_line_spread(h->d_col_ind, h->nnz, kSpreadSamples, d_bits, nullptr);
    if (e == hipSuccess)
        e = hipMemcpy(bits, d_bits, sizeof bits, hipMemcpyDeviceToHost);
    (void)hipFree(d_bits);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return -1.0;
    }
    const double m = 512.0 * 1024.0;  // bits in the map
    double lines = 0.0;
    for (int b : bits)
        lines += -m * std::log(std::max(1.0 - b / m, 1.0 / m));  // linear counting: n = -m ln(share of clear bits)
    return std::min(1.0, lines / ((double)kSpreadSamples * smvp::kSpreadSpan));
}

// AUTO picks the column sweep when the tile kernel would be bound by L2-miss gathers and the sweep's window can hold:
//  * the operand is at least twice an XCD's 4 MB L2 (else the tile kernel's gathers hit anyway; measured on uniform
//    rows of 32: 8 MB operand 0.340 -> 0.205 ms, 16 MB 0.907 -> 0.403, 32 MB 2.20 -> 0.81, 80 MB 6.2 -> 2.45), rows
//    of at least 4 entries on average;
//  * most gathers pull their own line: spread >= kSweepMinSpread.  Tile kernel: spread * nnz / 54 G lines/s (measured:
//    config 4, spread 0.95, 5.98 ms; the SURVEY 8(d) random model, 0.42, 1.03 ms); sweep: nnz / 133 G/s;
//  * a workgroup's 8192 rows hold enough entries that a pass moves its window by about 1 MB at most
//    (choose_sweep_shape) -- the random model's 7 entries per row over 16.7 M columns do not, and the sweep loses there
//    (profiles/r02_colsweep_measured.txt);
//  * no row is so long that its strip becomes the critical path; the matrix is worth a second copy (>= 4 M entries).
constexpr double kSweepMinSpread = 0.6;
bool sweep_suits(smvp_csr *h)
{
    if (h->plain_only || h->flavor != smvp::kFlavorCsr || h->nnz < 4 * 1024 * 1024 || h->rows < 4096)
        return false;
    const double mean = (double)h->nnz / h->rows;
    if (mean < 4.0 || (double)h->cols * 8.0 < 8.0 * 1024 * 1024 || 8192.0 * mean * 160.0 < (double)h->cols ||
        (double)h->max_row_len > 256.0 * mean)
        return false;
    if (h->spread < -1.5)
        h->spread = csr_gather_spread(h);
    return h->spread >= kSweepMinSpread;
}

// AUTO picks the binned plan (near part on the tile kernel, far part in two LDS-binned passes: smvp_binned.hip) where the
// tile kernel is held back by gathers that miss the L2 and the column sweep does not fit:
//  * a large matrix (>= 4 M entries) over an operand of at least 16 MB;
//  * a good share of the gathers pull their own line of x: spread >= kBinnedMinSpread (the SURVEY 8(d) random model: 0.40,
//    tile kernel 1.03 ms; kron(I, memplus), pwt x459, a narrow band: < 0.1, which stay on the tile kernel);
//  * and those gathers are what the split removes: at least kBinnedMinFarShare of the entries lie further than
//    kBinNearBand from the diagonal (the model: 0.39).  A matrix whose scattered gathers are close to the diagonal
//    gains nothing from the split.
// The column sweep is asked first (spread >= 0.6 and rows long enough for its window: BASELINE config 4).
constexpr double kBinnedMinSpread = 0.2, kBinnedMinFarShare = 0.1;
bool binned_suits(smvp_csr *h)
{
    if (h->plain_only || h->flavor != smvp::kFlavorCsr || h->nnz < 4 * 1024 * 1024 || h->rows < 4096 ||
        (double)h->cols * 8.0 < 16.0 * 1024 * 1024)
        return false;
    if (h->spread < -1.5)
        h->spread = csr_gather_spread(h);
    if (h->spread < kBinnedMinSpread)
        return false;
    if (h->far_share < -1.5) {
        double share = -1.0;
        if (smvp::csr_far_share(h->d_row_ptr, h->d_col_ind, h->rows, h->nnz, smvp::kBinNearBand, h->row0, &share, nullptr) != SMVP_OK) {
            (void)hipGetLastError();
            share = -1.0;
        }
        h->far_share = share;
    }
    return h->far_share >= kBinnedMinFarShare;
}

void free_stream_plan(smvp_csr *h)
{
    if (h->d_tile_row)
        (void)hipFree(h->d_tile_row);
    if (h->d_carry_row)
        (void)hipFree(h->d_carry_row);
    if (h->d_carry)
        (void)hipFree(h->d_carry);
    if (h->d_tile_next)
        (void)hipFree(h->d_tile_next);
    if (h->d_col16)
        (void)hipFree(h->d_col16);
    if (h->d_col_base)
        (void)hipFree(h->d_col_base);
    if (h->d_row_rel)
        (void)hipFree(h->d_row_rel);
    h->d_col16 = nullptr;
    h->d_col_base = nullptr;
    h->d_row_rel = nullptr;
    for (void *p : {(void *)h->d_pos_sorted, (void *)h->d_meta, (void *)h->d_ovf_ptr, (void *)h->d_ovf_val, (void *)h->d_ovf_k,
                    (void *)h->d_cache_ptr, (void *)h->d_val_cache, (void *)h->d_word32, (void *)h->d_group_run,
                    (void *)h->d_run_ptr, (void *)h->d_run_tab})
        if (p)
            (void)hipFree(p);
    h->d_pos_sorted = h->d_meta = h->d_ovf_ptr = h->d_ovf_k = h->d_cache_ptr = nullptr;
    h->d_ovf_val = nullptr;
    h->d_run_ptr = h->d_run_tab = nullptr;
    h->d_group_run = nullptr;
    h->d_word32 = nullptr;
    h->d_val_cache = nullptr;
    h->cached_total = h->runs_total = 0;
    h->d_tile_row = h->d_carry_row = h->d_tile_next = nullptr;
    h->d_carry = nullptr;
    h->ntiles = 0;
}

// tile_row[b]  = first row whose first entry lies at or after b*TILE
// carry_row[b] = row that the entries in front of that row belong to, or -1
// 16-bit column offsets for the plain CSR tile kernel where every tile's columns are close together (banded and
// block-structured matrices): 10 instead of 12 bytes per entry.  Tried for 2048-entry tiles first on large matrices --
// with the narrower stream the larger tile wins (memplus x944: 0.2965 ms against 0.3131 with 1024-entry tiles; 32-bit
// columns: 0.319 / 0.325) -- unless the caller named the tile size.  Leaves h->vpt at the tile size that fits, or the
// handle without offsets.
void try_column_offsets(smvp_csr *h)
{
    if (h->flavor != smvp::kFlavorCsr || h->kernel != SMVP_CSR_KERNEL_STREAM || h->vpt < 4 || h->nnz <= 0 ||
        smvp::option("csr_col16", 1) == 0)  // (plan option: 0 keeps the 32-bit columns)
        return;
    std::vector<int> tiles;
    if (!h->tile_chosen && h->nnz >= 12 * 1024 * 1024)  // memplus x59 / x118 / x236 (7.4 / 14.9 / 29.8 M entries), 1024- against
        tiles.push_back(2048);                           // 2048-entry tiles: 0.0220 / 0.0364 / 0.0851 against 0.0219 / 0.0345 / 0.0787 ms
    tiles.push_back(smvp::kStreamBlock * h->vpt);
    const size_t max_tiles = ((size_t)h->nnz + 1023) / 1024 + 1;
    if (hipMalloc((void **)&h->d_col_base, max_tiles * sizeof(int)) == hipSuccess &&
        hipMalloc((void **)&h->d_col16, ((size_t)h->nnz + 8) * sizeof(unsigned short)) == hipSuccess) {
        for (int tile : tiles) {
            int fits = 0;
            if (smvp::build_column_offsets(h->d_col_ind, h->nnz, tile, h->d_col_base, h->d_col16, &fits, nullptr) == SMVP_OK && fits) {
                h->vpt = tile / smvp::kStreamBlock;
                return;
            }
        }
    }
    (void)hipGetLastError();  // no second copy: the kernel reads col_ind itself
    if (h->d_col16)
        (void)hipFree(h->d_col16);
    if (h->d_col_base)
        (void)hipFree(h->d_col_base);
    h->d_col16 = nullptr;
    h->d_col_base = nullptr;
}

int build_stream_plan(smvp_csr *h)
{
    free_stream_plan(h);
    try_column_offsets(h);
    const int tile = smvp::kStreamBlock * h->vpt;
    const long long nnz = h->nnz;
    const int ntiles = (int)std::max<long long>(1, (nnz + tile - 1) / tile);
    std::vector<int> tile_row((size_t)ntiles + 1), carry_row((size_t)ntiles), tile_next((size_t)ntiles);
    const int *rp = h->h_row_ptr.data();
    int r = 0;
    for (int b = 0; b < ntiles; ++b) {
        const long long s = (long long)b * tile;
        const long long e = std::min(s + tile, nnz);
        while (r < h->rows && rp[r] < s)
            ++r;
        tile_row[(size_t)b] = r;
        const long long first = r < h->rows ? rp[r] : nnz;
        carry_row[(size_t)b] = (std::min(first, e) > s) ? r - 1 : -1;
    }
    tile_row[(size_t)ntiles] = h->rows;
    for (int b = 0; b < ntiles; ++b)
        tile_next[(size_t)b] = rp[tile_row[(size_t)b + 1]];
    if (int rc = upload(&h->d_tile_row, tile_row))
        return rc;
    if (h->kernel == SMVP_CSR_KERNEL_STREAM) {
        if (int rc = upload(&h->d_tile_next, tile_next))
            return rc;
        if (h->rows > 0 && smvp::option("csr_rowrel", 1) != 0) {  // (plan option: 0 = keep row_ptr)
            std::vector<unsigned short> rel((size_t)h->rows);
            for (int b = 0; b < ntiles; ++b) {
                const long long s0 = (long long)b * tile;
                for (int rr = tile_row[(size_t)b]; rr < tile_row[(size_t)b + 1]; ++rr)
                    rel[(size_t)rr] = (unsigned short)(rp[rr] - s0);  // 0 ... tile (tile: trailing rows without entries)
            }
            if (int rc = upload(&h->d_row_rel, rel))
                return rc;
        }
    } else {
        if (int rc = upload(&h->d_carry_row, carry_row))
            return rc;
        HIP_TRY(hipMalloc((void **)&h->d_carry, std::max(ntiles, 1) * sizeof(double)));
        HIP_TRY(hipMemset(h->d_carry, 0, std::max(ntiles, 1) * sizeof(double)));
    }
    h->ntiles = ntiles;
    if (h->flavor == smvp::kFlavorTjdsS || h->flavor == smvp::kFlavorTjdsH) {
        // every tile's entries in TJDS order + what each tile reads past its end, in row order
        std::vector<int> ovf_ptr((size_t)ntiles + 1, 0);
        for (int b = 0; b < ntiles; ++b) {
            const long long e = std::min((long long)(b + 1) * tile, nnz);
            const bool owns = tile_row[(size_t)b] != tile_row[(size_t)b + 1];
            ovf_ptr[(size_t)b + 1] = ovf_ptr[(size_t)b] + (owns ? (int)(tile_next[(size_t)b] - e) : 0);
        }
        const int total = ovf_ptr[(size_t)ntiles];
        h->ovf_total = total;
        if (int rc = upload(&h->d_ovf_ptr, ovf_ptr))
            return rc;
        const size_t n = (size_t)std::max(h->nnz, 4), m = (size_t)std::max(total, 4);
        if (hipMalloc((void **)&h->d_ovf_val, m * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&h->d_ovf_k, m * sizeof(int)) != hipSuccess ||
            hipMalloc((void **)&h->d_cache_ptr, ((size_t)ntiles + 2) * sizeof(int)) != hipSuccess)
            return smvp::fail(SMVP_ERR_ALLOC, "cannot allocate the tile-ordered TJDS streams");
        if (h->flavor == smvp::kFlavorTjdsH) {
            if (hipMalloc((void **)&h->d_word32, n * sizeof(unsigned)) != hipSuccess ||
                hipMalloc((void **)&h->d_run_ptr, ((size_t)ntiles + 2) * sizeof(int)) != hipSuccess)
                return smvp::fail(SMVP_ERR_ALLOC, "cannot allocate the tile-ordered TJDS streams");
            if (int rc = smvp::build_tile_half_streams(h->d_pos, h->nnz, tile, h->d_start_pos, h->num_diag, h->d_val,
                                                       h->cache_min_tiles, h->d_word32, h->d_cache_ptr,
                                                       h->d_run_ptr, &h->d_val_cache, &h->d_run_tab, &h->d_group_run,
                                                       &h->cached_total, &h->runs_total, nullptr))
                return rc;
        } else {
            if (hipMalloc((void **)&h->d_pos_sorted, n * sizeof(int)) != hipSuccess ||
                hipMalloc((void **)&h->d_meta, n * sizeof(int)) != hipSuccess)
                return smvp::fail(SMVP_ERR_ALLOC, "cannot allocate the tile-ordered TJDS streams");
            if (int rc = smvp::sort_tile_windows(h->d_pos, h->nnz, tile, h->d_start_pos, h->num_diag, smvp::kSlotBits, h->d_val,
                                                 h->cache_min_tiles, h->d_pos_sorted, h->d_meta, h->d_cache_ptr, &h->d_val_cache,
                                                 &h->cached_total, nullptr))
                return rc;
        }
        if (int rc = smvp::build_tile_overflow(h->d_pos, h->d_ovf_ptr, total, ntiles, tile, h->nnz, h->d_start_pos,
                                               h->num_diag, h->d_val, h->d_ovf_val, h->d_ovf_k, h->unit_operand ? 1 : 0, nullptr))
            return rc;
    }
    return SMVP_OK;
}

int pow2_at_least(double v)
{
    int p = 2;
    while (p < 64 && p < v)
        p <<= 1;
    return p;
}

// AUTO: fixed-nnz tiles are insensitive to row-length skew and keep short rows
// at full lane use -- the owner-completes form unless some row is so long that
// one workgroup finishing it alone would be the critical path (then the carry
// form, which spreads a row over its tiles).  The wavefront-per-row kernel is
// never picked: measured on uniform rows of 64 / 128 / 400 entries it runs at
// 0.98 / 0.86 / 0.87 of the tile kernel's rate, and far below it on skewed rows.
constexpr int kOwnerMaxRow = 16 * 1024;

// false: `param` is no tile size the resolved kernel is compiled for (nothing is changed then)
bool choose_csr_kernel(smvp_csr *h, int kernel, int param)
{
    const double mean = h->rows > 0 ? (double)h->nnz / h->rows : 0.0;
    if (kernel == SMVP_CSR_KERNEL_AUTO)
        kernel = h->max_row_len > kOwnerMaxRow ? SMVP_CSR_KERNEL_STREAM_CARRY
                 : sweep_suits(h)              ? SMVP_CSR_KERNEL_COLSWEEP
                 : binned_suits(h)             ? SMVP_CSR_KERNEL_BINNED
                                               : SMVP_CSR_KERNEL_STREAM;
    if (kernel == SMVP_CSR_KERNEL_STREAM && param != 0 && param != 256 && param != 1024 && param != 2048)
        return false;
    if (kernel == SMVP_CSR_KERNEL_STREAM_CARRY && param != 0 && param != 1024 && param != 2048)
        return false;
    if (kernel == SMVP_CSR_KERNEL_COLSWEEP && !sweep_param_ok(param))
        return false;
    if (kernel == SMVP_CSR_KERNEL_BINNED && param < 0)
        return false;
    h->kernel = kernel;
    if (kernel == SMVP_CSR_KERNEL_COLSWEEP || kernel == SMVP_CSR_KERNEL_BINNED)
        return true;
    if (kernel == SMVP_CSR_KERNEL_VECTOR) {
        h->lanes_per_row = param > 0 ? param : pow2_at_least(mean);
    } else {
        int tile = param > 0 ? param : 0;
        h->tile_chosen = param > 0;
        if (tile == 0) {  // 1024 measured 1-4 % ahead of 2048 on memplus x944 and pwt x459; 256-entry tiles when
            tile = (kernel == SMVP_CSR_KERNEL_STREAM && h->nnz < 512 * 1024) ? 256 : 1024;  // 1024 would leave CUs idle
            if ((h->flavor == smvp::kFlavorTjdsS || h->flavor == smvp::kFlavorTjdsH) && tile == 1024 && h->nnz >= 12 * 1024 * 1024)
                tile = 2048;  // more entries per val line inside a tile.  With the value cache and the 16-bit second word (round 3),
                              // memplus x59 / x118 / x236 / x472 (7.4 / 14.9 / 29.8 / 59.5 M entries), 1024 against 2048:
                              // 0.0285 / 0.0530 / 0.1157 / 0.2327 against 0.0280 / 0.0493 / 0.1070 / 0.2151 ms
        }
        h->vpt = tile / smvp::kStreamBlock;
    }
    return true;
}

}  // namespace

// what a TJDS flavour borrows from its smvp_tjds owner (set before the tile plan is built)
struct TjdsSource {
    const int *pos = nullptr;
    const int *start_pos = nullptr;
    int num_diag = 0;
    // the two-phase product's second phase: `val` holds the products of the first phase (written anew before every launch: no value
    // cache, overflow entries by position), the operand is the unit vector (no x gather)
    bool unit_operand = false;
};

static int build_binned(smvp_csr *h, int band);
static double wall_ms();

static int csr_create_impl(smvp_csr_t **out, int device, int rows, int cols, int nnz,
                           const int *row_ptr, const int *col_ind, const double *val,
                           int mem_kind, const int *host_row_ptr, int flavor, const TjdsSource *src = nullptr,
                           bool plain_only = false, long long first_row = 0)
{
    const bool plain = flavor == smvp::kFlavorCsr;
    if (!out || rows < 0 || cols < 0 || nnz < 0 || !row_ptr ||
        (nnz > 0 && ((!col_ind && flavor != smvp::kFlavorTjdsS && flavor != smvp::kFlavorTjdsH) || !val)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_create: bad argument");
    if (mem_kind != SMVP_MEM_HOST && mem_kind != SMVP_MEM_DEVICE)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_create: bad mem_kind");
    if (nnz > kMaxEntries)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "smvp_csr_create: %d entries: shard blocks this large by rows", nnz);
    if (int rc = usable_device(device))
        return rc;
    DeviceScope on(device);

    smvp_csr *h = new smvp_csr;
    h->device = device;
    h->flavor = flavor;
    h->plain_only = plain_only;
    h->row0 = first_row;
    if (src) {
        h->d_pos = src->pos, h->d_start_pos = src->start_pos;
        h->num_diag = src->num_diag;
        h->unit_operand = src->unit_operand;
        if (h->unit_operand)
            h->cache_min_tiles = 0;
    }
    h->rows = rows, h->cols = cols, h->nnz = nnz;
    h->h_row_ptr.resize((size_t)rows + 1);
    int rc = SMVP_OK;
    if (mem_kind == SMVP_MEM_HOST || host_row_ptr) {
        memcpy(h->h_row_ptr.data(), mem_kind == SMVP_MEM_HOST ? row_ptr : host_row_ptr, sizeof(int) * ((size_t)rows + 1));
    } else if (hipMemcpy(h->h_row_ptr.data(), row_ptr, sizeof(int) * ((size_t)rows + 1), hipMemcpyDeviceToHost) != hipSuccess) {
        rc = smvp::fail(SMVP_ERR_HIP, "smvp_csr_create: cannot read row_ptr back from the device");
    }
    // The kernels index with these; a malformed row_ptr would read out of bounds.
    if (rc == SMVP_OK) {
        const std::vector<int> &rp = h->h_row_ptr;
        bool ok = rp[0] == 0 && rp[(size_t)rows] == nnz;
        for (int r = 0; r < rows && ok; ++r) {
            ok = rp[(size_t)r] <= rp[(size_t)r + 1];
            h->max_row_len = std::max(h->max_row_len, rp[(size_t)r + 1] - rp[(size_t)r]);
        }
        if (!ok)
            rc = smvp::fail(SMVP_ERR_INVALID, "smvp_csr_create: row_ptr is not a non-decreasing 0..nnz sequence");
    }
    if (rc == SMVP_OK && mem_kind == SMVP_MEM_HOST) {
        for (int j = 0; j < nnz; ++j)
            if (col_ind[j] < 0 || col_ind[j] >= cols) {
                rc = smvp::fail(SMVP_ERR_INVALID, "smvp_csr_create: col_ind[%d] = %d outside [0, %d)", j, col_ind[j], cols);
                break;
            }
    }
    if (rc == SMVP_OK && mem_kind == SMVP_MEM_DEVICE &&
        (((uintptr_t)col_ind | (plain ? (uintptr_t)val : 0)) & 15u) != 0)
        rc = smvp::fail(SMVP_ERR_INVALID, "smvp_csr_create: adopted device arrays must be 16-byte aligned");
    if (rc == SMVP_OK && mem_kind == SMVP_MEM_DEVICE && col_ind)
        rc = check_device_indices(col_ind, nnz, cols, "smvp_csr_create: col_ind");
    if (rc == SMVP_OK)
        rc = to_device(&h->d_row_ptr, row_ptr, (size_t)rows + 1, mem_kind, &h->own_row_ptr);
    if (rc == SMVP_OK && col_ind)
        rc = to_device(&h->d_col_ind, col_ind, (size_t)nnz, mem_kind, &h->own_col_ind);
    if (rc == SMVP_OK)
        rc = to_device(&h->d_val, val, (size_t)nnz, mem_kind, &h->own_val);
    if (rc == SMVP_OK) {
        const double t0 = wall_ms();
        // every flavour but plain CSR exists for the owner-completes kernel only
        choose_csr_kernel(h, plain ? SMVP_CSR_KERNEL_AUTO : SMVP_CSR_KERNEL_STREAM, 0);
        if ((h->kernel == SMVP_CSR_KERNEL_COLSWEEP && build_sweep_plan(h, 0) != SMVP_OK) ||
            (h->kernel == SMVP_CSR_KERNEL_BINNED && build_binned(h, 0) != SMVP_OK)) {
            // AUTO's second copy of the entries did not fit: the tile kernel needs none
            (void)hipGetLastError();
            free_sweep_plan(h);
            free_binned(h);
            h->spread = -1.0;
            choose_csr_kernel(h, SMVP_CSR_KERNEL_STREAM, 0);
        }
        if (h->kernel != SMVP_CSR_KERNEL_VECTOR && h->kernel != SMVP_CSR_KERNEL_COLSWEEP && h->kernel != SMVP_CSR_KERNEL_BINNED)
            rc = build_stream_plan(h);
        h->plan_build_ms = wall_ms() - t0;
    }
    if (rc != SMVP_OK) {
        smvp_csr_destroy(h);
        return rc;
    }
    *out = h;
    return SMVP_OK;
}

extern "C" int smvp_csr_create(smvp_csr_t **out, int device, int rows, int cols, int nnz,
                               const int *row_ptr, const int *col_ind, const double *val,
                               int mem_kind, const int *host_row_ptr)
{
    return csr_create_impl(out, device, rows, cols, nnz, row_ptr, col_ind, val, mem_kind, host_row_ptr, smvp::kFlavorCsr);
}

// A row block [first_row, first_row + rows) of a larger matrix (what a rank of a sharded product holds): the same handle,
// but plans that go by the distance from the diagonal -- the binned plan's near / far split, its windows of x, AUTO's
// far share -- take the diagonal where it really lies.  Without this every block beyond the first 4096 rows looked all far.
extern "C" int smvp_csr_create_block(smvp_csr_t **out, int device, int rows, int cols, int nnz,
                                     const int *row_ptr, const int *col_ind, const double *val,
                                     int mem_kind, const int *host_row_ptr, long long first_row)
{
    if (first_row < 0)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_create_block: first_row < 0");
    return csr_create_impl(out, device, rows, cols, nnz, row_ptr, col_ind, val, mem_kind, host_row_ptr, smvp::kFlavorCsr, nullptr,
                           false, first_row);
}

// share (0 ... 1) of the entries further than the binned plan's default band from the diagonal: what AUTO's choice rests on
extern "C" int smvp_csr_far_share(smvp_csr_t *h, double *share)
{
    if (!h || !share)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_far_share: bad argument");
    DeviceScope on(h->device);
    if (h->far_share < -1.5) {
        double s = -1.0;
        if (h->flavor != smvp::kFlavorCsr ||
            smvp::csr_far_share(h->d_row_ptr, h->d_col_ind, h->rows, h->nnz, smvp::kBinNearBand, h->row0, &s, nullptr) != SMVP_OK) {
            (void)hipGetLastError();
            s = -1.0;
        }
        h->far_share = s;
    }
    *share = h->far_share;
    return SMVP_OK;
}

static double wall_ms()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

// The binned plan: near / far split on the device; the near part as the window plan (smvp_near_window.hip) where that suits,
// else behind a nested handle that stays on the tile kernels.
static int build_binned(smvp_csr *h, int band)
{
    free_binned(h);
    // both passes (and K6) ask for more than 64 KB of dynamic LDS: a device that refuses makes the PLAN fail here -- AUTO then
    // falls back to the tile kernel (csr_create_impl) -- instead of every later product
    if (hipError_t le = smvp::binned_reserve_lds(); le != hipSuccess) {
        (void)hipGetLastError();
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "the binned plan needs 154 KB of LDS per workgroup: %s", hipGetErrorString(le));
    }
    const bool window = smvp::option("binned_near", 0) == 0;  // (plan option: 1 keeps the near part on the tile kernel)
    if (int rc = smvp::build_binned_plan(h->d_row_ptr, h->d_col_ind, h->d_val, h->rows, h->cols, h->nnz, band, h->row0, window, &h->bin, nullptr))
        return rc;
    if (h->bin.nw.on) {
        if (smvp::option("binned_overlap", 1) != 0 && h->bin.nf > 0) {  // (plan option: 0 = pass A behind the near part, one stream)
            HIP_TRY(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
        return SMVP_OK;
    }
    return csr_create_impl(&h->near, h->device, h->rows, h->cols, h->bin.nnz_near, h->bin.near_ptr, h->bin.near_col, h->bin.near_val,
                           SMVP_MEM_DEVICE, nullptr, smvp::kFlavorCsr, nullptr, true, h->row0);
}

extern "C" int smvp_csr_set_kernel(smvp_csr_t *h, int kernel, int param)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    if (h->flavor != smvp::kFlavorCsr && kernel != SMVP_CSR_KERNEL_STREAM)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "this matrix flavour runs on the stream kernel only");
    if (kernel < SMVP_CSR_KERNEL_AUTO || kernel > SMVP_CSR_KERNEL_BINNED)
        return smvp::fail(SMVP_ERR_INVALID, "unknown CSR kernel %d", kernel);
    if (kernel == SMVP_CSR_KERNEL_VECTOR && param != 0 &&
        (param < 2 || param > 64 || (param & (param - 1)) != 0))
        return smvp::fail(SMVP_ERR_INVALID, "lanes per row must be a power of two in [2, 64]");
    DeviceScope on(h->device);
    if (!choose_csr_kernel(h, kernel, param))
        return smvp::fail(SMVP_ERR_INVALID, "entries per tile must be 256 (stream only), 1024 or 2048 for the kernel "
                                            "this matrix resolves to (column sweep: 256 ... 20480 rows per block, a multiple of 4; binned: the near band, >= 0)");
    const double t0 = wall_ms();
    free_sweep_plan(h);
    free_binned(h);
    int rc = SMVP_OK;
    if (h->kernel == SMVP_CSR_KERNEL_COLSWEEP || h->kernel == SMVP_CSR_KERNEL_BINNED) {
        free_stream_plan(h);
        rc = h->kernel == SMVP_CSR_KERNEL_COLSWEEP ? build_sweep_plan(h, param) : build_binned(h, param);
        if (rc != SMVP_OK) {
            // the second copy of the entries did not fit (or could not be built): back to the tile kernel, which needs
            // none, so that the handle stays usable -- the error is still reported
            const std::string why = smvp_last_error();
            (void)hipGetLastError();
            free_sweep_plan(h);
            free_binned(h);
            h->spread = -1.0;
            choose_csr_kernel(h, SMVP_CSR_KERNEL_STREAM, 0);
            if (build_stream_plan(h) != SMVP_OK)
                free_stream_plan(h);  // smvp_csr_spmv refuses a handle without a plan
            h->plan_build_ms = wall_ms() - t0;
            return smvp::fail(rc, "%s", why.c_str());
        }
    } else if (h->kernel != SMVP_CSR_KERNEL_VECTOR) {
        rc = build_stream_plan(h);
    } else {
        free_stream_plan(h);
    }
    h->plan_build_ms = wall_ms() - t0;
    return rc;
}

extern "C" int smvp_csr_get_kernel(const smvp_csr_t *h, int *kernel, int *param)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    if (kernel)
        *kernel = h->kernel;
    if (param)
        *param = h->kernel == SMVP_CSR_KERNEL_VECTOR ? h->lanes_per_row
                 : h->kernel == SMVP_CSR_KERNEL_COLSWEEP ? (h->sweep_rb | ((h->sweep_parts == 8 ? 3 : h->sweep_parts == 4 ? 2 : h->sweep_parts == 2 ? 1 : 0) << kSweepPartsShift))
                 : h->kernel == SMVP_CSR_KERNEL_BINNED   ? h->bin.band
                                                         : h->vpt * smvp::kStreamBlock;
    return SMVP_OK;
}

extern "C" int smvp_csr_gather_spread(smvp_csr_t *h, double *spread)
{
    if (!h || !spread)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_gather_spread: bad argument");
    DeviceScope on(h->device);
    if (h->spread < -1.5)
        h->spread = h->flavor == smvp::kFlavorCsr ? csr_gather_spread(h) : -1.0;
    *spread = h->spread;
    return SMVP_OK;
}

static void fill_owner_launch(const smvp_csr_t *h, const double *d_x, double *d_y, unsigned long long *stamps, smvp::OwnerLaunch *out)
{
    smvp::OwnerLaunch &l = *out;
    l.row_ptr = h->d_row_ptr, l.col_ind = h->d_col_ind, l.val = h->d_val, l.x = d_x, l.y = d_y;
    l.tile_row = h->d_tile_row, l.tile_next = h->d_tile_next;
    l.pos = h->d_pos, l.start_pos = h->d_start_pos;
    if (h->flavor == smvp::kFlavorTjdsS || h->flavor == smvp::kFlavorTjdsH) {
        l.pos = h->d_pos_sorted, l.col_ind = h->d_meta;
        l.ovf_ptr = h->d_ovf_ptr, l.ovf_val = h->d_ovf_val, l.ovf_k = h->d_ovf_k;
        l.cache_ptr = h->d_cache_ptr, l.val_cache = h->d_val_cache;
        l.word32 = h->d_word32, l.group_run = h->d_group_run, l.run_ptr = h->d_run_ptr, l.run_tab = h->d_run_tab;
    }
    l.stamps = stamps;
    l.rows = h->rows, l.nnz = h->nnz, l.ntiles = h->ntiles;
    l.col16 = h->d_col16, l.col_base = h->d_col_base;
    l.row_rel = h->d_row_rel;
    l.unit_x = h->unit_operand ? 1 : 0;
}

// `reps` products of the tile kernel in ONE launch, each product's window stamped (csr_stream_owner_repeat); grid from
// csr_repeat_grid.  The products are those of csr_spmv_impl, bit for bit: the same kernel body walks the same tiles.
static int csr_repeat_grid(const smvp_csr_t *h)
{
    if (!h || h->kernel != SMVP_CSR_KERNEL_STREAM || !h->d_tile_row || h->rows <= 0)
        return 0;
    DeviceScope on(h->device);
    return smvp::owner_repeat_grid(h->vpt, h->d_col16 ? smvp::kFlavorCsr16 : h->flavor, h->ntiles);
}

static int csr_spmv_repeat(smvp_csr_t *h, const double *d_x, double *d_y, void *stream, unsigned long long *stamps, int reps, int grid,
                           unsigned *ctl_words, bool first_of_run, unsigned long long patience)
{
    DeviceScope on(h->device);
    smvp::OwnerLaunch l{};
    fill_owner_launch(h, d_x, d_y, stamps, &l);
    const hipError_t e = smvp::launch_csr_stream_owner_repeat(h->vpt, h->d_col16 ? smvp::kFlavorCsr16 : h->flavor, l, reps, grid, ctl_words,
                                                              first_of_run, patience, (hipStream_t)stream);
    if (e != hipSuccess)
        return smvp::fail(SMVP_ERR_HIP, "repeating CSR launch failed: %s", hipGetErrorString(e));
    return SMVP_OK;
}

// stamps: device-side timing slots of this launch (owner kernel only), or nullptr
static int csr_spmv_impl(smvp_csr_t *h, const double *d_x, double *d_y, void *stream, unsigned long long *stamps)
{
    if (!h || (h->rows > 0 && !d_y) || (h->nnz > 0 && !d_x))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_spmv: bad argument");
    if ((h->kernel == SMVP_CSR_KERNEL_COLSWEEP && !h->d_sweep_ptr) || (h->kernel == SMVP_CSR_KERNEL_BINNED && !h->near && !h->bin.nw.on) ||
        (h->kernel != SMVP_CSR_KERNEL_VECTOR && h->kernel != SMVP_CSR_KERNEL_COLSWEEP && h->kernel != SMVP_CSR_KERNEL_BINNED &&
         !h->d_tile_row))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_spmv: the handle has no launch plan (a re-plan failed earlier)");
    DeviceScope on(h->device);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (h->kernel == SMVP_CSR_KERNEL_BINNED) {
        // the near part writes every row of y; pass A puts the far products into the bins; pass B adds each row's far sum.
        // (With the near part on the TILE kernel, pass A was tried beside it twice.  On a stream of its own its one workgroup per
        // CU -- 132 KB of LDS -- only got onto a CU once the tile kernel's six had drained: one after the other anyway,
        // 0.777 against 0.757 ms.  As one persistent workgroup per CU enqueued AHEAD of the near product the two did share
        // the CUs -- and pass A then took 264 instead of 238 us while the near product finished 241 us after it instead of
        // 303: 0.7485 against 0.7540 ms.  What one gains the other loses: profiles/r04_binned_measured.txt.)
        if (h->bin.nw.on && h->side) {
            // The window plan's near part and pass A are both one-workgroup-per-CU kernels (148 / 132 KB of LDS): enqueued
            // side by side -- pass A first, on a stream of its own: it needs x only -- their workgroups share the chip one
            // CU at a time, the near part's take the CUs that pass A's last workgroups leave, and a kernel that mostly
            // reads runs beside one that writes 40 % of its bytes: 0.683-0.697 -> 0.651-0.669 ms on the random model
            // (profiles/r04_binned_measured.txt, section 12; the near part first: 0.668-0.670).  Pass B waits for both.
            HIP_TRY(hipEventRecord(h->ev_fork, st));
            HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
            e = smvp::launch_binned_products(h->bin, d_x, h->side);
            if (e != hipSuccess)
                return smvp::fail(SMVP_ERR_HIP, "CSR launch failed: %s", hipGetErrorString(e));
            // From here on pass A is in flight on the side stream, reading x and writing the bins: whatever fails below, the
            // caller's stream is joined to it before the error goes back -- the caller may free x or destroy the handle next.
            int rc = SMVP_OK;
            hipError_t je = hipEventRecord(h->ev_join, h->side);
            if (je == hipSuccess) {
                e = smvp::launch_near_window(h->bin.nw, d_x, d_y, st);
                if (e != hipSuccess)
                    rc = smvp::fail(SMVP_ERR_HIP, "CSR launch failed: %s", hipGetErrorString(e));
                je = hipStreamWaitEvent(st, h->ev_join, 0);
            }
            if (je != hipSuccess) {
                (void)hipStreamSynchronize(h->side);  // the join could not be enqueued: wait for pass A here
                return smvp::fail(SMVP_ERR_HIP, "joining the binned plan's side stream failed: %s", hipGetErrorString(je));
            }
            if (rc != SMVP_OK)
                return rc;
            e = smvp::launch_binned_sums(h->bin, d_y, st);
            if (e != hipSuccess)
                return smvp::fail(SMVP_ERR_HIP, "CSR launch failed: %s", hipGetErrorString(e));
            return SMVP_OK;
        }
        if (h->bin.nw.on) {
            e = smvp::launch_near_window(h->bin.nw, d_x, d_y, st);
            if (e != hipSuccess)
                return smvp::fail(SMVP_ERR_HIP, "CSR launch failed: %s", hipGetErrorString(e));
        } else if (int rc = csr_spmv_impl(h->near, d_x, d_y, stream, nullptr))
            return rc;
        e = smvp::launch_binned_products(h->bin, d_x, st);
        if (e != hipSuccess)
            return smvp::fail(SMVP_ERR_HIP, "CSR launch failed: %s", hipGetErrorString(e));
        e = smvp::launch_binned_sums(h->bin, d_y, st);
    } else if (h->kernel == SMVP_CSR_KERNEL_COLSWEEP)
        e = smvp::launch_csr_colsweep(h->d_sweep_ptr, h->d_sweep_col, h->d_sweep_val, h->d_sweep_row, d_x, d_y, h->rows,
                                      (h->sweep_parts == smvp::kSweepXcdParts ? h->sweep_rb : h->sweep_rb * h->sweep_parts) / smvp::kSweepWaves,
                                      h->sweep_parts, h->sweep_per_launch, h->sweep_g, h->d_sweep_part, st);
    else if (h->kernel == SMVP_CSR_KERNEL_VECTOR)
        e = smvp::launch_csr_vector(h->lanes_per_row, h->d_row_ptr, h->d_col_ind, h->d_val, d_x, d_y, h->rows, st);
    else if (h->kernel == SMVP_CSR_KERNEL_STREAM) {
        smvp::OwnerLaunch l{};
        fill_owner_launch(h, d_x, d_y, stamps, &l);
        e = smvp::launch_csr_stream_owner(h->vpt, h->d_col16 ? smvp::kFlavorCsr16 : h->flavor, l, st);
    } else
        e = smvp::launch_csr_stream(h->vpt, h->d_row_ptr, h->d_col_ind, h->d_val, d_x, d_y, h->d_tile_row,
                                    h->d_carry_row, h->d_carry, h->rows, h->nnz, h->ntiles, st);
    if (e != hipSuccess)
        return smvp::fail(SMVP_ERR_HIP, "CSR launch failed: %s", hipGetErrorString(e));
    return SMVP_OK;
}

extern "C" int smvp_csr_spmv(smvp_csr_t *h, const double *d_x, double *d_y, void *stream)
{
    return csr_spmv_impl(h, d_x, d_y, stream, nullptr);
}

// the owner kernel of a launch small enough to be timed on the device (see StampTimer)?
static bool csr_can_stamp(const smvp_csr_t *h) { return h && h->kernel == SMVP_CSR_KERNEL_STREAM && h->d_tile_row; }

extern "C" int smvp_csr_describe(const smvp_csr_t *h, char *kernel_name, size_t cap, double *alg_bytes)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    if (kernel_name && cap) {
        if (h->kernel == SMVP_CSR_KERNEL_BINNED && h->bin.nw.on)
            snprintf(kernel_name, cap, "csr_binned: csr_near_window + csr_binned_far_products<2> + csr_binned_far_sums<%d, %d, 2>", h->bin.slots,
                     h->bin.threads_b);
        else if (h->kernel == SMVP_CSR_KERNEL_BINNED)
            snprintf(kernel_name, cap, "csr_binned: csr_stream_owner<%d, %d, false> + csr_binned_far_products<2> + csr_binned_far_sums<%d, %d, 2>",
                     h->near ? h->near->vpt : 0, h->near ? (h->near->d_col16 ? smvp::kFlavorCsr16 : h->near->flavor) : 0,
                     h->bin.slots, h->bin.threads_b);
        else if (h->kernel == SMVP_CSR_KERNEL_COLSWEEP)
            if (h->sweep_parts == smvp::kSweepXcdParts)
                snprintf(kernel_name, cap, "csr_colsweep<%d> (8 column parts, one per XCD) + sweep_combine", h->sweep_g);
            else if (h->sweep_parts > 1)
                snprintf(kernel_name, cap, "csr_colsweep<%d> (%d column parts)", h->sweep_g, h->sweep_parts);
            else
                snprintf(kernel_name, cap, "csr_colsweep<%d>", h->sweep_g);
        else if (h->kernel == SMVP_CSR_KERNEL_VECTOR)
            snprintf(kernel_name, cap, "csr_vector_rows<%d>", h->lanes_per_row);
        else if (h->kernel == SMVP_CSR_KERNEL_STREAM)
            snprintf(kernel_name, cap, "csr_stream_owner<%d, %d, false>", h->vpt, h->d_col16 ? smvp::kFlavorCsr16 : h->flavor);
        else
            snprintf(kernel_name, cap, "csr_stream_tiles<%d>", h->vpt);
    }
    if (alg_bytes)
        *alg_bytes = 12.0 * h->nnz + 4.0 * (h->rows + 1.0) + 8.0 * h->cols + 8.0 * h->rows;
    return SMVP_OK;
}

// kernel launches one product of the current plan takes (the column sweep starts its workgroups in generations)
extern "C" int smvp_csr_plan_launches(const smvp_csr_t *h, int *launches)
{
    if (!h || !launches)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_plan_launches: bad argument");
    *launches = 1;
    if (h->kernel == SMVP_CSR_KERNEL_COLSWEEP && h->sweep_rb > 0 && h->sweep_per_launch > 0) {
        const bool xcd = h->sweep_parts == smvp::kSweepXcdParts;
        const int nwg = (h->rows + h->sweep_rb - 1) / h->sweep_rb * (xcd ? smvp::kSweepXcdParts : 1);
        *launches = std::max(1, (nwg + h->sweep_per_launch - 1) / h->sweep_per_launch) + (xcd ? 1 : 0);  // (+ sweep_combine)
    } else if (h->kernel == SMVP_CSR_KERNEL_STREAM_CARRY && h->ntiles > 1) {
        *launches = 2;
    } else if (h->kernel == SMVP_CSR_KERNEL_BINNED && h->bin.nw.on) {
        *launches = 1 + (h->bin.nw.n_out > 0 ? 1 : 0) + (h->bin.nf > 0 ? 2 : 0);
    } else if (h->kernel == SMVP_CSR_KERNEL_BINNED && h->near) {
        int near = 1;
        (void)smvp_csr_plan_launches(h->near, &near);
        *launches = near + (h->bin.nf > 0 ? 2 : 0);
    }
    return SMVP_OK;
}

// bytes of device memory the current launch plan keeps beside row_ptr / col_ind / val
static double csr_plan_bytes(const smvp_csr_t *h)
{
    const double n = h->nnz, t = h->ntiles;
    if (h->kernel == SMVP_CSR_KERNEL_BINNED)
        return (double)h->bin.plan_bytes + (h->near ? csr_plan_bytes(h->near) : 0.0);
    if (h->kernel == SMVP_CSR_KERNEL_COLSWEEP) {
        const bool xcd = h->sweep_parts == smvp::kSweepXcdParts;
        const int strip_rows = std::max(1, (xcd ? h->sweep_rb : h->sweep_rb * h->sweep_parts) / smvp::kSweepWaves);
        return 14.0 * n + 8.0 * ((double)((h->rows + strip_rows - 1) / strip_rows) * h->sweep_parts + 2) + (xcd ? 64.0 * h->rows : 0.0);
    }
    if (h->kernel == SMVP_CSR_KERNEL_VECTOR)
        return 0.0;
    double b = 4.0 * (t + 1) + 4.0 * t;  // tile_row + tile_next (or carry_row)
    if (h->kernel == SMVP_CSR_KERNEL_STREAM_CARRY)
        b += 8.0 * t;
    if (h->d_col16)
        b += 2.0 * n + 4.0 * (n / 1024 + 1);
    if (h->d_row_rel)
        b += 2.0 * h->rows;
    if (h->flavor == smvp::kFlavorTjdsS || h->flavor == smvp::kFlavorTjdsH) {
        b += 4.0 * (t + 1) + 12.0 * h->ovf_total + 4.0 * (t + 2) + 8.0 * h->cached_total;
        b += h->flavor == smvp::kFlavorTjdsH ? 4.0 * n + 4.0 * (t + 2) + 8.0 * h->runs_total + 2.0 * (n / 32 + 1) : 8.0 * n;
    }
    return b;
}

extern "C" int smvp_csr_plan_info(const smvp_csr_t *h, smvp_plan_info_t *out)
{
    if (!h || !out)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_plan_info: bad argument");
    out->matrix_bytes = 12.0 * h->nnz + 4.0 * (h->rows + 1.0);
    out->plan_bytes = csr_plan_bytes(h);
    out->build_ms = h->plan_build_ms;
    return SMVP_OK;
}

extern "C" void smvp_csr_destroy(smvp_csr_t *h)
{
    if (!h)
        return;
    DeviceScope on(h->device);
    smvp::debug_owner_phases();  // (print in diagnostic builds only)
    smvp::debug_binned_phases();
    free_stream_plan(h);
    free_sweep_plan(h);
    free_binned(h);
    if (h->own_row_ptr && h->d_row_ptr)
        (void)hipFree(h->d_row_ptr);
    if (h->own_col_ind && h->d_col_ind)
        (void)hipFree(h->d_col_ind);
    if (h->own_val && h->d_val)
        (void)hipFree(h->d_val);
    delete h;
}

// ===========================================================================
// TJDS
// ===========================================================================
struct smvp_tjds {
    int device = 0;
    int rows = 0, cols = 0, nnz = 0, num_diag = 0;
    int *d_perm = nullptr;
    int *d_start_pos = nullptr;  // num_diag + 1 entries (+1 pad)
    int *d_row_ind = nullptr;
    double *d_val = nullptr;
    bool own_perm = false, own_start_pos = false, own_row_ind = false, own_val = false;
    std::vector<int> h_start_pos;

    double *d_x_perm = nullptr;  // max(rows, cols) doubles
    bool x_set = false;

    int mode = SMVP_TJDS_MODE_ROW_GATHER;

    // one-kernel product (ROW_GATHER): the entries regrouped by row -- segment bounds and TJDS positions, plus the
    // permuted columns when the 32-bit form is used; `rg` is the owner-kernel plan over that stream (the
    // tile-ordered form keeps its own sorted copies).  Built on first use of the mode.
    int *d_rg_ptr = nullptr;      // rows + 1
    int *d_rg_pos = nullptr;      // nnz, row order
    int *d_rg_k = nullptr;        // nnz, kFlavorTjdsK only
    smvp_csr *rg = nullptr;

    // two-phase product (TWO_PHASE): per-entry products + their sum per row through the row-inverted index;
    // built on first use of the mode
    double *d_prod = nullptr;    // nnz doubles
    int *d_inv_ptr = nullptr;    // rows + 1
    int *d_inv_pos = nullptr;    // nnz: positions j grouped by row_ind[j], ascending inside a row
    smvp_csr *inv = nullptr;     // unit-value CSR over (inv_ptr, inv_pos), x = prod

    // launch plan (rebuilt when ref-quirks mode changes)
    bool quirks = false;
    int *d_plan_start_pos = nullptr;  // start_pos as the kernel should see it
    int4 *d_work = nullptr;
    int nwork = 0;
    long long planned_nnz = 0;
    double plan_build_ms = 0.0;  // host wall time of the plan builds so far (work items + the modes' plans)
};

namespace {

int build_tjds_plan(smvp_tjds *h, bool quirks, int ref_num_tjdiag, int last_diag_single)
{
    if (h->d_plan_start_pos)
        (void)hipFree(h->d_plan_start_pos);
    if (h->d_work)
        (void)hipFree(h->d_work);
    h->d_plan_start_pos = nullptr;
    h->d_work = nullptr;

    // start_pos as the product loop sees it, plus two readable pads
    std::vector<int> sp((size_t)h->num_diag + 3, 0);
    for (int d = 0; d <= h->num_diag; ++d)
        sp[(size_t)d] = h->h_start_pos[(size_t)d];
    int diag_limit = h->num_diag;
    if (quirks) {
        // main-cli.c:865 + :1013: diagonals 0 .. ref_num_tjdiag inclusive;
        // main-cli.c:951-966: terminator never written after a one-entry last
        // diagonal, and the malloc'd array reads as zero there.
        if (last_diag_single)
            sp[(size_t)h->num_diag] = 0;
        diag_limit = std::min(h->num_diag, ref_num_tjdiag + 1);
    }
    std::vector<int4> work;
    long long planned = 0;
    for (int d0 = 0; d0 < diag_limit; d0 += smvp::kTjdsDiagChunk) {
        const int d1 = std::min(d0 + smvp::kTjdsDiagChunk, diag_limit);
        const int width = sp[(size_t)d0 + 1] - sp[(size_t)d0];  // widest diagonal of the chunk
        for (int k0 = 0; k0 < width; k0 += smvp::kTjdsBlock)
            work.push_back(make_int4(k0, d0, d1, 0));
        for (int d = d0; d < d1; ++d)
            planned += std::max(0, sp[(size_t)d + 1] - sp[(size_t)d]);
    }
    if (int rc = upload(&h->d_plan_start_pos, sp))
        return rc;
    if (int rc = upload(&h->d_work, work))
        return rc;
    h->nwork = (int)work.size();
    h->quirks = quirks;
    h->planned_nnz = planned;
    return SMVP_OK;
}


void free_row_gather(smvp_tjds *h)
{
    smvp_csr_destroy(h->rg);
    h->rg = nullptr;
    for (void *p : {(void *)h->d_rg_ptr, (void *)h->d_rg_pos, (void *)h->d_rg_k})
        if (p)
            (void)hipFree(p);
    h->d_rg_ptr = h->d_rg_pos = h->d_rg_k = nullptr;
}

// How the row-gather stream names an entry: tile-ordered streams with two 16-bit words per entry -- the low half of the
// position and slot | run hint -- plus the tiles' run tables (kFlavorTjdsH, 4 bytes of index per entry: the default); the same
// order with the 32-bit position and a 32-bit slot | diagonal word (kFlavorTjdsS, 8 bytes; needs the diagonals to fit 21 bits);
// or 32-bit permuted columns in row
// order (kFlavorTjdsK).  The plan option "tjds_index" = 0 | 1 | 2 selects (smvp_set_option; the tests run all three).
int row_gather_index(const smvp_tjds *h)
{
    const int e = smvp::option("tjds_index", 0);
    const bool fits_sorted = ((long long)std::max(h->num_diag - 1, 0) >> (32 - smvp::kSlotBits)) == 0;
    if (e == 2)
        return smvp::kFlavorTjdsK;
    if (e == 1 && fits_sorted)
        return smvp::kFlavorTjdsS;
    return smvp::kFlavorTjdsH;
}

int ensure_row_gather(smvp_tjds *h)
{
    if (h->rg)
        return SMVP_OK;
    free_row_gather(h);
    const size_t n = (size_t)std::max(h->nnz, 4);
    const int index = row_gather_index(h);
    if (hipMalloc((void **)&h->d_rg_ptr, ((size_t)h->rows + 4) * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&h->d_rg_pos, n * sizeof(int)) != hipSuccess ||
        (index == smvp::kFlavorTjdsK && hipMalloc((void **)&h->d_rg_k, n * sizeof(int)) != hipSuccess))
        return smvp::fail(SMVP_ERR_ALLOC, "TJDS: cannot allocate the row-gather plan");
    // the true start_pos (d_plan_start_pos may carry the ref-quirks edit)
    if (int rc = smvp::build_row_gather_plan(h->d_row_ind, h->d_start_pos, h->num_diag, h->nnz, h->rows, h->d_rg_ptr,
                                             h->d_rg_pos, h->d_rg_k, nullptr))
        return rc;
    TjdsSource src;
    src.pos = h->d_rg_pos, src.start_pos = h->d_start_pos, src.num_diag = h->num_diag;
    return csr_create_impl(&h->rg, h->device, h->rows, std::max(h->cols, 1), h->nnz, h->d_rg_ptr, h->d_rg_k, h->d_val,
                           SMVP_MEM_DEVICE, nullptr, index, &src);
}

int ensure_two_phase(smvp_tjds *h)
{
    if (h->inv)
        return SMVP_OK;
    const size_t n = (size_t)std::max(h->nnz, 4);
    if ((!h->d_prod && hipMalloc((void **)&h->d_prod, n * sizeof(double)) != hipSuccess) ||
        (!h->d_inv_pos && hipMalloc((void **)&h->d_inv_pos, n * sizeof(int)) != hipSuccess) ||
        (!h->d_inv_ptr && hipMalloc((void **)&h->d_inv_ptr, ((size_t)h->rows + 4) * sizeof(int)) != hipSuccess))
        return smvp::fail(SMVP_ERR_ALLOC, "TJDS: cannot allocate the two-phase buffers");
    if (int rc = smvp::build_row_inverse(h->d_row_ind, h->nnz, h->rows, h->d_inv_ptr, h->d_inv_pos, nullptr))
        return rc;
    // the second phase walks the products the way the one-kernel form walks val: every tile's entries in TJDS order (neighbouring
    // lanes read neighbouring products), one 32-bit index word per entry, no operand (round 5; before: a unit-value CSR over the
    // row-inverted index, every product a gather of its own: 0.83 ms on memplus x944)
    TjdsSource src;
    src.pos = h->d_inv_pos, src.start_pos = h->d_start_pos, src.num_diag = h->num_diag, src.unit_operand = true;
    return csr_create_impl(&h->inv, h->device, h->rows, std::max(h->cols, 1), h->nnz, h->d_inv_ptr, nullptr, h->d_prod,
                           SMVP_MEM_DEVICE, nullptr, smvp::kFlavorTjdsH, &src);
}

int ensure_mode_plan(smvp_tjds *h)
{
    switch (h->mode) {
    case SMVP_TJDS_MODE_ROW_GATHER:
        return ensure_row_gather(h);
    case SMVP_TJDS_MODE_TWO_PHASE:
        return ensure_two_phase(h);
    default:
        return SMVP_OK;
    }
}

bool overwrites_y(const smvp_tjds *h) { return h->mode != SMVP_TJDS_MODE_ATOMIC && !h->quirks; }

}  // namespace

extern "C" int smvp_tjds_create(smvp_tjds_t **out, int device, int rows, int cols, int nnz, int num_diag,
                                const int *perm, const int *start_pos, const int *row_ind,
                                const double *val, int mem_kind)
{
    if (!out || rows < 0 || cols < 0 || nnz < 0 || num_diag < 0 || !start_pos || (cols > 0 && !perm) ||
        (nnz > 0 && (!row_ind || !val)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_create: bad argument");
    if (mem_kind != SMVP_MEM_HOST && mem_kind != SMVP_MEM_DEVICE)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_create: bad mem_kind");
    if (nnz > kMaxEntries)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "smvp_tjds_create: %d entries: shard blocks this large by rows", nnz);
    if (int rc = usable_device(device))
        return rc;
    DeviceScope on(device);

    smvp_tjds *h = new smvp_tjds;
    h->device = device;
    h->rows = rows, h->cols = cols, h->nnz = nnz, h->num_diag = num_diag;
    h->h_start_pos.resize((size_t)num_diag + 1);
    int rc = SMVP_OK;
    if (mem_kind == SMVP_MEM_HOST)
        memcpy(h->h_start_pos.data(), start_pos, sizeof(int) * ((size_t)num_diag + 1));
    else if (hipMemcpy(h->h_start_pos.data(), start_pos, sizeof(int) * ((size_t)num_diag + 1), hipMemcpyDeviceToHost) != hipSuccess)
        rc = smvp::fail(SMVP_ERR_HIP, "smvp_tjds_create: cannot read start_pos back from the device");
    if (rc == SMVP_OK) {
        // diagonals start at 0, end at nnz, and never get longer; the first one
        // has at most `cols` entries -- the kernel's indexing relies on all of it
        const std::vector<int> &sp = h->h_start_pos;
        bool ok = sp[0] == 0 && sp[(size_t)num_diag] == nnz;
        int prev = cols;
        for (int d = 0; d < num_diag && ok; ++d) {
            const int len = sp[(size_t)d + 1] - sp[(size_t)d];
            ok = len >= 1 && len <= prev;
            prev = len;
        }
        if (!ok)
            rc = smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_create: start_pos is not a valid jagged-diagonal index");
    }
    if (rc == SMVP_OK && mem_kind == SMVP_MEM_HOST) {
        for (int j = 0; j < nnz && rc == SMVP_OK; ++j)
            if (row_ind[j] < 0 || row_ind[j] >= rows)
                rc = smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_create: row_ind[%d] = %d outside [0, %d)", j, row_ind[j], rows);
        for (int k = 0; k < cols && rc == SMVP_OK; ++k)
            if (perm[k] < 0 || perm[k] >= cols)
                rc = smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_create: perm[%d] = %d outside [0, %d)", k, perm[k], cols);
    }
    if (rc == SMVP_OK && mem_kind == SMVP_MEM_DEVICE)
        rc = check_device_indices(row_ind, nnz, rows, "smvp_tjds_create: row_ind");
    if (rc == SMVP_OK && mem_kind == SMVP_MEM_DEVICE)
        rc = check_device_indices(perm, cols, cols, "smvp_tjds_create: perm");
    if (rc == SMVP_OK)
        rc = to_device(&h->d_perm, perm, (size_t)cols, mem_kind, &h->own_perm);
    if (rc == SMVP_OK)
        rc = to_device(&h->d_start_pos, start_pos, (size_t)num_diag + 1, mem_kind, &h->own_start_pos);
    if (rc == SMVP_OK)
        rc = to_device(&h->d_row_ind, row_ind, (size_t)nnz, mem_kind, &h->own_row_ind);
    if (rc == SMVP_OK)
        rc = to_device(&h->d_val, val, (size_t)nnz, mem_kind, &h->own_val);
    if (rc == SMVP_OK) {
        const size_t n = (size_t)std::max(std::max(rows, cols), 1);
        if (hipMalloc((void **)&h->d_x_perm, n * sizeof(double)) != hipSuccess ||
            hipMemset(h->d_x_perm, 0, n * sizeof(double)) != hipSuccess)
            rc = smvp::fail(SMVP_ERR_ALLOC, "smvp_tjds_create: cannot allocate the permuted operand");
    }
    const double t0 = wall_ms();
    if (rc == SMVP_OK)
        rc = build_tjds_plan(h, false, 0, 0);
    if (rc == SMVP_OK)
        rc = ensure_mode_plan(h);
    h->plan_build_ms = wall_ms() - t0;
    if (rc != SMVP_OK) {
        smvp_tjds_destroy(h);
        return rc;
    }
    *out = h;
    return SMVP_OK;
}

extern "C" int smvp_tjds_set_mode(smvp_tjds_t *h, int mode)
{
    if (!h || mode < SMVP_TJDS_MODE_AUTO || mode > SMVP_TJDS_MODE_ROW_GATHER)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_set_mode: bad argument");
    DeviceScope on(h->device);
    const int before = h->mode;
    h->mode = mode == SMVP_TJDS_MODE_AUTO ? SMVP_TJDS_MODE_ROW_GATHER : mode;
    const double t0 = wall_ms();
    if (int rc = ensure_mode_plan(h)) {
        h->mode = before;
        return rc;
    }
    h->plan_build_ms += wall_ms() - t0;
    return SMVP_OK;
}

extern "C" int smvp_tjds_set_x(smvp_tjds_t *h, const double *d_x, void *stream)
{
    if (!h || (h->cols > 0 && !d_x))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_set_x: bad argument");
    DeviceScope on(h->device);
    hipError_t e = smvp::launch_tjds_permute(h->d_perm, d_x, h->d_x_perm, h->cols, (hipStream_t)stream);
    if (e != hipSuccess)
        return smvp::fail(SMVP_ERR_HIP, "operand permute launch failed: %s", hipGetErrorString(e));
    h->x_set = true;
    return SMVP_OK;
}

extern "C" int smvp_tjds_zero_y(smvp_tjds_t *h, double *d_y, void *stream)
{
    if (!h || (h->rows > 0 && !d_y))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_zero_y: bad argument");
    if (overwrites_y(h))
        return SMVP_OK;  // the row-gather and two-phase products overwrite y
    DeviceScope on(h->device);
    if (h->rows > 0)
        HIP_TRY(hipMemsetAsync(d_y, 0, sizeof(double) * (size_t)h->rows, (hipStream_t)stream));
    return SMVP_OK;
}

static int tjds_spmv_impl(smvp_tjds_t *h, double *d_y, void *stream, unsigned long long *stamps)
{
    if (!h || (h->rows > 0 && !d_y))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_spmv: bad argument");
    if (!h->x_set)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_spmv: call smvp_tjds_set_x first");
    if (h->quirks && h->rows != h->cols)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "ref-quirks mode indexes the operand by row and needs a square matrix");
    DeviceScope on(h->device);
    if (!h->quirks && h->mode == SMVP_TJDS_MODE_ROW_GATHER)
        return csr_spmv_impl(h->rg, h->d_x_perm, d_y, stream, stamps);
    if (h->mode == SMVP_TJDS_MODE_TWO_PHASE && !h->quirks) {
        hipError_t e1 = smvp::launch_tjds_products(h->d_plan_start_pos, h->d_val, h->d_x_perm, h->d_prod, h->d_work,
                                                   h->nwork, h->cols, (hipStream_t)stream);
        if (e1 != hipSuccess)
            return smvp::fail(SMVP_ERR_HIP, "TJDS products launch failed: %s", hipGetErrorString(e1));
        return smvp_csr_spmv(h->inv, h->d_prod, d_y, stream);
    }
    hipError_t e = smvp::launch_tjds_scatter(h->quirks, h->d_plan_start_pos, h->d_row_ind, h->d_val, h->d_x_perm, d_y,
                                             h->d_work, h->nwork, h->cols, (hipStream_t)stream);
    if (e != hipSuccess)
        return smvp::fail(SMVP_ERR_HIP, "TJDS launch failed: %s", hipGetErrorString(e));
    return SMVP_OK;
}

extern "C" int smvp_tjds_spmv(smvp_tjds_t *h, double *d_y, void *stream)
{
    return tjds_spmv_impl(h, d_y, stream, nullptr);
}

static bool tjds_can_stamp(const smvp_tjds_t *h)
{
    return h && !h->quirks && h->mode == SMVP_TJDS_MODE_ROW_GATHER && csr_can_stamp(h->rg);
}

extern "C" int smvp_tjds_set_ref_quirks(smvp_tjds_t *h, int enable, int ref_num_tjdiag, int last_diag_single)
{
    if (!h || (enable && ref_num_tjdiag < 0))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_set_ref_quirks: bad argument");
    DeviceScope on(h->device);
    return build_tjds_plan(h, enable != 0, ref_num_tjdiag, last_diag_single);
}

extern "C" int smvp_tjds_describe(const smvp_tjds_t *h, char *kernel_name, size_t cap, double *alg_bytes)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    if (kernel_name && cap) {
        if (h->quirks || h->mode == SMVP_TJDS_MODE_ATOMIC)
            snprintf(kernel_name, cap, "tjds_colmajor_scatter<%s>", h->quirks ? "true" : "false");
        else if (h->mode == SMVP_TJDS_MODE_TWO_PHASE)
            snprintf(kernel_name, cap, "tjds_colmajor_products + csr_stream_owner<%d, %d, false>", h->inv ? h->inv->vpt : 0,
                     h->inv ? h->inv->flavor : 0);
        else
            snprintf(kernel_name, cap, "csr_stream_owner<%d, %d, false>", h->rg ? h->rg->vpt : 0, h->rg ? h->rg->flavor : 0);
    }
    if (alg_bytes)
        *alg_bytes = 12.0 * h->planned_nnz + 4.0 * (h->num_diag + 1.0) + 8.0 * h->cols + 8.0 * h->rows;
    return SMVP_OK;
}

// Which values the one-kernel product keeps a second copy of: those of val lines whose 16 entries belong to
// `min_tiles` tiles or more (0: none -- every value is read from val itself).  Rebuilds the plan.
extern "C" int smvp_tjds_set_value_cache(smvp_tjds_t *h, int min_tiles)
{
    if (!h || !h->rg || min_tiles < 0 || min_tiles > 16)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_set_value_cache: needs the row-gather plan and 0 <= min_tiles <= 16");
    if (h->rg->flavor != smvp::kFlavorTjdsS && h->rg->flavor != smvp::kFlavorTjdsH)
        return min_tiles == 0 ? (int)SMVP_OK
                              : smvp::fail(SMVP_ERR_UNSUPPORTED, "the value cache belongs to the tile-ordered TJDS stream");
    DeviceScope on(h->device);
    h->rg->cache_min_tiles = min_tiles;
    return build_stream_plan(h->rg);
}

extern "C" int smvp_tjds_get_value_cache(const smvp_tjds_t *h, int *min_tiles, long long *cached_entries)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    const bool on = h->rg && (h->rg->flavor == smvp::kFlavorTjdsS || h->rg->flavor == smvp::kFlavorTjdsH);
    if (min_tiles)
        *min_tiles = on ? h->rg->cache_min_tiles : 0;
    if (cached_entries)
        *cached_entries = on ? h->rg->cached_total : 0;
    return SMVP_OK;
}

extern "C" int smvp_tjds_plan_info(const smvp_tjds_t *h, smvp_plan_info_t *out)
{
    if (!h || !out)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_plan_info: bad argument");
    const double n = h->nnz;
    out->matrix_bytes = 12.0 * n + 4.0 * (h->num_diag + 1.0) + 4.0 * h->cols;
    double b = 8.0 * std::max(h->rows, h->cols);             // x_perm
    b += 4.0 * (h->num_diag + 3.0) + 16.0 * h->nwork;        // the column-major work items (atomic / two-phase / ref-quirks)
    if (h->rg)
        b += 4.0 * (h->rows + 4.0) + 4.0 * n + (h->d_rg_k ? 4.0 * n : 0.0) + csr_plan_bytes(h->rg);
    if (h->inv)
        b += 8.0 * n + 4.0 * (h->rows + 4.0) + 4.0 * n + csr_plan_bytes(h->inv);
    out->plan_bytes = b;
    out->build_ms = h->plan_build_ms;
    return SMVP_OK;
}

// tile size of the row-gather product (development knob; 256, 1024 or 2048 entries)
extern "C" int smvp_tjds_set_tile(smvp_tjds_t *h, int entries_per_tile)
{
    if (!h || !h->rg)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_set_tile: the handle has no row-gather plan");
    return smvp_csr_set_kernel(h->rg, SMVP_CSR_KERNEL_STREAM, entries_per_tile);
}

extern "C" void smvp_tjds_destroy(smvp_tjds_t *h)
{
    if (!h)
        return;
    DeviceScope on(h->device);
    free_row_gather(h);
    if (h->own_perm && h->d_perm)
        (void)hipFree(h->d_perm);
    if (h->own_start_pos && h->d_start_pos)
        (void)hipFree(h->d_start_pos);
    if (h->own_row_ind && h->d_row_ind)
        (void)hipFree(h->d_row_ind);
    if (h->own_val && h->d_val)
        (void)hipFree(h->d_val);
    if (h->d_x_perm)
        (void)hipFree(h->d_x_perm);
    if (h->d_plan_start_pos)
        (void)hipFree(h->d_plan_start_pos);
    if (h->d_work)
        (void)hipFree(h->d_work);
    smvp_csr_destroy(h->inv);
    for (void *p : {(void *)h->d_prod, (void *)h->d_inv_ptr, (void *)h->d_inv_pos})
        if (p)
            (void)hipFree(p);
    delete h;
}

// ===========================================================================
// device queries
// ===========================================================================
extern "C" int smvp_device_count(int *count)
{
    if (!count)
        return smvp::fail(SMVP_ERR_INVALID, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        n = 0;
    *count = n;
    return SMVP_OK;
}

extern "C" int smvp_device_info(int device, char *name, size_t name_cap, int *compute_units, size_t *hbm_bytes)
{
    if (int rc = usable_device(device))
        return rc;
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_cap) {
        if (p.name[0])
            snprintf(name, name_cap, "%s (%s)", p.name, p.gcnArchName);
        else  // some driver stacks leave the marketing name empty
            snprintf(name, name_cap, "%s", p.gcnArchName);
    }
    if (compute_units)
        *compute_units = p.multiProcessorCount;
    if (hbm_bytes)
        *hbm_bytes = p.totalGlobalMem;
    return SMVP_OK;
}

// ===========================================================================
// reference-shaped entry points
// ===========================================================================
extern "C" void smvp_run_opts_default(smvp_run_opts_t *o)
{
    if (!o)
        return;
    memset(o, 0, sizeof *o);
    o->struct_size = (unsigned)sizeof *o;
    o->shard_exchange = SMVP_EXCHANGE_AUTO;
    o->csr_kernel = SMVP_CSR_KERNEL_AUTO;
    o->tjds_mode = SMVP_TJDS_MODE_AUTO;
    o->timing = SMVP_TIMING_AUTO;
}

namespace {

constexpr int kEventRing = 1024;

// Scope guard for the scratch the two entry points allocate.
struct RunScratch {
    std::vector<double> ms;   // per-product times, filled as the event ring is drained
    double *d_x = nullptr, *d_y = nullptr;
    double *d_result = nullptr;                          // where the last product was written (d_x or d_y when iterating)
    unsigned long long *d_norm = nullptr;                // scratch of the normalisation
    void *d_coo = nullptr;                               // convert_on_device: the uploaded COO
    int *d_i0 = nullptr, *d_i1 = nullptr, *d_i2 = nullptr;  // ... and the arrays built from it
    double *d_v = nullptr;
    std::vector<hipEvent_t> ev;
    hipStream_t stream = nullptr;
    smvp_csr_t *csr = nullptr;
    smvp_tjds_t *tjds = nullptr;
    ~RunScratch()
    {
        for (hipEvent_t e : ev)
            (void)hipEventDestroy(e);
        if (d_norm)
            (void)hipFree(d_norm);
        if (d_x)
            (void)hipFree(d_x);
        if (d_y)
            (void)hipFree(d_y);
        if (stream)
            (void)hipStreamDestroy(stream);
        smvp_csr_destroy(csr);
        smvp_tjds_destroy(tjds);
        for (void *p : {d_coo, (void *)d_i0, (void *)d_i1, (void *)d_i2, (void *)d_v})
            if (p)
                (void)hipFree(p);
    }
};

int prepare_run(RunScratch &s, int rows, int cols, int iters, const smvp_run_opts_t *o)
{
    HIP_TRY(hipStreamCreate(&s.stream));
    HIP_TRY(hipMalloc((void **)&s.d_x, sizeof(double) * (size_t)std::max(std::max(cols, rows), 1)));
    HIP_TRY(hipMalloc((void **)&s.d_y, sizeof(double) * (size_t)std::max(std::max(cols, rows), 1)));
    HIP_TRY(hipMalloc((void **)&s.d_norm, sizeof(unsigned long long)));
    s.d_result = s.d_y;
    if (o->x) {
        HIP_TRY(hipMemcpy(s.d_x, o->x, sizeof(double) * (size_t)cols, hipMemcpyHostToDevice));
    } else {
        // vectorInit(rows, onesVector, 1), main-cli.c:368-369 / :761-762
        hipError_t e = smvp::launch_fill(s.d_x, 1.0, std::max(cols, rows), s.stream);
        if (e != hipSuccess)
            return smvp::fail(SMVP_ERR_HIP, "fill launch failed: %s", hipGetErrorString(e));
    }
    // a ring of event pairs, drained every kEventRing products: -n may be in the millions
    s.ev.assign((size_t)std::min(iters, kEventRing) * 2, nullptr);
    for (auto &e : s.ev)
        HIP_TRY(hipEventCreate(&e));
    s.ms.assign((size_t)iters, 0.0);
    return SMVP_OK;
}

// events of product i
inline hipEvent_t &ev_start(RunScratch &s, int i) { return s.ev[(size_t)2 * (i % kEventRing)]; }
inline hipEvent_t &ev_stop(RunScratch &s, int i) { return s.ev[(size_t)2 * (i % kEventRing) + 1]; }

// after product i has been enqueued: when the ring is full (or i is the last product) wait and read it out
int drain_ring(RunScratch &s, int i, int iters)
{
    if ((i + 1) % kEventRing != 0 && i + 1 != iters)
        return SMVP_OK;
    HIP_TRY(hipStreamSynchronize(s.stream));
    for (int k = i - (i % kEventRing); k <= i; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev_start(s, k), ev_stop(s, k)));
        s.ms[(size_t)k] = (double)ms;
    }
    return SMVP_OK;
}

int finish_run(RunScratch &s, int rows, int iters, double *y, double *time_each_ms, smvp_time_stats_t *stats)
{
    HIP_TRY(hipStreamSynchronize(s.stream));
    if (time_each_ms)
        memcpy(time_each_ms, s.ms.data(), sizeof(double) * (size_t)iters);
    if (stats)
        smvp_time_stats(s.ms.data(), iters, stats);
    if (rows > 0)
        HIP_TRY(hipMemcpy(y, s.d_result, sizeof(double) * (size_t)rows, hipMemcpyDeviceToHost));
    return SMVP_OK;
}

int check_iterate(const smvp_run_opts_t *o, int rows, int cols)
{
    if (o->struct_size != (unsigned)sizeof(smvp_run_opts_t))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_run_opts_t of %u bytes, this library's has %u: initialise it with "
                                            "smvp_run_opts_default and build against this library's header",
                          o->struct_size, (unsigned)sizeof(smvp_run_opts_t));
    if (o->iterate && rows != cols)
        return smvp::fail(SMVP_ERR_INVALID, "power iteration needs a square matrix (%d x %d given)", rows, cols);
    if (o->timing < SMVP_TIMING_AUTO || o->timing > SMVP_TIMING_DEVICE_GRAPH)
        return smvp::fail(SMVP_ERR_INVALID, "unknown timing method %d", o->timing);
    return SMVP_OK;
}

thread_local smvp_run_info_t g_last_run = {SMVP_TIMING_EVENTS, 0, 0.0, 0.0, 0, 0};

double host_ms()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC_RAW, &t);  // the reference's clock, main-cli.c:408
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

// Per-product times of launches too short for an event pair to time.  The reference brackets its product with
// clock_gettime (main-cli.c:408-419): nothing but the product is inside the window.  A hipEvent pair around a launch
// of a few microseconds measures mostly the events themselves (an empty launch between two events reads 6.6 us on
// MI355X, memplus.mtx's CSR kernel runs 3.8 us), so for such launches the kernel times itself: every wave writes the
// constant-rate wall clock when it starts and when its last store has been acknowledged, stamp_reduce takes
// max(last) - min(first) per product.  The products of a run are captured into one hipGraph (kStampRing products per
// replay, their timing slots baked into the nodes) so that the host's launch rate is not what the run waits for.
constexpr int kStampRing = 256;        // products per graph replay (even: power iteration swaps x and y)
constexpr int kStampMaxSlots = 16384;  // waves per launch up to which the kernel times itself (4096 workgroups)

struct StampTimer {
    unsigned long long *d_stamps = nullptr, *d_first_last = nullptr;
    unsigned *d_ctl = nullptr;  // the repeating launch's barrier counters
    hipGraphExec_t exec = nullptr;
    int exec_n = 0;
    ~StampTimer()
    {
        if (exec)
            (void)hipGraphExecDestroy(exec);
        if (d_stamps)
            (void)hipFree(d_stamps);
        if (d_first_last)
            (void)hipFree(d_first_last);
        if (d_ctl)
            (void)hipFree(d_ctl);
    }
};

// `iters` products on s.stream, each timed on its own.  pre(y): work the reference keeps outside its window (clearing
// y); product(x, y, stamps): the launches of one product.  stamp_slots > 0: the product can time itself on the device.
// repeat_grid > 0: the product has a repeating form -- repeat(x, y, stamps, reps, grid, ctl_words) enqueues `reps` products as
// ONE launch that stamps every product's window (needs no `pre`); used for device-timed runs unless SMVP_TIMING_DEVICE_GRAPH asks
// for one launch per product.
template <class Pre, class Product, class Repeat>
int run_timed_products(RunScratch &s, int rows, int iters, const smvp_run_opts_t *o, int stamp_slots, Pre pre, Product product,
                       int repeat_grid, Repeat repeat)
{
    double *xc = s.d_x, *yc = s.d_y;
    const bool device_asked = o->timing == SMVP_TIMING_DEVICE || o->timing == SMVP_TIMING_DEVICE_GRAPH;
    const bool stamped = o->timing != SMVP_TIMING_EVENTS && !o->iterate && stamp_slots > 0 && (device_asked || stamp_slots <= kStampMaxSlots);
    if (device_asked && !stamped)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "device-side timing needs the tile kernel of one GPU and no --iterate");
    g_last_run.timing = stamped ? SMVP_TIMING_DEVICE : SMVP_TIMING_EVENTS;
    g_last_run.graph_replays = 0;
    g_last_run.repeat_launches = 0;
    g_last_run.repeat_gave_up = 0;
    const unsigned long long patience = smvp::repeat_patience_ticks(o->repeat_patience_us);
    HIP_TRY(hipStreamSynchronize(s.stream));
    const double t0 = host_ms();
    bool repeated = false;
    int khz = 0;
    if (stamped) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
        if (khz <= 0)
            return smvp::fail(SMVP_ERR_HIP, "the device reports no wall-clock rate");
        g_last_run.device_clock_khz = khz;
    }
    if (stamped && repeat_grid > 0 && o->timing != SMVP_TIMING_DEVICE_GRAPH) {
        // Up to kRepeatRing products per launch of the repeating kernel, the launches of a run enqueued one behind the other:
        // every launch's windows are reduced on the device into first_last[product] and its give-up word is set aside; the
        // host waits once per kRepeatSuper products -- and once for the FIRST launch of the run, before it queues any other.  A
        // launch that gave up at one of its barriers (the grid was not resident as a whole: another stream, thread or process on
        // the device -- the occupancy query the grid was sized from knows nothing of those) sends the whole run to the single
        // launches below: it has waited `repeat_patience_us` (50 ms) at most, and launches queued behind it read the run's
        // sticky give-up word and leave as they start.
        constexpr int kRepeatRing = 1024, kRepeatSuper = 1 << 20;
        StampTimer st;
        unsigned *d_tops = nullptr;
        const int slots = repeat_grid * (smvp::kStreamBlock / 64);
        const size_t per_product = (size_t)slots * 2;
        int ring = std::min(iters, kRepeatRing);
        while (ring > 64 && sizeof(unsigned long long) * per_product * (size_t)ring > (64u << 20))
            ring /= 2;  // (stamps of one launch: at most 64 MB)
        const int super = std::min(iters, kRepeatSuper), launches_per_super = (super + ring - 1) / ring;
        HIP_TRY(hipMalloc((void **)&st.d_stamps, sizeof(unsigned long long) * per_product * (size_t)ring));
        HIP_TRY(hipMalloc((void **)&st.d_first_last, sizeof(unsigned long long) * 2 * (size_t)super + sizeof(unsigned) * (size_t)launches_per_super));
        HIP_TRY(hipMalloc((void **)&st.d_ctl, sizeof(unsigned) * smvp::kRepeatCtlWords));
        d_tops = reinterpret_cast<unsigned *>(st.d_first_last + 2 * (size_t)super);
        std::vector<unsigned long long> fl((size_t)super * 2);
        std::vector<unsigned> tops((size_t)launches_per_super);
        repeated = true;
        for (int s0 = 0; s0 < iters && repeated; s0 += super) {
            const int ns = std::min(super, iters - s0);
            int launches = 0;
            for (int i0 = 0; i0 < ns; i0 += ring, ++launches) {
                const int n = std::min(ring, ns - i0);
                const bool first_of_run = s0 == 0 && i0 == 0;
                if (int rc = repeat(xc, yc, st.d_stamps, n, repeat_grid, st.d_ctl, first_of_run, patience))
                    return rc;
                HIP_TRY(smvp::launch_stamp_reduce(st.d_stamps, slots, n, st.d_first_last + 2 * (size_t)i0, s.stream));
                HIP_TRY(hipMemcpyAsync(d_tops + launches, st.d_ctl + smvp::kRepeatCtlWords - 32, sizeof(unsigned), hipMemcpyDeviceToDevice, s.stream));
                if (first_of_run && ns > n) {  // more launches would follow: has this one held?
                    HIP_TRY(hipMemcpyAsync(tops.data(), d_tops, sizeof(unsigned), hipMemcpyDeviceToHost, s.stream));
                    HIP_TRY(hipStreamSynchronize(s.stream));
                    if (tops[0] & 0x80000000u) {
                        repeated = false;
                        break;
                    }
                }
            }
            if (!repeated)
                break;
            HIP_TRY(hipMemcpyAsync(fl.data(), st.d_first_last, sizeof(unsigned long long) * 2 * (size_t)ns, hipMemcpyDeviceToHost, s.stream));
            HIP_TRY(hipMemcpyAsync(tops.data(), d_tops, sizeof(unsigned) * (size_t)launches, hipMemcpyDeviceToHost, s.stream));
            HIP_TRY(hipStreamSynchronize(s.stream));
            for (int l = 0; l < launches; ++l)
                if (tops[(size_t)l] & 0x80000000u)
                    repeated = false;  // gave up at a barrier: nothing of this run is trusted
            if (!repeated)
                break;
            g_last_run.repeat_launches += launches;
            for (int k = 0; k < ns; ++k)
                s.ms[(size_t)(s0 + k)] = (double)(fl[2 * (size_t)k + 1] - fl[2 * (size_t)k]) / (double)khz;
        }
        if (!repeated) {
            g_last_run.repeat_launches = 0;
            g_last_run.repeat_gave_up = 1;
        }
        s.d_result = yc;
    }
    if (stamped && !repeated) {
        StampTimer st;
        const size_t per_product = (size_t)stamp_slots * 2;
        const int ring = std::min(iters, kStampRing);
        HIP_TRY(hipMalloc((void **)&st.d_stamps, sizeof(unsigned long long) * per_product * (size_t)ring));
        HIP_TRY(hipMalloc((void **)&st.d_first_last, sizeof(unsigned long long) * 2 * (size_t)ring));
        std::vector<unsigned long long> fl((size_t)ring * 2);
        bool use_graph = true;
        for (int i0 = 0; i0 < iters; i0 += ring) {
            const int n = std::min(ring, iters - i0);
            auto enqueue = [&]() -> int {
                for (int k = 0; k < n; ++k) {
                    if (int rc = pre(yc))
                        return rc;
                    if (int rc = product(xc, yc, st.d_stamps + per_product * (size_t)k))
                        return rc;
                }
                return SMVP_OK;
            };
            if (use_graph && st.exec_n != n) {
                if (st.exec)
                    (void)hipGraphExecDestroy(st.exec);
                st.exec = nullptr;
                st.exec_n = 0;
                hipGraph_t graph = nullptr;
                bool ok = hipStreamBeginCapture(s.stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                int rc = ok ? enqueue() : SMVP_OK;
                if (ok)
                    ok = hipStreamEndCapture(s.stream, &graph) == hipSuccess && graph && rc == SMVP_OK;
                if (ok)
                    ok = hipGraphInstantiate(&st.exec, graph, nullptr, nullptr, 0) == hipSuccess;
                if (graph)
                    (void)hipGraphDestroy(graph);
                if (rc != SMVP_OK)
                    return rc;
                if (!ok) {  // no graph support for this sequence: plain launches, still timed on the device
                    (void)hipGetLastError();
                    st.exec = nullptr;
                    use_graph = false;
                } else {
                    st.exec_n = n;
                }
            }
            if (use_graph) {
                HIP_TRY(hipGraphLaunch(st.exec, s.stream));
                ++g_last_run.graph_replays;
            } else if (int rc = enqueue()) {
                return rc;
            }
            HIP_TRY(smvp::launch_stamp_reduce(st.d_stamps, stamp_slots, n, st.d_first_last, s.stream));
            HIP_TRY(hipMemcpyAsync(fl.data(), st.d_first_last, sizeof(unsigned long long) * 2 * (size_t)n,
                                   hipMemcpyDeviceToHost, s.stream));
            HIP_TRY(hipStreamSynchronize(s.stream));
            for (int k = 0; k < n; ++k)
                s.ms[(size_t)(i0 + k)] = (double)(fl[2 * (size_t)k + 1] - fl[2 * (size_t)k]) / (double)khz;
        }
        s.d_result = yc;
    } else if (!stamped) {
        for (int i = 0; i < iters; ++i) {
            if (int rc = pre(yc))
                return rc;
            HIP_TRY(hipEventRecord(ev_start(s, i), s.stream));
            if (int rc = product(xc, yc, nullptr))
                return rc;
            HIP_TRY(hipEventRecord(ev_stop(s, i), s.stream));
            if (o->iterate && o->normalize)  // scaling the iterate is not part of the product: outside the window,
                HIP_TRY(smvp::launch_normalize_max(yc, rows, s.d_norm, s.stream));  // on one GPU and on several alike
            if (int rc = drain_ring(s, i, iters))
                return rc;
            s.d_result = yc;
            if (o->iterate)
                std::swap(xc, yc);  // x_{k+1} = y_k
        }
    }
    HIP_TRY(hipStreamSynchronize(s.stream));
    g_last_run.wall_ms = host_ms() - t0;
    return SMVP_OK;
}

}  // namespace

extern "C" int smvp_last_run_info(smvp_run_info_t *out)
{
    if (!out)
        return smvp::fail(SMVP_ERR_INVALID, "null argument");
    *out = g_last_run;
    return SMVP_OK;
}

// opts.ngpus > 1: the same timed loop over row blocks on several GPUs; the window is the local products
// plus the all-gather of y, the longest GPU counts (smvp_sharded.hip)
static int sharded_compute(bool tjds, const smvp_coo_t *coo, int rows, int cols, int nnz, int iters,
                           const smvp_run_opts_t *o, double *y, double *time_each_ms, smvp_time_stats_t *stats)
{
    if (o->timing == SMVP_TIMING_DEVICE || o->timing == SMVP_TIMING_DEVICE_GRAPH)  // the in-kernel stamps time one launch of one GPU; a sharded product is several
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "device-side timing is not available with more than one GPU (use events)");
    smvp_sharded_t *h = nullptr;
    smvp_shard_opts_t so;
    smvp_shard_opts_default(&so);
    so.exchange = o->shard_exchange;
    int rc;
    if (tjds) {
        rc = smvp_tjds_sharded_create_ex(&h, o->ngpus, nullptr, coo, rows, cols, nnz, &so);
    } else {
        std::vector<int> row_ptr((size_t)rows + 1), col_ind((size_t)std::max(nnz, 1));
        std::vector<double> val((size_t)std::max(nnz, 1));
        rc = smvp_csr_from_coo(coo, rows, nnz, row_ptr.data(), col_ind.data(), val.data());
        if (rc == SMVP_OK)
            rc = smvp_csr_sharded_create_ex(&h, o->ngpus, nullptr, rows, cols, nnz, row_ptr.data(), col_ind.data(), val.data(), &so);
    }
    std::vector<double> local;
    if (!time_each_ms) {
        local.resize((size_t)iters);
        time_each_ms = local.data();
    }
    if (rc == SMVP_OK && !tjds && (o->csr_kernel != SMVP_CSR_KERNEL_AUTO || o->csr_param != 0))
        rc = smvp_sharded_set_csr_kernel(h, o->csr_kernel, o->csr_param);
    if (rc == SMVP_OK)
        rc = smvp_sharded_set_x(h, o->x);
    for (int i = 0; rc == SMVP_OK && i < iters; ++i) {
        rc = smvp_sharded_spmv(h, SMVP_GATHER_OVERLAPPED, 1);
        if (rc == SMVP_OK)
            rc = smvp_sharded_synchronize(h, &time_each_ms[i]);
        if (rc == SMVP_OK && o->iterate && (i + 1 < iters || o->normalize))
            rc = smvp_sharded_feed_back(h, o->normalize);  // the gathered y is the next operand on every GPU
    }
    if (rc == SMVP_OK)
        rc = smvp_sharded_get_y(h, 0, 1, y);
    if (rc == SMVP_OK && stats)
        smvp_time_stats(time_each_ms, iters, stats);
    smvp_sharded_destroy(h);
    return rc;
}

extern "C" int smvp_csr_compute(const smvp_coo_t *coo, int rows, int cols, int nnz, int iters,
                                const smvp_run_opts_t *opts, double *y, double *time_each_ms,
                                smvp_time_stats_t *stats)
{
    smvp_run_opts_t def;
    smvp_run_opts_default(&def);
    const smvp_run_opts_t *o = opts ? opts : &def;
    if (iters < 1 || rows < 0 || cols < 0 || nnz < 0 || (rows > 0 && !y))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_compute: bad argument");
    if (int rc = check_iterate(o, rows, cols))
        return rc;
    if (o->ngpus > 1)
        return sharded_compute(false, coo, rows, cols, nnz, iters, o, y, time_each_ms, stats);
    if (int rc = usable_device(o->device))
        return rc;
    DeviceScope on(o->device);

    RunScratch s;
    if (o->convert_on_device) {
        // COO goes to HBM as it is; sort + scan there; the arrays stay where they were built
        HIP_TRY(hipMalloc(&s.d_coo, sizeof(smvp_coo_t) * (size_t)std::max(nnz, 1)));
        HIP_TRY(hipMalloc((void **)&s.d_i0, sizeof(int) * ((size_t)rows + 1)));
        HIP_TRY(hipMalloc((void **)&s.d_i1, sizeof(int) * (size_t)std::max(nnz, 1)));
        HIP_TRY(hipMalloc((void **)&s.d_v, sizeof(double) * (size_t)std::max(nnz, 1)));
        if (nnz > 0)
            HIP_TRY(hipMemcpy(s.d_coo, coo, sizeof(smvp_coo_t) * (size_t)nnz, hipMemcpyHostToDevice));
        if (int rc = smvp_csr_from_coo_device((const smvp_coo_t *)s.d_coo, rows, cols, nnz, s.d_i0, s.d_i1, s.d_v, nullptr))
            return rc;
        if (int rc = smvp_csr_create(&s.csr, o->device, rows, cols, nnz, s.d_i0, s.d_i1, s.d_v, SMVP_MEM_DEVICE, nullptr))
            return rc;
    } else {
        std::vector<int> row_ptr((size_t)rows + 1), col_ind((size_t)std::max(nnz, 1));
        std::vector<double> val((size_t)std::max(nnz, 1));
        if (int rc = smvp_csr_from_coo(coo, rows, nnz, row_ptr.data(), col_ind.data(), val.data()))
            return rc;
        if (int rc = smvp_csr_create(&s.csr, o->device, rows, cols, nnz, row_ptr.data(), col_ind.data(), val.data(),
                                     SMVP_MEM_HOST, nullptr))
            return rc;
    }
    if (o->csr_kernel != SMVP_CSR_KERNEL_AUTO || o->csr_param != 0)
        if (int rc = smvp_csr_set_kernel(s.csr, o->csr_kernel, o->csr_param))
            return rc;
    if (int rc = prepare_run(s, rows, cols, iters, o))
        return rc;

    // The reference clears y before every product, outside its timed window (main-cli.c:405).  Every CSR kernel
    // here overwrites all of y, so nothing is cleared per product; y is poisoned with NaN once instead, so that a
    // kernel that skipped a row could not hide behind a cleared (or an earlier) result.
    HIP_TRY(hipMemsetAsync(s.d_y, 0xff, sizeof(double) * (size_t)std::max(rows, 1), s.stream));
    smvp_csr_t *A = s.csr;
    const int slots = csr_can_stamp(A) ? smvp::owner_stamp_slots(A->ntiles, A->flavor) : 0;
    const int rgrid = slots > 0 ? csr_repeat_grid(A) : 0;  // the tile kernel's repeating form: n products per launch
    if (int rc = run_timed_products(
            s, rows, iters, o, slots, [](double *) { return (int)SMVP_OK; },
            [A, &s](const double *x, double *yy, unsigned long long *stamps) { return csr_spmv_impl(A, x, yy, s.stream, stamps); }, rgrid,
            [A, &s](const double *x, double *yy, unsigned long long *stamps, int reps, int grid, unsigned *ctl, bool first, unsigned long long patience) {
                return csr_spmv_repeat(A, x, yy, s.stream, stamps, reps, grid, ctl, first, patience);
            }))
        return rc;
    return finish_run(s, rows, iters, y, time_each_ms, stats);
}

extern "C" int smvp_tjds_compute(const smvp_coo_t *coo, int rows, int cols, int nnz, int iters,
                                 const smvp_run_opts_t *opts, double *y, double *time_each_ms,
                                 smvp_time_stats_t *stats)
{
    smvp_run_opts_t def;
    smvp_run_opts_default(&def);
    const smvp_run_opts_t *o = opts ? opts : &def;
    if (iters < 1 || rows < 0 || cols < 0 || nnz < 0 || (rows > 0 && !y))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_compute: bad argument");
    if (int rc = check_iterate(o, rows, cols))
        return rc;
    if (o->iterate && o->tjds_ref_quirks)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "ref-quirks TJDS indexes the operand by row: it has no meaning for a changing operand");
    if (o->ngpus > 1) {
        if (o->tjds_ref_quirks)
            return smvp::fail(SMVP_ERR_UNSUPPORTED, "ref-quirks TJDS is a whole-matrix artefact: use one GPU");
        return sharded_compute(true, coo, rows, cols, nnz, iters, o, y, time_each_ms, stats);
    }
    if (int rc = usable_device(o->device))
        return rc;
    DeviceScope on(o->device);

    int num_diag = 0, ref_num = 0, last_single = 0;
    RunScratch s;
    if (o->convert_on_device) {
        const int cap = std::max(rows, nnz) + 2;
        HIP_TRY(hipMalloc(&s.d_coo, sizeof(smvp_coo_t) * (size_t)std::max(nnz, 1)));
        HIP_TRY(hipMalloc((void **)&s.d_i0, sizeof(int) * (size_t)std::max(cols, 1)));   // perm
        HIP_TRY(hipMalloc((void **)&s.d_i1, sizeof(int) * (size_t)std::max(nnz, 1)));    // row_ind
        HIP_TRY(hipMalloc((void **)&s.d_i2, sizeof(int) * (size_t)cap));                 // start_pos
        HIP_TRY(hipMalloc((void **)&s.d_v, sizeof(double) * (size_t)std::max(nnz, 1)));
        if (nnz > 0)
            HIP_TRY(hipMemcpy(s.d_coo, coo, sizeof(smvp_coo_t) * (size_t)nnz, hipMemcpyHostToDevice));
        if (int rc = smvp_tjds_from_coo_device((const smvp_coo_t *)s.d_coo, rows, cols, nnz, s.d_i0, s.d_i2, cap, s.d_i1,
                                               s.d_v, &num_diag, &ref_num, &last_single, nullptr))
            return rc;
        if (int rc = smvp_tjds_create(&s.tjds, o->device, rows, cols, nnz, num_diag, s.d_i0, s.d_i2, s.d_i1, s.d_v,
                                      SMVP_MEM_DEVICE))
            return rc;
    } else {
        std::vector<int> perm((size_t)std::max(cols, 1)), start_pos((size_t)std::max(rows, nnz) + 2),
            row_ind((size_t)std::max(nnz, 1));
        std::vector<double> val((size_t)std::max(nnz, 1));
        if (int rc = smvp_tjds_from_coo(coo, rows, cols, nnz, perm.data(), start_pos.data(), (int)start_pos.size(),
                                        row_ind.data(), val.data(), &num_diag, &ref_num, &last_single))
            return rc;
        if (int rc = smvp_tjds_create(&s.tjds, o->device, rows, cols, nnz, num_diag, perm.data(), start_pos.data(),
                                      row_ind.data(), val.data(), SMVP_MEM_HOST))
            return rc;
    }
    if (o->tjds_ref_quirks)
        if (int rc = smvp_tjds_set_ref_quirks(s.tjds, 1, ref_num, last_single))
            return rc;
    if (int rc = prepare_run(s, rows, cols, iters, o))
        return rc;
    if (int rc = smvp_tjds_set_x(s.tjds, s.d_x, s.stream))  // main-cli.c:907-923, setup
        return rc;

    if (o->tjds_mode != SMVP_TJDS_MODE_AUTO)
        if (int rc = smvp_tjds_set_mode(s.tjds, o->tjds_mode))
            return rc;
    HIP_TRY(hipMemsetAsync(s.d_y, 0xff, sizeof(double) * (size_t)std::max(rows, 1), s.stream));  // NaN, as in the CSR path
    smvp_tjds_t *T = s.tjds;
    const int slots = tjds_can_stamp(T) ? smvp::owner_stamp_slots(T->rg->ntiles, T->rg->flavor) : 0;
    bool first = true;
    const bool iterate = o->iterate != 0;
    if (int rc = run_timed_products(
            s, rows, iters, o, slots,
            [T, &s](double *yy) { return smvp_tjds_zero_y(T, yy, s.stream); },  // main-cli.c:1008, outside the window
            [T, &s, &first, iterate](const double *x, double *yy, unsigned long long *stamps) {
                if (iterate && !first)  // a new operand: its permutation is part of this product
                    if (int rc = smvp_tjds_set_x(T, x, s.stream))
                        return rc;
                first = false;
                return tjds_spmv_impl(T, yy, s.stream, stamps);
            },
            slots > 0 ? csr_repeat_grid(T->rg) : 0,  // (tjds_can_stamp: the row-gather product, which overwrites y and needs no `pre`)
            [T, &s](const double *, double *yy, unsigned long long *stamps, int reps, int grid, unsigned *ctl, bool first, unsigned long long patience) {
                return csr_spmv_repeat(T->rg, T->d_x_perm, yy, s.stream, stamps, reps, grid, ctl, first, patience);
            }))
        return rc;
    return finish_run(s, rows, iters, y, time_each_ms, stats);
}
