// smvp_synth.cpp -- synthetic CSR workloads for the benchmark configurations.
//
// The reference ships only small sample matrices (sample-data/*.mtx) and cannot
// hold a large one at all (stack VLA at main-cli.c:1426).  BASELINE.json's
// configs 2-4 need HBM-sized inputs, so they are generated here, straight into
// CSR, as a pure function of (kind, seed, global row): any row block can be
// produced on its own, which is what row-block sharding across GPUs needs.
//
//   MEMPLUS_SHAPED  row lengths drawn from memplus.mtx's exact row-length
//                   histogram (17758 rows, mean 7.10, max 574); every row holds
//                   its diagonal; off-diagonal distances follow memplus's
//                   measured band profile (cumulative share of all entries with
//                   |row-col| <= 8 / 64 / 512 / 4096: 29.5 / 33.4 / 42.1 / 61.5 %,
//                   fitted separately for short and long rows), the remainder
//                   uniform over all columns.
//   UNIFORM         `param` entries per row, distinct uniform columns.
// Values are uniform in [-1, 1).  Columns are sorted and distinct inside a row.
#include "smvp_common.h"

#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace {

inline uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// Independent streams per (row, purpose, index).
inline uint64_t draw(uint64_t seed, int64_t row, uint32_t stream, uint64_t idx)
{
    return mix64(mix64(seed ^ ((uint64_t)row * 0xd1342543de82ef95ull)) + ((uint64_t)stream << 56) + idx);
}

// memplus.mtx row-length histogram: {length, rows with that length}; 17758 rows.
const int kMemplusHist[][2] = {
    {2, 52},   {3, 5036}, {4, 7323}, {5, 2458}, {6, 139},  {7, 140},  {8, 59},   {9, 54},   {10, 76},
    {11, 35},  {12, 1026}, {13, 1050}, {14, 3},  {15, 5},   {16, 6},   {17, 8},   {18, 11},  {19, 7},
    {20, 8},   {21, 5},   {22, 11},  {23, 6},   {24, 5},   {25, 5},   {27, 3},   {28, 3},   {29, 4},
    {30, 5},   {32, 3},   {33, 3},   {34, 7},   {35, 1},   {36, 2},   {37, 2},   {38, 3},   {39, 3},
    {42, 1},   {43, 1},   {45, 2},   {46, 2},   {47, 2},   {48, 2},   {49, 2},   {51, 1},   {52, 1},
    {54, 1},   {55, 1},   {56, 1},   {57, 2},   {58, 2},   {60, 1},   {61, 2},   {62, 1},   {64, 1},
    {70, 2},   {71, 1},   {74, 3},   {76, 1},   {77, 1},   {81, 1},   {84, 1},   {86, 1},   {88, 1},
    {94, 1},   {95, 1},   {104, 1},  {109, 1},  {112, 1},  {113, 2},  {116, 2},  {118, 1},  {119, 1},
    {121, 1},  {124, 1},  {126, 2},  {129, 1},  {131, 1},  {132, 1},  {134, 1},  {136, 1},  {140, 1},
    {153, 1},  {155, 1},  {160, 3},  {162, 1},  {164, 1},  {165, 1},  {166, 2},  {167, 1},  {168, 2},
    {169, 2},  {170, 3},  {171, 5},  {172, 4},  {174, 1},  {175, 2},  {176, 2},  {177, 1},  {185, 3},
    {188, 1},  {189, 1},  {193, 1},  {196, 1},  {197, 1},  {198, 1},  {200, 1},  {202, 1},  {204, 1},
    {208, 2},  {210, 1},  {211, 1},  {212, 3},  {213, 1},  {214, 3},  {215, 1},  {216, 3},  {218, 2},
    {219, 1},  {220, 1},  {221, 4},  {222, 6},  {223, 7},  {224, 4},  {225, 3},  {226, 1},  {227, 3},
    {228, 2},  {256, 1},  {261, 1},  {262, 1},  {274, 1},  {343, 1},  {344, 2},  {345, 3},  {346, 5},
    {347, 5},  {348, 4},  {349, 4},  {350, 4},  {351, 4},  {430, 1},  {574, 1}};
constexpr int kMemplusRows = 17758;

struct LengthTable {
    std::vector<int> by_slot;  // by_slot[u] for u in [0, 17758): the inverse CDF
    LengthTable()
    {
        by_slot.reserve(kMemplusRows);
        for (auto &e : kMemplusHist)
            by_slot.insert(by_slot.end(), (size_t)e[1], e[0]);
    }
};
const LengthTable &length_table()
{
    static const LengthTable t;
    return t;
}

// Off-diagonal distance classes, in 1/1000 of a row's off-diagonal entries,
// measured on memplus separately for its short rows (<= 64 entries: 72 % of the
// entries, 39 % of them within 8 of the diagonal) and its long rows (the 165
// rows > 64: hardly anything near the diagonal, 77 % within 4096).
struct Band { int per_mille; int64_t lo, hi; };  // distance in [lo, hi]; the remainder is uniform
const Band kBandsShort[] = {{248, 1, 8}, {18, 9, 64}, {33, 65, 512}, {149, 513, 4096}};
const Band kBandsLong[] = {{34, 1, 8}, {103, 9, 64}, {241, 65, 512}, {388, 513, 4096}};
constexpr int kLongRowLen = 64;

int row_length(int kind, uint64_t seed, int64_t cols_total, int param, int64_t row)
{
    int64_t len;
    if (kind == SMVP_SYNTH_MEMPLUS_SHAPED)
        len = length_table().by_slot[(size_t)(draw(seed, row, 0, 0) % kMemplusRows)];
    else
        len = param;
    return (int)std::min<int64_t>(len, cols_total);
}

// One candidate column for entry k of `row`, attempt a.  Negative = out of range.
int64_t candidate(int kind, uint64_t seed, int64_t cols_total, int64_t row, int len, int k, int attempt)
{
    const uint64_t u = draw(seed, row, 1, (uint64_t)k * 64 + (uint64_t)attempt);
    if (kind == SMVP_SYNTH_MEMPLUS_SHAPED && attempt < 4) {
        int cls = (int)(u % 1000);
        const uint64_t v = u / 1000;
        const Band *bands = len > kLongRowLen ? kBandsLong : kBandsShort;
        for (int bi = 0; bi < 4; ++bi) {
            const Band &b = bands[bi];
            if (cls < b.per_mille) {
                const int64_t dist = b.lo + (int64_t)((v >> 1) % (uint64_t)(b.hi - b.lo + 1));
                return (v & 1) ? row + dist : row - dist;
            }
            cls -= b.per_mille;
        }
        return (int64_t)(v % (uint64_t)cols_total);
    }
    return (int64_t)(u % (uint64_t)cols_total);
}

void fill_row(int kind, uint64_t seed, int64_t cols_total, int64_t row, int len, int *cols, double *vals)
{
    int have = 0;
    if (kind == SMVP_SYNTH_MEMPLUS_SHAPED && len > 0 && row < cols_total)
        cols[have++] = (int)row;  // memplus stores every diagonal entry
    for (int k = have; k < len; ++k) {
        for (int attempt = 0;; ++attempt) {
            const int64_t c = candidate(kind, seed, cols_total, row, len, k, attempt);
            if (c < 0 || c >= cols_total)
                continue;
            bool dup = false;
            for (int i = 0; i < k && !dup; ++i)
                dup = (cols[i] == (int)c);
            if (!dup) {
                cols[k] = (int)c;
                break;
            }
        }
    }
    std::sort(cols, cols + len);
    for (int k = 0; k < len; ++k) {
        const uint64_t u = draw(seed, row, 2, (uint64_t)(uint32_t)cols[k]);
        vals[k] = (double)(u >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}

int check_args(int kind, int64_t rows_total, int64_t cols_total, int param, int64_t r0, int64_t r1)
{
    if (kind != SMVP_SYNTH_MEMPLUS_SHAPED && kind != SMVP_SYNTH_UNIFORM)
        return smvp::fail(SMVP_ERR_INVALID, "synth: unknown kind %d", kind);
    if (rows_total < 0 || cols_total < 1 || cols_total > INT32_MAX || r0 < 0 || r1 < r0 || r1 > rows_total)
        return smvp::fail(SMVP_ERR_INVALID, "synth: bad shape / row range");
    if (kind == SMVP_SYNTH_UNIFORM && param < 0)
        return smvp::fail(SMVP_ERR_INVALID, "synth: uniform kind needs param = entries per row");
    return SMVP_OK;
}

}  // namespace

extern "C" int smvp_synth_row_lengths(int kind, uint64_t seed, int64_t rows_total, int64_t cols_total,
                                      int param, int64_t row_begin, int64_t row_end, int *lens)
{
    if (int rc = check_args(kind, rows_total, cols_total, param, row_begin, row_end))
        return rc;
    if (row_end > row_begin && !lens)
        return smvp::fail(SMVP_ERR_INVALID, "synth: null lens");
    for (int64_t r = row_begin; r < row_end; ++r)
        lens[r - row_begin] = row_length(kind, seed, cols_total, param, r);
    return SMVP_OK;
}

extern "C" int smvp_synth_fill(int kind, uint64_t seed, int64_t rows_total, int64_t cols_total,
                               int param, int64_t row_begin, int64_t row_end, const int *row_ptr,
                               int *col_ind, double *val, int threads)
{
    if (int rc = check_args(kind, rows_total, cols_total, param, row_begin, row_end))
        return rc;
    const int64_t nrows = row_end - row_begin;
    if (nrows > 0 && (!row_ptr || !col_ind || !val))
        return smvp::fail(SMVP_ERR_INVALID, "synth: null output");
    for (int64_t i = 0; i < nrows; ++i)
        if (row_ptr[i + 1] - row_ptr[i] != row_length(kind, seed, cols_total, param, row_begin + i))
            return smvp::fail(SMVP_ERR_INVALID, "synth: row_ptr does not match smvp_synth_row_lengths at local row %lld",
                              (long long)i);
    if (threads < 1)
        threads = 1;
    auto work = [&](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i)
            fill_row(kind, seed, cols_total, row_begin + i, row_ptr[i + 1] - row_ptr[i],
                     col_ind + row_ptr[i], val + row_ptr[i]);
    };
    if (threads == 1 || nrows < 4096) {
        work(0, nrows);
        return SMVP_OK;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back(work, nrows * t / threads, nrows * (t + 1) / threads);
    for (auto &th : pool)
        th.join();
    return SMVP_OK;
}

// x[i] = the top 53 bits of mix64(seed + i) (the splitmix64 finaliser) * 2^-53: uniform in [0, 1), a pure function of
// (seed, i).  The command line's --x random uses seed 67890; numpy restatement in tests/test_cli_cpu.py.
extern "C" int smvp_vector_random(double *x, int64_t n, uint64_t seed)
{
    if (n < 0 || (n > 0 && !x))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_vector_random: bad argument");
    for (int64_t i = 0; i < n; ++i)
        x[i] = (double)(mix64(seed + (uint64_t)i) >> 11) * (1.0 / 9007199254740992.0);
    return SMVP_OK;
}
