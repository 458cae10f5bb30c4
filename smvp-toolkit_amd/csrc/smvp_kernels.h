// smvp_kernels.h -- internal interface between the kernel TU and the engine TU.
#pragma once
#include <hip/hip_runtime.h>

namespace smvp {

constexpr int kVectorBlock = 256;   // threads per block, csr_vector_rows
constexpr int kStreamBlock = 256;   // threads per block, csr_stream_tiles
constexpr int kLongRow = 32;        // segments longer than this are summed by a wavefront
constexpr int kStreamOver = 1024;    // entries past its end a tile may finish its last row with, through LDS
constexpr int kStreamTileGroup = 64;  // consecutive tiles per XCD turn (see tile_of_block)
constexpr int kTjdsTileGroup = 16;    // ... for the tile-ordered TJDS stream
constexpr int kSweepBlock = 256;   // threads per block, csr_colsweep: four wavefronts, each with its own strip of rows
constexpr int kSweepWaves = kSweepBlock / 64;
constexpr int kSweepUnroll = 4;    // stream entries per lane and pass: a wavefront takes its strip 256 entries at a time
constexpr int kSweepChunk = 64 * kSweepUnroll;
constexpr int kSweepRowBits = 11;  // a strip holds at most 2048 rows (16 KB of sums in LDS) ...
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
constexpr int kFlavorUnit = 1;   // x[col_ind[j]]
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
    const int *ovf_ptr = nullptr, *ovf_pos = nullptr, *ovf_k = nullptr;  // kFlavorTjdsS
    const int *cache_ptr = nullptr;                                      // kFlavorTjdsS
    const double *val_cache = nullptr;
    const unsigned short *col16 = nullptr;                               // kFlavorCsr16
    const int *col_base = nullptr;
    const unsigned short *meta16 = nullptr, *group_run = nullptr;        // kFlavorTjdsH
    const int *run_ptr = nullptr, *run_sp = nullptr;
    int rows = 0, nnz = 0, ntiles = 0;
};
hipError_t launch_csr_stream_owner(int vpt, int flavor, const OwnerLaunch &l, hipStream_t stream);
int owner_stamp_slots(int ntiles, int flavor);  // {first, last} tick pairs one stamped launch writes
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
                               const double *x, double *y, int rows, int strip_rows, int per_launch, hipStream_t stream);
int sweep_chunks_in_flight(int strip_rows);  // the G of csr_colsweep<G> a strip height runs with
// entries of 64 K-entry samples of a CSR matrix that gather from distinct 128-byte lines of x (see csr_line_spread)
constexpr int kSpreadSpan = 64 * 1024;
hipError_t launch_csr_line_spread(const int *col_ind, long long nnz, int samples, int *distinct, hipStream_t stream);

}  // namespace smvp
