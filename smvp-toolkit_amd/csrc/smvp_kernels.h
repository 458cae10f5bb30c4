// smvp_kernels.h -- internal interface between the kernel TU and the engine TU.
#pragma once
#include <hip/hip_runtime.h>

namespace smvp {

constexpr int kVectorBlock = 256;   // threads per block, csr_vector_rows
constexpr int kStreamBlock = 256;   // threads per block, csr_stream_tiles
constexpr int kLongRow = 32;        // segments longer than this are summed by a wavefront
constexpr int kStreamOver = 1024;    // entries past its end a tile may finish its last row with, through LDS
constexpr int kStreamTileGroup = 64;  // consecutive tiles per XCD turn (see tile_of_block)
constexpr int kTjdsBlock = 256;     // permuted columns per work item
constexpr int kTjdsDiagChunk = 8;   // jagged diagonals per work item

hipError_t launch_csr_vector(int lanes_per_row, const int *row_ptr, const int *col_ind, const double *val,
                             const double *x, double *y, int rows, hipStream_t stream);
hipError_t launch_csr_stream(int vpt, const int *row_ptr, const int *col_ind, const double *val,
                             const double *x, double *y, const int *tile_row, const int *carry_row,
                             double *carry, int rows, int nnz, int ntiles, hipStream_t stream);
hipError_t launch_csr_stream_owner(int vpt, bool unit_values, const int *row_ptr, const int *col_ind,
                                   const double *val, const double *x, double *y, const int *tile_row,
                                   const int *tile_next, int rows, int nnz, int ntiles, hipStream_t stream);
hipError_t launch_tjds_products(const int *start_pos, const double *val, const double *x_perm, double *prod,
                                const int4 *work, int nwork, int cols, hipStream_t stream);
hipError_t launch_tjds_scatter(bool operand_by_row, const int *start_pos, const int *row_ind, const double *val,
                               const double *x_perm, double *y, const int4 *work, int nwork, int cols,
                               hipStream_t stream);
hipError_t launch_tjds_permute(const int *perm, const double *x, double *x_perm, int cols, hipStream_t stream);
hipError_t launch_find_out_of_range(const int *a, long long n, int limit, int *bad, hipStream_t stream);
hipError_t launch_normalize_max(double *v, long long n, unsigned long long *scratch, hipStream_t stream);
hipError_t launch_fill(double *p, double v, long long n, hipStream_t stream);

}  // namespace smvp
