// smvp_common.h -- internal helpers shared by the host-side translation units.
#pragma once
#include "smvp_amd.h"

#include <cstdarg>
#include <cstdio>

namespace smvp {
// Records a message for smvp_last_error() and hands back `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();
int option(const char *name, int fallback);  // smvp_set_option's value, or `fallback` where it is not set (smvp_error.cpp)
}  // namespace smvp

// internal, device side (smvp_convert_device.hip): see its definition
struct ihipStream_t;
namespace smvp {
int build_row_inverse(const int *d_row_ind, int nnz, int rows, int *d_inv_ptr, int *d_inv_pos, ihipStream_t *stream);
int build_row_gather_plan(const int *d_row_ind, const int *d_start_pos, int num_diag, int nnz, int rows,
                          int *d_seg_ptr, int *d_pos, int *d_kcol, ihipStream_t *stream);
int build_colsweep_plan(const int *d_row_ptr, const int *d_col_ind, const double *d_val, int rows, int cols, int nnz, int strip_rows,
                        int parts, int chunk, int row_bits, int turn_cap, long long *d_strip_ptr, int *d_e_col, double *d_e_val,
                        unsigned short *d_e_row, ihipStream_t *stream);
int sort_tile_windows(const int *d_pos, int nnz, int tile, const int *d_start_pos, int num_diag, int slot_bits,
                      const double *d_val, int cache_min_tiles, int *d_pos_sorted, int *d_meta, int *d_cache_ptr,
                      double **d_val_cache, int *cached_total, ihipStream_t *stream);
int build_tile_half_streams(const int *d_pos, int nnz, int tile, const int *d_start_pos, int num_diag, const double *d_val,
                            int cache_min_tiles, unsigned *d_word32, int *d_cache_ptr, int *d_run_ptr,
                            double **d_val_cache, int **d_run_tab, unsigned short **d_group_run, int *cached_total,
                            int *runs_total, ihipStream_t *stream);
int build_column_offsets(const int *d_col_ind, int nnz, int tile, int *d_col_base, unsigned short *d_col16, int *fits,
                         ihipStream_t *stream);
int build_tile_overflow(const int *d_pos, const int *d_ovf_ptr, int total, int ntiles, int tile, int nnz,
                        const int *d_start_pos, int num_diag, const double *d_val, double *d_ovf_val, int *d_ovf_k, int positions,
                        ihipStream_t *stream);
}
