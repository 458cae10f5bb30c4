/*
 * smvp_oracle.h -- CPU restatement of smvp-toolkit's CSR / TJDS path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under smvp-toolkit_amd/ (the product) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker / the CPU baseline.
 *
 * Every function names the reference lines it restates (paths relative to the
 * reference checkout, v0.6.4).  The restatement is pinned against the eleven
 * report files the reference commits (the files in output-test/ and the one in build/); see
 * tests/test_oracle_golden.py.  The reference's main-cli.c itself cannot be
 * built in this image (it needs libpopt, which is absent), so there is no
 * oracle/_ref build of it; mmio/mmio.c is self-contained and is built into
 * oracle/_ref/libmmio_ref.so by oracle/Makefile to cross-check the reader.
 */
#ifndef SMVP_ORACLE_H
#define SMVP_ORACLE_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One stored Matrix Market entry, 0-based (main-cli.c:42-47, 1439-1440). */
typedef struct {
    int row;
    int col;
    double val;
} orc_coo;

/* mmio error numbers (mmio/mmio.h, "Matrix Market error codes"). */
enum {
    ORC_MM_OK = 0,
    ORC_MM_COULD_NOT_READ_FILE = 11,
    ORC_MM_PREMATURE_EOF = 12,
    ORC_MM_NOT_MTX = 13,
    ORC_MM_NO_HEADER = 14,
    ORC_MM_UNSUPPORTED_TYPE = 15
};

/* typecode[4] as mmio builds it: 'M', 'C'|'A', 'R'|'C'|'P'|'I', 'G'|'S'|'H'|'K'. */
int orc_mm_read_header(const char *path, char typecode[4], int *rows, int *cols, int *nnz);
/* Banner + size + the CLI's entry loop.  Returns 0, an ORC_MM_* code, or -1 (no file). */
int orc_mm_read_coo(const char *path, orc_coo *out, int cap, char typecode[4],
                    int *rows, int *cols, int *nnz);

/* CSR ------------------------------------------------------------------ */
void orc_sort_row_col(orc_coo *coo, int nnz);
void orc_csr_build(const orc_coo *coo, int rows, int nnz,
                   int *row_ptr, int *col_ind, double *val);
void orc_csr_build_literal(const orc_coo *coo, int rows, int nnz,
                           int *row_ptr, int *col_ind, double *val);
void orc_csr_spmv(int rows, const int *row_ptr, const int *col_ind, const double *val,
                  const double *x, double *y);

void orc_csr_iterate(int rows, const int *row_ptr, const int *col_ind, const double *val,
                     const double *x0, int iters, int normalize, double *y);

/* TJDS ----------------------------------------------------------------- */
/* perm[cols], start_pos[max_diag+1 needed, caller gives rows+2], row_ind[nnz],
 * val[nnz].  *num_diag = D (longest column); *ref_num_tjdiag = the count the
 * reference derives at main-cli.c:865 (length of original column 0);
 * *last_diag_single = 1 when the reference would leave start_pos[D] unwritten. */
int orc_tjds_build(const orc_coo *coo, int rows, int cols, int nnz,
                   int *perm, int *start_pos, int *row_ind, double *val,
                   int *num_diag, int *ref_num_tjdiag, int *last_diag_single);
void orc_tjds_spmv(int rows, int cols, int num_diag, const int *perm, const int *start_pos,
                   const int *row_ind, const double *val, const double *x, double *y);
void orc_tjds_spmv_refquirks(int rows, int cols, int num_diag, int ref_num_tjdiag,
                             int last_diag_single, const int *perm, const int *start_pos,
                             const int *row_ind, const double *val, const double *x, double *y);

/* stats + report -------------------------------------------------------- */
typedef struct {
    double total, avg, stdev, min, max;
} orc_stats;
void orc_time_stats(const double *ms, int n, orc_stats *out);
int orc_write_report(const char *full_path, const char *alg_name, unsigned long unix_time,
                     const char *input_name, int nnz, int rows, int iters,
                     const double *y, const orc_stats *st);

/* CPU baseline: the reference's timed loops, one thread ------------------ */
void orc_csr_timed(int rows, const int *row_ptr, const int *col_ind, const double *val,
                   const double *x, double *y, int iters, double *ms_each);
void orc_tjds_timed(int rows, int cols, int num_diag, const int *perm, const int *start_pos,
                    const int *row_ind, const double *val, const double *x, double *y,
                    int iters, double *ms_each);

/* CISR .coe generator, main-cli.c:473-729 (PARITY UNPINNED: no reference output exists).  0 = written,
 * 1 = the reference's "slot_group_iter overran" exit. */
int orc_cisr_coegen(const orc_coo *coo, int rows, int nnz, int slots, FILE *out);
int orc_cisr_coegen_path(const orc_coo *coo, int rows, int nnz, int slots, const char *path);

#ifdef __cplusplus
}
#endif
#endif
