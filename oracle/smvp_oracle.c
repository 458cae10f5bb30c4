/*
 * smvp_oracle.c -- CPU restatement of smvp-toolkit's CSR / TJDS path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see smvp_oracle.h).  Written from the behaviour of
 * the reference, not from its text; each function cites the lines it follows.
 * Pinning: tests/test_oracle_golden.py checks the y vectors printed by this
 * file (with "%g", like main-cli.c:308) against every report the reference
 * commits, including the defective TJDS ones via orc_tjds_spmv_refquirks().
 *
 * Build: oracle/Makefile (gcc -O3 -DNDEBUG, the reference's Release flags,
 * build/build.ninja:111).  No -ffast-math, no FMA contraction: the sums must
 * round exactly like the reference's x86-64 build.
 */
#define _POSIX_C_SOURCE 200809L
#include "smvp_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef CLOCK_MONOTONIC_RAW
#define CLOCK_MONOTONIC_RAW 4
#endif

/* ---------------------------------------------------------------------------
 * Matrix Market header: mmio/mmio.c:96-170 (banner) and :180-208 (size line).
 * ------------------------------------------------------------------------- */
#define ORC_LINE 1025 /* MM_MAX_LINE_LENGTH, mmio/mmio.h */
#define ORC_TOK 64    /* MM_MAX_TOKEN_LENGTH */

static void lower_inplace(char *s)
{
    for (; *s; ++s)
        *s = (char)tolower((unsigned char)*s);
}

static int read_banner(FILE *f, char tc[4])
{
    char line[ORC_LINE];
    char tok[5][ORC_LINE]; /* the reference uses 64-byte tokens and can overflow; we cannot */

    tc[0] = tc[1] = tc[2] = ' ';
    tc[3] = 'G'; /* mm_clear_typecode */

    if (!fgets(line, sizeof line, f))
        return ORC_MM_PREMATURE_EOF; /* mmio.c:108 */
    if (sscanf(line, "%s %s %s %s %s", tok[0], tok[1], tok[2], tok[3], tok[4]) != 5)
        return ORC_MM_PREMATURE_EOF; /* mmio.c:111 */
    for (int i = 1; i < 5; ++i)
        lower_inplace(tok[i]); /* the banner token itself stays case-sensitive */

    if (strncmp(tok[0], "%%MatrixMarket", 14) != 0)
        return ORC_MM_NO_HEADER; /* mmio.c:125 */
    if (strcmp(tok[1], "matrix") != 0)
        return ORC_MM_UNSUPPORTED_TYPE;
    tc[0] = 'M';

    if (strcmp(tok[2], "coordinate") == 0)
        tc[1] = 'C';
    else if (strcmp(tok[2], "array") == 0)
        tc[1] = 'A';
    else
        return ORC_MM_UNSUPPORTED_TYPE;

    if (strcmp(tok[3], "real") == 0)
        tc[2] = 'R';
    else if (strcmp(tok[3], "complex") == 0)
        tc[2] = 'C';
    else if (strcmp(tok[3], "pattern") == 0)
        tc[2] = 'P';
    else if (strcmp(tok[3], "integer") == 0)
        tc[2] = 'I';
    else
        return ORC_MM_UNSUPPORTED_TYPE;

    if (strcmp(tok[4], "general") == 0)
        tc[3] = 'G';
    else if (strcmp(tok[4], "symmetric") == 0)
        tc[3] = 'S';
    else if (strcmp(tok[4], "hermitian") == 0)
        tc[3] = 'H';
    else if (strcmp(tok[4], "skew-symmetric") == 0)
        tc[3] = 'K';
    else
        return ORC_MM_UNSUPPORTED_TYPE;
    return ORC_MM_OK;
}

static int read_crd_size(FILE *f, int *m, int *n, int *nz)
{
    char line[ORC_LINE];
    *m = *n = *nz = 0;
    do {
        if (!fgets(line, sizeof line, f))
            return ORC_MM_PREMATURE_EOF; /* mmio.c:191 */
    } while (line[0] == '%');
    if (sscanf(line, "%d %d %d", m, n, nz) == 3)
        return ORC_MM_OK;
    /* blank line after the comments: keep scanning the stream (mmio.c:200-205).
     * The reference spins forever on a non-numeric token; we step over it. */
    for (;;) {
        int got = fscanf(f, "%d %d %d", m, n, nz);
        if (got == EOF)
            return ORC_MM_PREMATURE_EOF;
        if (got == 3)
            return ORC_MM_OK;
        if (fgetc(f) == EOF)
            return ORC_MM_PREMATURE_EOF;
    }
}

int orc_mm_read_header(const char *path, char typecode[4], int *rows, int *cols, int *nnz)
{
    FILE *f = fopen(path, "r");
    if (!f)
        return -1;
    int rc = read_banner(f, typecode);
    if (rc == ORC_MM_OK)
        rc = read_crd_size(f, rows, cols, nnz);
    fclose(f);
    return rc;
}

/* main-cli.c:1405-1441: banner, size, then one fscanf per stored entry.
 * Pattern files carry no value (val = 1); every other field type is read with
 * %lg; 1-based -> 0-based; symmetric storage is NOT mirrored. */
int orc_mm_read_coo(const char *path, orc_coo *out, int cap, char typecode[4],
                    int *rows, int *cols, int *nnz)
{
    FILE *f = fopen(path, "r");
    if (!f)
        return -1;
    int rc = read_banner(f, typecode);
    if (rc == ORC_MM_OK)
        rc = read_crd_size(f, rows, cols, nnz);
    if (rc != ORC_MM_OK) {
        fclose(f);
        return rc;
    }
    if (*nnz > cap) {
        fclose(f);
        return ORC_MM_COULD_NOT_READ_FILE;
    }
    const int pattern = (typecode[2] == 'P');
    for (int i = 0; i < *nnz; ++i) {
        int r = 0, c = 0, got;
        double v = 1.0;
        if (pattern)
            got = fscanf(f, "%d %d\n", &r, &c) - 2;
        else
            got = fscanf(f, "%d %d %lg\n", &r, &c, &v) - 3;
        if (got != 0) { /* the reference does not check; its entries would be garbage */
            fclose(f);
            return ORC_MM_PREMATURE_EOF;
        }
        out[i].row = r - 1;
        out[i].col = c - 1;
        out[i].val = v;
    }
    fclose(f);
    return ORC_MM_OK;
}

/* ---------------------------------------------------------------------------
 * CSR: main-cli.c:171-185 (comparator), :340 (qsort), :343-365 (build),
 *      :402-420 (timed product).
 * ------------------------------------------------------------------------- */
static int cmp_row_col(const void *a, const void *b)
{
    const orc_coo *p = (const orc_coo *)a, *q = (const orc_coo *)b;
    if (p->row != q->row)
        return p->row < q->row ? -1 : 1;
    if (p->col != q->col)
        return p->col < q->col ? -1 : 1;
    return 0;
}

void orc_sort_row_col(orc_coo *coo, int nnz)
{
    qsort(coo, (size_t)nnz, sizeof *coo, cmp_row_col);
}

/* Standard counting form.  Equal to the reference's row_ptr whenever no row is
 * empty (all five runnable samples); for empty rows the reference leaves
 * row_ptr unwritten, which has no defined meaning to reproduce. */
void orc_csr_build(const orc_coo *coo_in, int rows, int nnz,
                   int *row_ptr, int *col_ind, double *val)
{
    orc_coo *coo = (orc_coo *)malloc(sizeof *coo * (size_t)(nnz ? nnz : 1));
    memcpy(coo, coo_in, sizeof *coo * (size_t)nnz);
    orc_sort_row_col(coo, nnz);
    for (int r = 0; r <= rows; ++r)
        row_ptr[r] = 0;
    for (int i = 0; i < nnz; ++i) {
        val[i] = coo[i].val;
        col_ind[i] = coo[i].col;
        row_ptr[coo[i].row + 1] += 1;
    }
    for (int r = 0; r < rows; ++r)
        row_ptr[r + 1] += row_ptr[r];
    free(coo);
}

/* The reference's own branch structure (main-cli.c:348-365), run on a
 * zero-filled row_ptr (what a fresh glibc malloc hands it).  Kept to show that
 * it and orc_csr_build() agree on every sample matrix. */
void orc_csr_build_literal(const orc_coo *coo_in, int rows, int nnz,
                           int *row_ptr, int *col_ind, double *val)
{
    orc_coo *coo = (orc_coo *)malloc(sizeof *coo * (size_t)(nnz ? nnz : 1));
    memcpy(coo, coo_in, sizeof *coo * (size_t)nnz);
    orc_sort_row_col(coo, nnz);
    for (int r = 0; r <= rows; ++r)
        row_ptr[r] = 0;
    for (int i = 0; i < nnz; ++i) {
        val[i] = coo[i].val;
        col_ind[i] = coo[i].col;
        if (i == nnz - 1)
            row_ptr[coo[i].row + 1] = nnz;
        else if (coo[i].row < coo[i + 1].row)
            row_ptr[coo[i].row + 1] = i + 1;
        else if (i == 0)
            row_ptr[coo[i].row] = 0;
    }
    free(coo);
}

/* main-cli.c:410-416.  y starts at zero (vectorInit at :405) and the products
 * are added left to right, one rounding for the multiply and one for the add. */
void orc_csr_spmv(int rows, const int *row_ptr, const int *col_ind, const double *val,
                  const double *x, double *y)
{
    for (int r = 0; r < rows; ++r) {
        double acc = 0.0;
        for (int j = row_ptr[r]; j < row_ptr[r + 1]; ++j)
            acc += val[j] * x[col_ind[j]];
        y[r] = acc;
    }
}

/* Power iteration x <- A x (the product the assignment asked for according to the comment at
 * main-cli.c:401 -- the reference itself repeats y = A x with the same x).  No reference output exists
 * for it: parity of this mode is UNPINNED, the oracle only restates the serial definition.  With
 * `normalize` every iterate is divided by its largest magnitude. */
void orc_csr_iterate(int rows, const int *row_ptr, const int *col_ind, const double *val,
                     const double *x0, int iters, int normalize, double *y)
{
    double *x = (double *)malloc(sizeof *x * (size_t)(rows ? rows : 1));
    memcpy(x, x0, sizeof *x * (size_t)rows);
    for (int it = 0; it < iters; ++it) {
        orc_csr_spmv(rows, row_ptr, col_ind, val, x, y);
        if (normalize) {
            double m = 0.0;
            for (int r = 0; r < rows; ++r)
                if (fabs(y[r]) > m)
                    m = fabs(y[r]);
            if (m > 0.0)
                for (int r = 0; r < rows; ++r)
                    y[r] = y[r] / m;
        }
        memcpy(x, y, sizeof *x * (size_t)rows);
    }
    free(x);
}

/* ---------------------------------------------------------------------------
 * TJDS: main-cli.c:190-242 (comparators), :766 (sort by column), :789-826
 * (vertical compression), :845-868 (column table + sort), :894-904 (column
 * renumbering), :926-967 (diagonal-major emit), :1004-1024 (timed product).
 * ------------------------------------------------------------------------- */
static int cmp_col_row(const void *a, const void *b)
{
    const orc_coo *p = (const orc_coo *)a, *q = (const orc_coo *)b;
    if (p->col != q->col)
        return p->col < q->col ? -1 : 1;
    if (p->row != q->row)
        return p->row < q->row ? -1 : 1;
    return 0;
}

typedef struct {
    int diag;     /* rank of the entry inside its column = jagged diagonal number */
    int row_orig; /* original row */
    int pcol;     /* original column, later the permuted position */
    double val;
} tj_entry;

typedef struct {
    int origin;
    int len_m1; /* the reference stores (length - 1), main-cli.c:851 */
} tj_col;

static int cmp_col_len(const void *a, const void *b)
{
    const tj_col *p = (const tj_col *)a, *q = (const tj_col *)b;
    if (p->len_m1 != q->len_m1)
        return p->len_m1 > q->len_m1 ? -1 : 1; /* longest first */
    if (p->origin != q->origin)
        return p->origin < q->origin ? -1 : 1;
    return 0;
}

static int cmp_diag_pcol(const void *a, const void *b)
{
    const tj_entry *p = (const tj_entry *)a, *q = (const tj_entry *)b;
    if (p->diag != q->diag)
        return p->diag < q->diag ? -1 : 1;
    if (p->pcol != q->pcol)
        return p->pcol < q->pcol ? -1 : 1;
    return 0;
}

int orc_tjds_build(const orc_coo *coo_in, int rows, int cols, int nnz,
                   int *perm, int *start_pos, int *row_ind, double *val,
                   int *num_diag, int *ref_num_tjdiag, int *last_diag_single)
{
    (void)rows;
    orc_coo *coo = (orc_coo *)malloc(sizeof *coo * (size_t)(nnz ? nnz : 1));
    tj_entry *e = (tj_entry *)malloc(sizeof *e * (size_t)(nnz ? nnz : 1));
    tj_col *tab = (tj_col *)malloc(sizeof *tab * (size_t)(cols ? cols : 1));
    int *where = (int *)malloc(sizeof *where * (size_t)(cols ? cols : 1));
    memcpy(coo, coo_in, sizeof *coo * (size_t)nnz);
    qsort(coo, (size_t)nnz, sizeof *coo, cmp_col_row); /* :766 */

    /* :789-826 -- push every column's entries up against row 0 */
    for (int i = 0; i < nnz; ++i) {
        e[i].row_orig = coo[i].row;
        e[i].pcol = coo[i].col;
        e[i].val = coo[i].val;
        if (i == 0 || coo[i].col > coo[i - 1].col)
            e[i].diag = 0;
        else
            e[i].diag = e[i - 1].diag + 1; /* rows are distinct inside a column */
    }

    /* :845-862 -- per-column (length-1).  The reference leaves empty columns
     * unwritten; we give them length 0 (len_m1 = -1) so they sort last. */
    for (int c = 0; c < cols; ++c) {
        tab[c].origin = c;
        tab[c].len_m1 = -1;
    }
    for (int i = 0; i < nnz; ++i)
        if (i == nnz - 1 || e[i].pcol < e[i + 1].pcol)
            tab[e[i].pcol].len_m1 = e[i].diag;

    *ref_num_tjdiag = (cols > 0 ? tab[0].len_m1 : -1) + 1; /* :865, taken BEFORE the sort */

    qsort(tab, (size_t)cols, sizeof *tab, cmp_col_len); /* :868 */
    for (int k = 0; k < cols; ++k) {
        perm[k] = tab[k].origin;
        where[tab[k].origin] = k;
    }
    *num_diag = (cols > 0 && nnz > 0) ? tab[0].len_m1 + 1 : 0;

    /* :894-904 does a linear search per entry; origins are unique so a lookup
     * table gives the same renumbering. */
    for (int i = 0; i < nnz; ++i)
        e[i].pcol = where[e[i].pcol];

    qsort(e, (size_t)nnz, sizeof *e, cmp_diag_pcol); /* :926 */

    /* :944-967 -- copy out, one start position per diagonal, terminator = nnz */
    int d = 0;
    for (int i = 0; i < nnz; ++i) {
        val[i] = e[i].val;
        row_ind[i] = e[i].row_orig;
        if (i == 0 || e[i].diag > e[i - 1].diag)
            start_pos[d++] = i;
    }
    start_pos[d] = nnz;
    /* the reference only writes that terminator from the branch taken when the
     * LAST entry does not open a diagonal (:963), so a one-entry last diagonal
     * leaves it unwritten */
    *last_diag_single = (nnz >= 1 && (nnz == 1 || e[nnz - 1].diag > e[nnz - 2].diag)) ? 1 : 0;

    free(where);
    free(tab);
    free(e);
    free(coo);
    return d == *num_diag ? 0 : 1;
}

/* A correct transposed-jagged-diagonal product: entry j of diagonal d sits in
 * permuted column k = j - start_pos[d] and multiplies x[perm[k]]. */
void orc_tjds_spmv(int rows, int cols, int num_diag, const int *perm, const int *start_pos,
                   const int *row_ind, const double *val, const double *x, double *y)
{
    (void)cols;
    for (int r = 0; r < rows; ++r)
        y[r] = 0.0;
    for (int d = 0; d < num_diag; ++d) {
        const int base = start_pos[d];
        for (int j = base; j < start_pos[d + 1]; ++j)
            y[row_ind[j]] += val[j] * x[perm[j - base]];
    }
}

/* What main-cli.c:1013-1020 really computes:
 *  - the diagonal loop runs d = 0 .. ref_num_tjdiag inclusive (:1013), where
 *    ref_num_tjdiag is the length of ORIGINAL column 0 (:865);
 *  - start_pos lives in an int[rows] malloc that reads as zero where it was
 *    never written, in particular the terminator when last_diag_single;
 *  - the operand is the permuted vector indexed by the ROW (:1018).
 * With those three the four committed TJDS reports are reproduced line for line. */
void orc_tjds_spmv_refquirks(int rows, int cols, int num_diag, int ref_num_tjdiag,
                             int last_diag_single, const int *perm, const int *start_pos,
                             const int *row_ind, const double *val, const double *x, double *y)
{
    int span = (num_diag > ref_num_tjdiag ? num_diag : ref_num_tjdiag) + 3;
    int *sp = (int *)calloc((size_t)span, sizeof *sp);
    double *xp = (double *)calloc((size_t)(rows > cols ? rows : cols) + 1, sizeof *xp);
    for (int d = 0; d <= num_diag; ++d)
        sp[d] = start_pos[d];
    if (last_diag_single)
        sp[num_diag] = 0;
    for (int k = 0; k < cols; ++k) /* :907-923, square matrices */
        if (perm[k] < rows && k < rows)
            xp[k] = x[perm[k]];
    for (int r = 0; r < rows; ++r)
        y[r] = 0.0;
    for (int d = 0; d < ref_num_tjdiag + 1; ++d)
        for (int j = sp[d]; j < sp[d + 1]; ++j) {
            int p = row_ind[j];
            y[p] += val[j] * xp[p];
        }
    free(xp);
    free(sp);
}

/* ---------------------------------------------------------------------------
 * Timing statistics (main-cli.c:428-456, :114-130) and the report file
 * (:246-320).  The reference's stdev reads two uninitialised locals; this is
 * the population standard deviation it means to compute.
 * ------------------------------------------------------------------------- */
void orc_time_stats(const double *ms, int n, orc_stats *out)
{
    double total = 0.0, lo = 0.0, hi = 0.0;
    for (int i = 0; i < n; ++i) {
        total += ms[i];
        if (i == 0 || ms[i] < lo)
            lo = ms[i];
        if (i == 0 || ms[i] > hi)
            hi = ms[i];
    }
    double mean = n > 0 ? total / n : 0.0, ss = 0.0;
    for (int i = 0; i < n; ++i)
        ss += (ms[i] - mean) * (ms[i] - mean);
    out->total = total;
    out->avg = mean;
    out->min = lo;
    out->max = hi;
    out->stdev = n > 0 ? sqrt(ss / n) : 0.0;
}

int orc_write_report(const char *full_path, const char *alg_name, unsigned long unix_time,
                     const char *input_name, int nnz, int rows, int iters,
                     const double *y, const orc_stats *st)
{
    FILE *f = fopen(full_path, "a+"); /* :293 */
    if (!f)
        return -1;
    fprintf(f, "Execution results for smvp-toolbox v.0.6.4, %s algorithm\n", alg_name);
    fprintf(f, "Generated on %lu (Unix time)\n\n", unix_time);
    fprintf(f, "Sparse matrix file in use:\n%s\n\n", input_name);
    fprintf(f, "Non-zero numbers contained in matrix: %d\n\n", nnz);
    fprintf(f, "Compute times for %d iterations:\n\n", iters);
    fprintf(f, "Total Time: %g ms\n", st->total);
    fprintf(f, "Average Time: %g ms\n", st->avg);
    fprintf(f, "Fastest Time: %g ms\n", st->min);
    fprintf(f, "Slowest Time: %g ms\n", st->max);
    fprintf(f, "Time StDev: %g ms\n\n", st->stdev);
    fprintf(f, "Output vector (one cell per line):\n[\n");
    for (int r = 0; r < rows; ++r)
        fprintf(f, r < rows - 1 ? "%g\n" : "%g\n]\n\n", y[r]);
    fclose(f);
    return 0;
}

/* ---------------------------------------------------------------------------
 * CPU baseline legs: same window as the reference -- y reset outside the
 * window, CLOCK_MONOTONIC_RAW around the product only (:405-419, :1008-1023).
 * ------------------------------------------------------------------------- */
static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC_RAW, &t);
    return (double)t.tv_sec * 1e3 + (double)t.tv_nsec / 1e6;
}

void orc_csr_timed(int rows, const int *row_ptr, const int *col_ind, const double *val,
                   const double *x, double *y, int iters, double *ms_each)
{
    for (int it = 0; it < iters; ++it) {
        for (int r = 0; r < rows; ++r)
            y[r] = 0.0;
        double t0 = now_ms();
        for (int r = 0; r < rows; ++r)
            for (int j = row_ptr[r]; j < row_ptr[r + 1]; ++j)
                y[r] += val[j] * x[col_ind[j]];
        ms_each[it] = now_ms() - t0;
    }
}

void orc_tjds_timed(int rows, int cols, int num_diag, const int *perm, const int *start_pos,
                    const int *row_ind, const double *val, const double *x, double *y,
                    int iters, double *ms_each)
{
    double *xp = (double *)malloc(sizeof *xp * (size_t)(cols ? cols : 1));
    for (int k = 0; k < cols; ++k)
        xp[k] = x[perm[k]];
    for (int it = 0; it < iters; ++it) {
        for (int r = 0; r < rows; ++r)
            y[r] = 0.0;
        double t0 = now_ms();
        for (int d = 0; d < num_diag; ++d) {
            const int base = start_pos[d];
            for (int j = base; j < start_pos[d + 1]; ++j)
                y[row_ind[j]] += val[j] * xp[j - base];
        }
        ms_each[it] = now_ms() - t0;
    }
    free(xp);
}

/* ---------------------------------------------------------------------------
 * CISR .coe generator: main-cli.c:473-729 (smvp_cisr_coegen).  PARITY UNPINNED:
 * the reference holds no .coe output and main-cli.c cannot be built here
 * (libpopt), so nothing but a reading of the source checks this restatement.
 *
 * It keeps the reference's structure: the whole slot-group table first
 * (:524-607), then the value records (:624-648), then the packed words
 * (:689-728).  Returns 0, or 1 where the reference prints "slot_group_iter
 * overran fInputNonZeros!" and exits (:603-607) -- which it does for every
 * matrix when there is a single slot.  row_ptr is the standard prefix sum; the
 * reference's own is undefined for matrices with empty rows (:497-512).
 * ------------------------------------------------------------------------- */
int orc_cisr_coegen(const orc_coo *coo_in, int rows, int nnz, int slots, FILE *out)
{
    int *row_ptr = (int *)malloc(sizeof(int) * ((size_t)rows + 1));
    int *col_ind = (int *)malloc(sizeof(int) * (size_t)(nnz ? nnz : 1));
    double *val = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
    int *row_len = (int *)calloc((size_t)rows + 1, sizeof(int));
    orc_csr_build(coo_in, rows, nnz, row_ptr, col_ind, val);

    /* worst case one group per entry (+1): the reference allocates nnz groups and bails out beyond */
    int *grp = (int *)malloc(sizeof(int) * (size_t)slots * ((size_t)nnz + 2));
    int *row_end = (int *)calloc((size_t)slots, sizeof(int));
    int g = 0, next_row = 0, eof = 0, rc = 0;
    while (!eof) {
        for (int s = 0; s < slots; ++s) {
            int *cur = &grp[(size_t)g * slots + s];
            if (g == 0 || grp[(size_t)(g - 1) * slots + s] >= row_end[s] - 1) { /* :534 first group / :559 row used up */
                if (next_row < rows) {                                           /* :539, :569 */
                    *cur = row_ptr[next_row];
                    row_end[s] = row_ptr[next_row + 1];
                    row_len[next_row] = row_ptr[next_row + 1] - row_ptr[next_row];
                    ++next_row;
                } else {
                    *cur = row_ptr[rows] + 1; /* :549, :565 "invalid index" */
                }
            } else {
                *cur = grp[(size_t)(g - 1) * slots + s] + 1; /* :580 */
            }
        }
        eof = 1; /* :586-591 */
        for (int s = 0; s < slots; ++s)
            if (grp[(size_t)g * slots + s] < nnz)
                eof = 0;
        ++g;
        if (g >= nnz) { /* :596-600 */
            rc = 1;
            break;
        }
    }
    if (rc == 0) {
        const int words = g * slots;
        fprintf(out, "\n;*********************************************");
        fprintf(out, "\n;* CISR COE File for Vivado Single-Port BRAM *");
        fprintf(out, "\n;*********************************************\n");
        fprintf(out, "\n;Generated with a slot/channel count of: %d\n\n", slots);
        fprintf(out, "memory_initialization_radix=16;\n");
        fprintf(out, "memory_initialization_vector=\n");
        fprintf(out, "00%08x,\n", 0xAAAAAAAAu);
        int rl = 0;
        for (int w = 0; w < words; ++w) {
            const int idx = grp[w], slot = w % slots;
            double v = 0.0;
            int c = 0;
            if (idx < nnz) { /* :631-645: padding is value 0, column 0 */
                v = val[idx];
                c = col_ind[idx];
            }
            /* :703 ((int)val << 20) | (col << 8) | slot, printed with %08x: the low 32 bits */
            const int iv = (v >= -2147483648.0 && v < 2147483648.0) ? (int)v : (int)0x80000000u;
            unsigned word = ((unsigned)iv << 20) | ((unsigned)c << 8) | (unsigned)slot;
            fprintf(out, "01%08x,\n", word);
            if (rl < rows) { /* :707-726 two row lengths per word while any remain */
                word = (1u << 28) | ((unsigned)row_len[rl] << 16);
                ++rl;
                if (rl < rows) {
                    word |= (1u << 12) | (unsigned)row_len[rl];
                    ++rl;
                }
                fprintf(out, "02%08x,\n", word);
            }
        }
        fprintf(out, "03%08x;\n\n", 0xFFFFFFFFu);
    }
    free(grp);
    free(row_end);
    free(row_len);
    free(row_ptr);
    free(col_ind);
    free(val);
    return rc;
}

int orc_cisr_coegen_path(const orc_coo *coo, int rows, int nnz, int slots, const char *path)
{
    FILE *f = fopen(path, "w");
    if (!f)
        return -1;
    const int rc = orc_cisr_coegen(coo, rows, nnz, slots, f);
    fclose(f);
    return rc;
}
