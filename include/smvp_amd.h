/*
 * smvp_amd.h -- C ABI of the MI355X-native SpMV engine (libsmvp_amd.so).
 *
 * This is the drop-in boundary for smvp-toolkit's CSR / TJDS path.  The
 * reference has no plugin or FFI layer: its boundary is the pair of C functions
 * main() calls (main-cli.c:325 smvp_csr_compute, main-cli.c:734
 * smvp_tjds_compute) plus the Matrix Market reader and the report writer around
 * them.  Each entry point below names the reference interface it replaces.
 * Paths are relative to the reference checkout (smvp-toolkit v0.6.4).
 *
 * Conventions
 *   - plain C types only; every function returns an int status (0 = SMVP_OK);
 *     the reference's functions cannot fail and its CLI exits on error, so the
 *     CLI in smvp-toolkit_amd/cli maps these codes onto the same messages.
 *   - indices are 32-bit `int`, values `double`, exactly like the reference
 *     (main-cli.c:42-47, 61-75).
 *   - inputs are never modified (the reference sorts the caller's COO array in
 *     place, main-cli.c:340,766) and outputs go to caller-owned buffers (the
 *     reference returns a malloc'd y it never frees, main-cli.c:370,468).
 *   - "d_" pointers are device (HBM) addresses, `stream` is a hipStream_t passed
 *     as void* (NULL = the null stream).  A handle belongs to one device; calls
 *     on different handles are independent, calls on one handle are not
 *     re-entrant.
 *   - there is no CPU fallback: compute entry points return
 *     SMVP_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef SMVP_AMD_H
#define SMVP_AMD_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMVP_VERSION_MAJOR 0
#define SMVP_VERSION_MINOR 6
#define SMVP_VERSION_REVISION 4 /* report header keeps the reference's version string, main-cli.c:7-9,294 */

/* ------------------------------------------------------------------ status */
enum {
    SMVP_OK = 0,
    SMVP_ERR_INVALID = 1,     /* bad argument / inconsistent arrays */
    SMVP_ERR_NO_DEVICE = 2,   /* no usable HIP device: the product has no CPU path */
    SMVP_ERR_HIP = 3,         /* a HIP runtime call failed (see smvp_last_error) */
    SMVP_ERR_ALLOC = 4,
    SMVP_ERR_IO = 5,
    SMVP_ERR_UNSUPPORTED = 6,
    /* Matrix Market codes keep mmio's numbers (mmio/mmio.h:75-81) */
    SMVP_MM_COULD_NOT_READ_FILE = 11,
    SMVP_MM_PREMATURE_EOF = 12,
    SMVP_MM_NOT_MTX = 13,
    SMVP_MM_NO_HEADER = 14,
    SMVP_MM_UNSUPPORTED_TYPE = 15,
    SMVP_MM_LINE_TOO_LONG = 16,
    SMVP_MM_COULD_NOT_WRITE_FILE = 17
};
const char *smvp_last_error(void);     /* thread-local text of the last failure */
const char *smvp_version_string(void); /* "0.6.4" */
/* Plan options for experiments and tests (process-wide; read where a plan or handle is built or a file is parsed, never on a launch
 * path; value < 0 = back to the library's default).  What earlier rounds read from the environment -- the library reads no
 * environment variable any more:
 *   "csr_col16" 0: the CSR tile kernel keeps 32-bit columns           "csr_rowrel" 0: its second phase reads row_ptr
 *   "binned_near" 1: the binned plan's near part on the tile kernel   "binned_overlap" 0: pass A behind the near part, one stream
 *   "tjds_index" 0 | 1 | 2: 16-bit position words (default) / 32-bit sorted / 32-bit columns
 *   "sharded_threads" 1: the sharded layer's issuing threads with one GPU too     "mm_threads" n: the reader's threads (1 = serial)
 * Unknown names are SMVP_ERR_INVALID. */
int smvp_set_option(const char *name, int value);
int smvp_get_option(const char *name, int *value); /* -1 = not set */

/* -------------------------------------------------------------------- types */
/* One stored entry, 0-based.  Replaces MMRawData, main-cli.c:42-47. */
typedef struct smvp_coo {
    int row;
    int col;
    double val;
} smvp_coo_t;

/* Replaces MM_typecode (mmio/mmio.h:17): [0]='M', [1]='C'oordinate|'A'rray,
 * [2]='R'eal|'C'omplex|'P'attern|'I'nteger, [3]='G'eneral|'S'ymmetric|'H'ermitian|s'K'ew. */
typedef char smvp_mm_typecode[4];

/* Replaces struct _time_data_ (main-cli.c:87-95) without the flexible array:
 * the per-iteration times go to a caller-owned double[iters]. */
typedef struct smvp_time_stats {
    double time_total, time_avg, time_stdev, time_min, time_max; /* milliseconds */
} smvp_time_stats_t;

/* ------------------------------------------------------ Matrix Market input */
/* Replace mm_read_banner (mmio/mmio.c:96) and mm_read_mtx_crd_size
 * (mmio/mmio.c:180): same accepted inputs, same return codes. */
int smvp_mm_read_banner(FILE *f, smvp_mm_typecode *matcode);
int smvp_mm_read_mtx_crd_size(FILE *f, int *rows, int *cols, int *nnz);
/* Replaces the entry loop in main(), main-cli.c:1426-1441: reads `nnz` entries
 * into caller-owned storage, 1-based -> 0-based, pattern files get val = 1,
 * symmetric storage is NOT mirrored.  A short file gives SMVP_MM_PREMATURE_EOF
 * (the reference would carry on with uninitialised entries). */
int smvp_mm_read_coo_entries(FILE *f, const smvp_mm_typecode matcode, int nnz, smvp_coo_t *out);
/* Convenience for non-C callers: open + banner + size (+ entries). */
int smvp_mm_read_header_path(const char *path, smvp_mm_typecode *matcode, int *rows, int *cols, int *nnz);
int smvp_mm_read_coo_path(const char *path, smvp_coo_t *out, int capacity,
                          smvp_mm_typecode *matcode, int *rows, int *cols, int *nnz);

/* Optional, NOT what the reference does (main-cli.c:1427-1441 multiplies the stored triangle of a symmetric file as it
 * stands, and so does everything here by default): mirror the off-diagonal entries of symmetric / hermitian (real) /
 * skew-symmetric storage so that the full matrix is multiplied.  General files are copied.  `out` may not alias `coo`. */
int smvp_mm_expanded_count(const smvp_mm_typecode matcode, const smvp_coo_t *coo, int nnz, int *count);
int smvp_mm_expand_symmetric(const smvp_mm_typecode matcode, const smvp_coo_t *coo, int nnz, int rows, int cols,
                             smvp_coo_t *out, int capacity, int *nnz_out);

/* Binary cache of a loaded matrix (new: the reference parses the text on every run): CSR arrays behind a header that
 * names the .mtx they were made from -- its size and the FNV-1a 64 of its bytes -- plus a checksum of the arrays.
 * flags bit 0 = symmetric storage was expanded.  smvp_cache_read_header fails with SMVP_ERR_IO when there is no
 * cache and with SMVP_ERR_INVALID when the file is not one, or was made from other bytes than mtx_path now holds. */
int smvp_cache_write_csr(const char *cache_path, const char *mtx_path, const smvp_mm_typecode matcode, int flags,
                         int rows, int cols, int nnz, const int *row_ptr, const int *col_ind, const double *val);
int smvp_cache_read_header(const char *cache_path, const char *mtx_path, smvp_mm_typecode *matcode, int *flags,
                           int *rows, int *cols, int *nnz);
int smvp_cache_read_csr(const char *cache_path, int rows, int nnz, int *row_ptr, int *col_ind, double *val);
int smvp_coo_from_csr(int rows, const int *row_ptr, const int *col_ind, const double *val, smvp_coo_t *out);

/* ------------------------------------------------------- format conversion */
/* Replaces the CSR build inside smvp_csr_compute, main-cli.c:340-365.
 * row_ptr[rows+1], col_ind[nnz], val[nnz]; entries ordered by (row, col) with
 * ties kept in input order.  Bit-identical to the reference's arrays whenever
 * the reference's are defined (no empty rows); empty rows get the standard
 * prefix-sum row_ptr. */
int smvp_csr_from_coo(const smvp_coo_t *coo, int rows, int nnz,
                      int *row_ptr, int *col_ind, double *val);
/* Replaces the TJDS build inside smvp_tjds_compute, main-cli.c:766-967.
 * perm[cols]: original column at permuted position k (columns by length
 * descending, ties by original index ascending, main-cli.c:209-223,868);
 * start_pos[*num_diag + 1] with the terminator start_pos[D] = nnz always
 * written (the reference omits it when the last diagonal has one entry,
 * main-cli.c:951-966); row_ind[nnz], val[nnz] in (diagonal, permuted column)
 * order.  start_pos_capacity must be >= D + 1 (rows + 1 always suffices when no
 * (row, col) pair repeats).  Optional outputs may be NULL:
 *   ref_num_tjdiag    the diagonal count the reference derives (main-cli.c:865:
 *                     length of ORIGINAL column 0) -- used by ref-quirks mode;
 *   last_diag_single  1 when the reference would leave the terminator unwritten. */
int smvp_tjds_from_coo(const smvp_coo_t *coo, int rows, int cols, int nnz,
                       int *perm, int *start_pos, int start_pos_capacity,
                       int *row_ind, double *val,
                       int *num_diag, int *ref_num_tjdiag, int *last_diag_single);

/* The same two conversions on the GPU (next row of SURVEY 8(f)): every pointer is a device
 * address, outputs are bit-identical to the host versions above.  Radix sort + scans instead of
 * the reference's qsorts and its O(nnz * cols) renumbering (main-cli.c:894-904).  Both return
 * after the work on `stream` has finished. */
int smvp_csr_from_coo_device(const smvp_coo_t *d_coo, int rows, int cols, int nnz,
                             int *d_row_ptr, int *d_col_ind, double *d_val, void *stream);
int smvp_tjds_from_coo_device(const smvp_coo_t *d_coo, int rows, int cols, int nnz,
                              int *d_perm, int *d_start_pos, int start_pos_capacity,
                              int *d_row_ind, double *d_val,
                              int *num_diag, int *ref_num_tjdiag, int *last_diag_single, void *stream);

/* ---------------------------------------------------------- device / engine */
int smvp_device_count(int *count);
int smvp_device_info(int device, char *name, size_t name_cap, int *compute_units,
                     size_t *hbm_bytes);

/* CSR kernel families (smvp_csr_set_kernel).  AUTO picks STREAM; STREAM_CARRY when some row is longer
 * than 16384 entries; COLSWEEP for a large matrix whose gathers scatter over an operand much larger than
 * the L2 (measured at create time on samples of col_ind: spread >= 0.6, rows long enough for the sweep's
 * window); BINNED for a large matrix of which a good share (>= 10 %) of the entries lies far from the
 * diagonal and scatters (spread >= 0.2) -- SURVEY 8(d)'s memplus-shaped random model.  Every family gives
 * the same result from run to run, bit for bit.
 * COLSWEEP, BINNED, the 16-bit column offsets of STREAM and the TJDS value cache keep (parts of) the entries a
 * second time, copied when the plan is built: a handle over adopted device arrays (SMVP_MEM_DEVICE) whose
 * val / col_ind are then changed in place must be re-planned (smvp_csr_set_kernel) or re-created. */
enum {
    SMVP_CSR_KERNEL_AUTO = 0,
    SMVP_CSR_KERNEL_VECTOR = 1,      /* one (sub-)wavefront per row, __shfl_down sums */
    SMVP_CSR_KERNEL_STREAM = 2,      /* fixed-nnz tiles, LDS-staged segmented reduction; a row is finished by
                                        the tile it starts in (one launch).  For the tiles whose columns span less than 65536 (if
                                        most are such) the plan keeps col_ind a second time as 16-bit offsets from the tile's
                                        smallest column and the product reads those (same sums, fewer bytes); it also keeps every
                                        row's first entry as a 16-bit offset from the first entry of the tile the row starts in,
                                        which the per-row sums read instead of row_ptr */
    SMVP_CSR_KERNEL_STREAM_CARRY = 3, /* same tiles; a row that crosses tiles is combined from per-tile carries
                                         by a second small launch (for matrices with extremely long rows) */
    SMVP_CSR_KERNEL_COLSWEEP = 4,     /* for columns scattered over an operand far larger than L2: strips of rows whose
                                         entries (kept a second time, sorted by column) are streamed so that all
                                         resident wavefronts gather from one L2-sized window of x; every row is
                                         summed in ascending order of the strip's stream, i.e. of the columns: for
                                         rows stored with ascending columns (what smvp_csr_from_coo and the reference
                                         build) bit-identical to main-cli.c:410-416; for a row whose col_ind is not
                                         ascending the sum is reproducible and within the rounding bound, but its
                                         order is not the serial loop's.  param: rows per workgroup of four strips
                                         (256 ... 20480, a multiple of 4; 0 = chosen from the block's rows: the height with the best share of busy CUs x rate,
                                         which may be the one that cuts the rows into whole generations of 256 workgroups).
                                         SMVP_CSR_SWEEP_PARTS(rb, parts), parts = 2 or 4 (rb * parts <= 20480): the workgroup's
                                         wavefronts share two strips / one strip, each taking a half / quarter of the columns
                                         into partial sums of its own -- strips 2x / 4x as tall for the same rows per workgroup,
                                         which is what a block of few rows (one rank's share of a sharded matrix) lacks; a row's
                                         sum is then its partial sums added part by part: the same from run to run, inside the
                                         rounding bound, NOT bit-identical to the serial loop.  parts = 8: one column part per XCD
                                         (workgroup b of a launch runs on XCD b % 8 and sweeps eighth b % 8 of the columns for
                                         the four whole strips of row group b / 8; the eight partial sums of a row meet in 8 *
                                         rows doubles of scratch and are added by a second kernel).  Never picked by AUTO */
    SMVP_CSR_KERNEL_BINNED = 5        /* for matrices with a band around the diagonal plus many entries far from it
                                         (anywhere in an operand much larger than the L2): the entries are kept a second
                                         time, split by |column - row| > band.  The near part is summed out of a row
                                         block's window of x in LDS (rows of a block of 8192 sorted by length, every
                                         lane its own row left to right; a row of more than 16 near entries by a
                                         wavefront) -- or runs on STREAM where that does not suit (band > 4096, or a
                                         block with more than 1024 such long rows).  The far part never gathers from
                                         memory either: pass A -- one workgroup per block of 16384
                                         columns, that block of x in LDS -- stores every far product into bins ordered
                                         (row block, column block); pass B -- one workgroup per row block -- sums each
                                         row's far products in ascending column order and adds them to the near sum.
                                         No atomics.  Pass A runs beside the near part on a stream the handle owns
                                         (ordered against the caller's stream by events); a handle is therefore used by
                                         one stream at a time, like every plan that keeps buffers.  param: the band
                                         (0 = 4096) */
};
enum {
    SMVP_MEM_HOST = 0,  /* arrays are host memory: copied to the device */
    SMVP_MEM_DEVICE = 1 /* arrays are device memory: adopted, must outlive the handle */
};

typedef struct smvp_csr smvp_csr_t;   /* device-resident CSR matrix + launch plan */
typedef struct smvp_tjds smvp_tjds_t; /* device-resident TJDS matrix + launch plan */

/* Device-side half of smvp_csr_compute (main-cli.c:343-370): the three arrays
 * live in HBM, laid out exactly as CSRData (main-cli.c:61-66).  row_ptr is
 * always read from the host copy as well to build the launch plan, so with
 * SMVP_MEM_DEVICE pass the host row_ptr in `host_row_ptr` (NULL = copy it back). */
int smvp_csr_create(smvp_csr_t **out, int device, int rows, int cols, int nnz,
                    const int *row_ptr, const int *col_ind, const double *val,
                    int mem_kind, const int *host_row_ptr);
/* The same handle for a ROW BLOCK [first_row, first_row + rows) of a larger matrix -- what one rank of a sharded product
 * holds (smvp_sharded.hip creates its chunks with it; new design, the reference is one thread): row_ptr is the block's own
 * (0-based), col_ind stays global.  Plans that go by the distance from the diagonal -- BINNED's near / far split and its
 * LDS windows of x, AUTO's far share -- then take the diagonal where it lies in the whole matrix: column first_row + r for
 * local row r.  (Without it every block beyond the first few thousand rows looks all far.)  The results are the same. */
int smvp_csr_create_block(smvp_csr_t **out, int device, int rows, int cols, int nnz,
                          const int *row_ptr, const int *col_ind, const double *val,
                          int mem_kind, const int *host_row_ptr, long long first_row);
int smvp_csr_set_kernel(smvp_csr_t *h, int kernel, int param); /* param: lanes per row (VECTOR) / nnz per tile (STREAM), 0 = default */
int smvp_csr_get_kernel(const smvp_csr_t *h, int *kernel, int *param);
/* What AUTO's choice of COLSWEEP rests on: the share (0 ... 1) of the matrix's gathers that pull their own
 * 128-byte line of x through the L2, estimated on 64 samples of 65536 consecutive entries of col_ind (measured
 * on first use, then kept); -1 for matrices of fewer than 4 M entries, which are not sampled. */
int smvp_csr_gather_spread(smvp_csr_t *h, double *spread);
/* What AUTO's choice of BINNED rests on besides the spread: the share (0 ... 1) of the entries further than 4096 from the
 * diagonal (of the whole matrix: smvp_csr_create_block); measured on first use, then kept; -1 where it could not be. */
int smvp_csr_far_share(smvp_csr_t *h, double *share);
/* The timed product, main-cli.c:410-416: d_y[0..rows) = A * d_x[0..cols).  Asynchronous
 * on `stream`; d_y is fully overwritten (no pre-zeroing needed). */
int smvp_csr_spmv(smvp_csr_t *h, const double *d_x, double *d_y, void *stream);
/* Name of the dominant kernel symbol of the current plan and its algorithmic
 * byte count per launch: 12*nnz + 4*(rows+1) + 8*cols + 8*rows. */
int smvp_csr_describe(const smvp_csr_t *h, char *kernel_name, size_t cap, double *alg_bytes);
/* Kernel launches per product of the current plan: 1, except STREAM_CARRY (2: tiles + carry fix-up), COLSWEEP
 * (its workgroups start in generations that are resident together; config 4 on one GPU: 5) and BINNED (3: near part,
 * far products, far sums; one more where rows that keep their far entries near are summed apart). */
int smvp_csr_plan_launches(const smvp_csr_t *h, int *launches);
/* What the current launch plan costs (the reference has no counterpart: its set-up is the three qsorts and the
 * O(nnz * N) renumbering of main-cli.c:340,766-926, untimed): bytes of HBM the format's own arrays take, bytes the plan
 * keeps beside them (second copies included), and the host wall time the last plan build took, device work included. */
typedef struct smvp_plan_info {
    double matrix_bytes; /* CSR: 12 nnz + 4 (rows + 1); TJDS: 12 nnz + 4 (D + 1) + 4 cols */
    double plan_bytes;
    double build_ms;
} smvp_plan_info_t;
int smvp_csr_plan_info(const smvp_csr_t *h, smvp_plan_info_t *out);
void smvp_csr_destroy(smvp_csr_t *h);

/* Device-side half of smvp_tjds_compute (main-cli.c:756-763,944-967): val,
 * row_ind, start_pos as TJDSData (main-cli.c:70-75) plus perm. */
int smvp_tjds_create(smvp_tjds_t **out, int device, int rows, int cols, int nnz, int num_diag,
                     const int *perm, const int *start_pos, const int *row_ind,
                     const double *val, int mem_kind);
/* Replaces the operand permutation main-cli.c:907-923: x_perm[k] = x[perm[k]],
 * kept inside the handle.  Call again whenever x changes. */
int smvp_tjds_set_x(smvp_tjds_t *h, const double *d_x, void *stream);
/* The timed product, main-cli.c:1013-1020, in its corrected form
 * y[row_ind[j]] += val[j] * x_perm[j - start_pos[d]].  In the default mode (ROW_GATHER) and in TWO_PHASE d_y is
 * overwritten; in ATOMIC mode (and ref-quirks mode) d_y must be zero on entry -- the reference
 * zeroes it outside its timed window, main-cli.c:1008 -- and smvp_tjds_zero_y does that on the
 * same stream (it is a no-op when the mode does not need it, so it is always safe to call). */
int smvp_tjds_zero_y(smvp_tjds_t *h, double *d_y, void *stream);
int smvp_tjds_spmv(smvp_tjds_t *h, double *d_y, void *stream);
/* How the scatter is carried out.
 * ROW_GATHER (default): ONE kernel per product.  At create time the entries are regrouped by row (val / row_ind /
 *   start_pos / perm are only read); every tile of 2048 entries lists its entries in TJDS order and walks its piece of
 *   the jagged diagonals the way the format stores them, the products meet in LDS and one lane (or wave) per row sums
 *   them in ascending TJDS position -- no atomics, bit-reproducible, y needs no zeroing.
 * TWO_PHASE: products stored once per entry (column-major kernel: thread k walks down permuted column k with x_perm[k] in a
 *   register), then summed per row by the one-kernel form's tile kernel reading the products where that reads val (no value
 *   cache, no operand) -- same order of summation, same bits, two launches and 16 B per entry more traffic.
 * ATOMIC: one column-major pass, fp64 atomic adds into a zeroed y (order varies from run to run).  Ref-quirks mode
 *   always runs this form.
 * The plans of the first two are built on the device the first time the mode is selected. */
enum {
    SMVP_TJDS_MODE_AUTO = 0,               /* = ROW_GATHER */
    SMVP_TJDS_MODE_ATOMIC = 1,
    SMVP_TJDS_MODE_TWO_PHASE = 2,
    SMVP_TJDS_MODE_ROW_GATHER = 3
};
int smvp_tjds_set_mode(smvp_tjds_t *h, int mode);
int smvp_tjds_set_tile(smvp_tjds_t *h, int entries_per_tile); /* ROW_GATHER: 256, 1024 or 2048 entries per workgroup (re-plans) */
/* ROW_GATHER's value cache.  Far down the jagged diagonals only the long columns are left: neighbours in val are
 * entries of unrelated rows, and a 128-byte line of val would be pulled through the L2 once for each of them.  The plan
 * therefore keeps a second copy of the values of every val line whose 16 entries belong to `min_tiles` or more
 * different tiles (default 2: every line that is not one tile's alone -- 56 % of the values of memplus x944, 21 % of pwt
 * x459; 4 until round 4), stored tile by tile and read coalesced; all other values are read from val itself.
 * 0 = no cache (every value from val).  The sums and their order do not depend on it.  A handle over adopted device
 * arrays (SMVP_MEM_DEVICE) must be re-created, or this called again, after val has been changed in place. */
int smvp_tjds_set_value_cache(smvp_tjds_t *h, int min_tiles);
int smvp_tjds_get_value_cache(const smvp_tjds_t *h, int *min_tiles, long long *cached_entries);
/* Reference-defect emulation for parity with the committed TJDS reports
 * (diagonal count from original column 0, missing terminator, operand indexed
 * by row: main-cli.c:865,951-966,1018).  A host-side edit of the launch plan of the atomic kernel. */
int smvp_tjds_set_ref_quirks(smvp_tjds_t *h, int enable, int ref_num_tjdiag, int last_diag_single);
int smvp_tjds_describe(const smvp_tjds_t *h, char *kernel_name, size_t cap, double *alg_bytes);
int smvp_tjds_plan_info(const smvp_tjds_t *h, smvp_plan_info_t *out); /* plan: x_perm, the work items, the selected modes' plans */
void smvp_tjds_destroy(smvp_tjds_t *h);

/* ------------------------------------------- several GPUs, one host process */
/* New design (the reference is one CPU thread): the matrix is cut into `ngpus` row blocks balanced by entries
 * (smvp_partition_rows); GPU g holds block g -- cut again into `chunks` row chunks, each its own CSR / TJDS handle --
 * plus all of x and produces its slice of y; the exchange (RCCL ncclAllGather or direct peer pushes over xGMI, below) puts
 * the full y on every GPU, chunk c travelling while chunk c+1 is multiplied.  devices NULL = 0 .. ngpus-1.  librccl is dlopen'ed on first use.
 * (bench.py does the same with one process per GPU.)  No multi-GPU box is available to this project's tests: on
 * hardware the RCCL path has run with ONE GPU only; the N > 1 logic of this layer runs in the test suite through
 * SMVP_EXCHANGE_COPIES / _DIRECT with 2 ... 8 virtual ranks on one GPU, the Python layer's through gloo.
 * When a rank fails inside a product its peers' collectives may never complete: smvp_sharded_spmv reports the error and
 * marks the handle unusable (later calls fail at once, smvp_sharded_destroy aborts the communicators instead of waiting). */
typedef struct smvp_sharded smvp_sharded_t;
/* How the y blocks travel (round 5: a measured choice, SURVEY 7 "hard parts" -- a ring all-gather pushes 7 blocks through
 * one xGMI link, direct pushes use all seven):
 *   RCCL    ncclAllGather over xGMI into a padded wire buffer, one communicator rank per GPU, then one small kernel per
 *           GPU places the pieces at their rows;
 *   COPIES  every rank pushes its chunk STRAIGHT INTO EVERY RANK'S FULL VECTOR with hipMemcpyAsync (peer access; the
 *           copy engines, no compute unit involved) -- no wire buffer, no padding, no placement pass;
 *   DIRECT  the same pushes by ONE kernel per chunk whose workgroups store to the peers' vectors over xGMI (one launch
 *           instead of N copy calls, all links at once);
 *   AUTO    (default) at handle creation every form that is available -- RCCL needs distinct devices and a librccl that
 *           loads, the pushes need peer access -- moves one product's y once, timed; the fastest is kept
 *           (smvp_sharded_exchange_info reports the times, smvp_sharded_set_exchange switches).
 * COPIES and DIRECT order the ranks with events and a host-side meeting point of the issuing threads, and accept a device
 * list that names one device several times ("virtual ranks": more ranks than GPUs must be asked for with one of these two
 * by name): the whole N-GPU code path then runs on a one-GPU box.  All forms give the same bits. */
enum { SMVP_EXCHANGE_RCCL = 0, SMVP_EXCHANGE_COPIES = 1, SMVP_EXCHANGE_DIRECT = 2, SMVP_EXCHANGE_AUTO = 3 };
typedef struct smvp_shard_opts {
    unsigned struct_size; /* sizeof(smvp_shard_opts_t) of the header the caller was built with: set by
                             smvp_shard_opts_default, checked by the create calls (a caller built against another layout
                             is refused instead of being misread) */
    int chunks;   /* row chunks per GPU (the granularity of the product / all-gather overlap); 0 = 4 when ngpus > 1, else 1 */
    int balance;  /* 1 (default): blocks and chunks balanced by entries; 0: equal heights */
    int exchange; /* SMVP_EXCHANGE_* (default AUTO); with COPIES / DIRECT ngpus may exceed the visible devices (devices NULL = g % visible) */
} smvp_shard_opts_t;
void smvp_shard_opts_default(smvp_shard_opts_t *o);
int smvp_csr_sharded_create(smvp_sharded_t **out, int ngpus, const int *devices, int rows, int cols, int nnz,
                            const int *row_ptr, const int *col_ind, const double *val); /* host CSR arrays */
int smvp_csr_sharded_create_ex(smvp_sharded_t **out, int ngpus, const int *devices, int rows, int cols, int nnz,
                               const int *row_ptr, const int *col_ind, const double *val, const smvp_shard_opts_t *opts);
int smvp_tjds_sharded_create(smvp_sharded_t **out, int ngpus, const int *devices, const smvp_coo_t *coo,
                             int rows, int cols, int nnz); /* an independent TJDS per row chunk */
int smvp_tjds_sharded_create_ex(smvp_sharded_t **out, int ngpus, const int *devices, const smvp_coo_t *coo,
                                int rows, int cols, int nnz, const smvp_shard_opts_t *opts);
int smvp_sharded_set_csr_kernel(smvp_sharded_t *h, int kernel, int param); /* smvp_csr_set_kernel on every chunk */
/* The exchange of one product's y (every chunk, nothing to overlap with), timed `reps` times under every available form;
 * a handle created with AUTO then keeps the fastest.  Called by the create calls for AUTO (reps = 3). */
int smvp_sharded_probe_exchange(smvp_sharded_t *h, int reps);
/* active: the SMVP_EXCHANGE_* in use; available: bit e set = form e can be selected; ms[3]: milliseconds of the last probe by
 * form (RCCL, COPIES, DIRECT; < 0: not measured); rccl_ranks: what the communicator itself reports (ncclCommCount), 0
 * without one.  NULL = skip. */
int smvp_sharded_exchange_info(const smvp_sharded_t *h, int *active, int *available, double *ms, int *rccl_ranks);
int smvp_sharded_set_exchange(smvp_sharded_t *h, int exchange); /* one of the available forms (not AUTO) */
int smvp_sharded_set_x(smvp_sharded_t *h, const double *x_host); /* NULL = ones; replicated to every GPU */
/* One product, asynchronous: the chunk products on every GPU and, by `allgather`, the exchange of y:
 * 0 none; SMVP_GATHER_OVERLAPPED: chunk c is gathered (communication stream) while chunk c+1 is multiplied;
 * SMVP_GATHER_AFTER: all gathers after all products.  timed != 0 brackets it with an event pair per GPU. */
enum { SMVP_GATHER_NONE = 0, SMVP_GATHER_OVERLAPPED = 1, SMVP_GATHER_AFTER = 2 };
int smvp_sharded_spmv(smvp_sharded_t *h, int allgather, int timed);
int smvp_sharded_synchronize(smvp_sharded_t *h, double *ms_of_last_timed_product); /* max over the GPUs */
/* power iteration: the gathered y (optionally divided by its largest magnitude) becomes x on every GPU;
 * call between two smvp_sharded_spmv(h, 1, ..), after which the all-gather is what feeds the next product */
int smvp_sharded_feed_back(smvp_sharded_t *h, int normalize);
int smvp_sharded_get_y(smvp_sharded_t *h, int slot, int gathered, double *y_host);
int smvp_sharded_info(const smvp_sharded_t *h, int *ngpus, int *rows_per_gpu); /* rows_per_gpu = the tallest block */
/* bounds[ngpus + 1] of the row blocks and chunk_bounds[ngpus * (chunks + 1)] of their chunks, global rows (NULL = skip) */
int smvp_sharded_layout(const smvp_sharded_t *h, int *chunks, int *bounds, int *chunk_bounds);
void smvp_sharded_destroy(smvp_sharded_t *h);

#define SMVP_CSR_SWEEP_PARTS(rows_per_block, parts) ((rows_per_block) | (((parts) == 8 ? 3 : (parts) == 4 ? 2 : (parts) == 2 ? 1 : 0) << 24))
/* for experiments: the same with the 256-entry chunks a wavefront keeps in flight forced to 1, 2 or 4 (0 = the library's rule) */
#define SMVP_CSR_SWEEP_PARAM(rows_per_block, parts, chunks) \
    (SMVP_CSR_SWEEP_PARTS(rows_per_block, parts) | (((chunks) == 4 ? 3 : (chunks) == 2 ? 2 : (chunks) == 1 ? 1 : 0) << 26))

/* ------------------------------------------------ reference-shaped entry points */
typedef struct smvp_run_opts {
    unsigned struct_size; /* sizeof(smvp_run_opts_t) of the caller's header: set by smvp_run_opts_default, checked by the
                             compute calls -- a struct that was never passed through smvp_run_opts_default, or was built
                             against another layout, is refused with SMVP_ERR_INVALID instead of being misread */
    int device;         /* HIP device ordinal */
    int csr_kernel;     /* SMVP_CSR_KERNEL_* */
    int csr_param;      /* 0 = default */
    int tjds_ref_quirks;/* 1: reproduce the reference's defective TJDS output */
    int convert_on_device; /* 1: COO -> CSR / TJDS on the GPU (smvp_*_from_coo_device), 0: on the host */
    int ngpus;          /* 0 or 1: one GPU (`device`); N > 1: row blocks on GPUs 0..N-1 + RCCL all-gather of y */
    int iterate;        /* 1: power iteration, x_{k+1} = A x_k for `iters` steps -- the product the assignment asked
                           for (comment at main-cli.c:401); square matrices; y = the last iterate; each step timed */
    int normalize;      /* with iterate: divide every iterate by its largest magnitude (keeps 1000 steps finite) */
    int tjds_mode;      /* SMVP_TJDS_MODE_* for smvp_tjds_compute (AUTO = ROW_GATHER) */
    int timing;         /* SMVP_TIMING_*: how each product is timed */
    int shard_exchange; /* ngpus > 1: SMVP_EXCHANGE_* (default AUTO; COPIES / DIRECT: ngpus may exceed the visible GPUs -- virtual ranks) */
    int repeat_patience_us; /* SMVP_TIMING_DEVICE, the repeating kernel: microseconds a workgroup waits at the barrier between two
                           products before the launch gives up and the run falls back to one launch per product (0 = 50 000;
                           negative = none at all: whoever has to wait gives up -- the fallback's test) */
    const double *x;    /* host operand, NULL = all ones (main-cli.c:368-369) */
} smvp_run_opts_t;
void smvp_run_opts_default(smvp_run_opts_t *o);

/* How the per-product window of main-cli.c:408-419 is taken.  EVENTS: a hipEvent pair around each product's
 * launches.  DEVICE: the kernel times itself -- every wave notes the device's constant-rate wall clock when it
 * starts and when its last store has been acknowledged; the product's time is max(last) - min(first) -- and up to 1024
 * products run as ONE launch of a repeating form of the kernel whose workgroups stay resident and meet at a (fence-free: the
 * products are independent) barrier between two products; where that form is not available (or gives up: the grid must be
 * resident as a whole) the products are replayed one launch each from a hipGraph.  AUTO picks DEVICE for launches of up to 4096 workgroups of the tile
 * kernels (where an event pair would measure mostly itself: the reference's own sample matrices), else EVENTS. */
enum { SMVP_TIMING_AUTO = 0, SMVP_TIMING_EVENTS = 1, SMVP_TIMING_DEVICE = 2,
       SMVP_TIMING_DEVICE_GRAPH = 3 /* DEVICE, but one launch per product replayed from a hipGraph: what DEVICE falls back to */ };
typedef struct smvp_run_info {
    int timing;            /* SMVP_TIMING_EVENTS or SMVP_TIMING_DEVICE (also for DEVICE_GRAPH: repeat_launches / graph_replays tell the
                              form): what the last smvp_*_compute on this thread used */
    int graph_replays;     /* hipGraph launches it took (0 = plain launches) */
    double wall_ms;        /* host wall time of the whole timed loop, launches, timing and waits included */
    double device_clock_khz; /* DEVICE: rate of the clock the times were taken with */
    int repeat_launches;   /* DEVICE: launches of the repeating kernel it took -- up to 1024 products per launch, every product's
                              window stamped between device-side barriers (0: one launch per product) */
    int repeat_gave_up;    /* DEVICE: 1 = a launch of the repeating kernel gave up at a barrier (its grid was not resident as a whole) and
                              the run was done over, one launch per product */
} smvp_run_info_t;
int smvp_last_run_info(smvp_run_info_t *out);

/* Replaces  double *smvp_csr_compute(MMRawData*, int rows, int nnz, int iters,
 *                                    struct _time_data_*)      main-cli.c:325-469
 * COO in -> CSR build -> upload -> `iters` products, each timed on its own with
 * hipEvents around the product only (the reference's clock_gettime window,
 * main-cli.c:408-419) -> y[rows] and time_each_ms[iters] out, stats reduced as
 * main-cli.c:428-456.  `cols` is new: the reference assumes a square matrix. */
int smvp_csr_compute(const smvp_coo_t *coo, int rows, int cols, int nnz, int iters,
                     const smvp_run_opts_t *opts, double *y, double *time_each_ms,
                     smvp_time_stats_t *stats);
/* Replaces  double *smvp_tjds_compute(MMRawData*, int rows, int cols, int nnz,
 *                                     int iters, struct _time_data_*)  main-cli.c:734-1162 */
int smvp_tjds_compute(const smvp_coo_t *coo, int rows, int cols, int nnz, int iters,
                      const smvp_run_opts_t *opts, double *y, double *time_each_ms,
                      smvp_time_stats_t *stats);

/* ------------------------------------------------------------ stats + report */
/* Replaces the reduction at main-cli.c:428-456 and calcStDevDouble (:114-130;
 * population standard deviation -- the reference's reads uninitialised locals). */
int smvp_time_stats(const double *time_each_ms, int iters, smvp_time_stats_t *out);
/* Replaces generateReportText, main-cli.c:246-320: writes
 * <report_dir>/smvp-toolbox_report_<alg_name>_<unix_time>.txt (opened "a+") with
 * the reference's exact text.  report_dir NULL or "" = current directory;
 * unix_time 0 = time(NULL).  The path written is returned in out_path if given. */
int smvp_generate_report_text(const char *input_file_name, const char *report_dir,
                              const char *alg_name, int nnz, int rows, int iters,
                              const double *y, const smvp_time_stats_t *stats,
                              unsigned long unix_time, char *out_path, size_t out_path_cap);

/* ------------------------------------------------------------- CISR export */
/* Replaces  void smvp_cisr_coegen(MMRawData*, int rows, int nnz, int slotCount)   main-cli.c:473-729
 * (-g / --cisr-gen, -s / --slots): the matrix dealt row by row onto `slots` channels and written as a Xilinx Vivado
 * .coe block-RAM image -- the reference prints it to stdout, here it goes to `out`.  Host only (no GPU).  Returns
 * SMVP_ERR_UNSUPPORTED where the reference prints "slot_group_iter overran fInputNonZeros!" and exits
 * (main-cli.c:596-600; always with one slot).  Parity unpinned: the reference holds no .coe output. */
int smvp_cisr_coegen(const smvp_coo_t *coo, int rows, int nnz, int slots, FILE *out);
int smvp_cisr_coegen_path(const smvp_coo_t *coo, int rows, int nnz, int slots, const char *path);

/* ------------------------------------------------------ synthetic workloads */
/* SURVEY 8(d) / BASELINE.json configs: matrices generated straight into CSR,
 * a pure function of (kind, seed, global row) so any row block can be produced
 * independently (row-block sharding needs no communication). */
enum {
    SMVP_SYNTH_MEMPLUS_SHAPED = 1, /* memplus's row-length histogram + band structure, scaled */
    SMVP_SYNTH_UNIFORM = 2         /* `param` nnz per row, uniform distinct columns */
};
/* Row lengths for global rows [row_begin, row_end): lens[row_end-row_begin]. */
int smvp_synth_row_lengths(int kind, uint64_t seed, int64_t rows_total, int64_t cols_total,
                           int param, int64_t row_begin, int64_t row_end, int *lens);
/* Fill col_ind/val for the block given its (local, 0-based) row_ptr; columns
 * sorted ascending and distinct inside a row; values uniform in [-1, 1). */
int smvp_synth_fill(int kind, uint64_t seed, int64_t rows_total, int64_t cols_total,
                    int param, int64_t row_begin, int64_t row_end, const int *row_ptr,
                    int *col_ind, double *val, int threads);
/* Row-block partition balanced by nnz: bounds[parts+1], bounds[0]=0, bounds[parts]=rows. */
int smvp_partition_rows(const int *row_ptr, int rows, int parts, int *bounds);
/* x[i] uniform in [0, 1): the top 53 bits of splitmix64's finaliser of (seed + i), times 2^-53.  The operand of the
 * command line's --x random (seed 67890); the reference only ever multiplies by ones (main-cli.c:368-369). */
int smvp_vector_random(double *x, int64_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* SMVP_AMD_H */
