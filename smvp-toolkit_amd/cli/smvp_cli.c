/*
 * smvp_cli.c -- smvp-toolkit-cli on the MI355X engine (C host code over the
 * C ABI in include/smvp_amd.h).
 *
 * Keeps the command-line surface of the reference's main() (main-cli.c:1219-1481):
 *   -a/--all-algs  -c/--csr  -t/--tjds  -g/--cisr-gen  -n/--number INT
 *   -s/--slots INT  -d/--dir DIR  -?/--help  --usage      <file>
 * options before the single positional file (POPT_CONTEXT_POSIXMEHARDER,
 * main-cli.c:1254), the same tagged stdout lines, the same error texts and exit
 * codes, and the same report file per algorithm (main-cli.c:246-320).
 *
 * Deliberate differences, all called out in DESIGN.md:
 *   - --all-algs runs CSR then TJDS.  In v0.6.4 ALG_ALL (256) shares no bit with
 *     ALG_CSR / ALG_TJDS, so the reference runs nothing (main-cli.c:34-38,1453).
 *   - without -d the report goes to the current directory (the reference reads
 *     an uninitialised pointer, main-cli.c:1223,1458).
 *   - TJDS is the corrected product unless --ref-quirks is given.
 *   - -g / -s (CISR .coe generation, main-cli.c:473-729) print the .coe image to stdout like the
 *     reference (host-only work, no GPU; --all-algs stays CSR + TJDS).
 *   - libpopt is not used (absent from the image): getopt_long("+...") gives the
 *     same POSIX ordering rule.
 * Additive flags: --device N, --gpus N, --iterate, --normalize, --ref-quirks, --device-convert, --csr-kernel auto|vector|stream|stream-carry|colsweep|binned,
 * --tjds-mode auto|row-gather|two-phase|atomic, --timing auto|events|device|device-graph, --expand-symmetric, --cache,
 * --x ones|random, --dump-arrays.  The reference's shipped binary prints whole arrays (SMVP_CSR_DEBUG 1,
 * main-cli.c:10,374-394,458-466,1166-1191); here those dumps -- and the TJDS ones its source holds behind
 * SMVP_TJDS_DEBUG (main-cli.c:870-892,969-992,1150-1158) -- are printed only with --dump-arrays.
 */
#define _POSIX_C_SOURCE 200809L
#include "smvp_amd.h"

#include <errno.h>
#include <getopt.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#define RED "\x1b[31m"
#define GREEN "\x1b[32m"
#define YELLOW "\x1b[33m"
#define MAGENTA "\x1b[35m"
#define CYAN "\x1b[36m"
#define RESET "\x1b[0m"

enum { ALG_NONE = 0, ALG_CSR = 1 << 1, ALG_TJDS = 1 << 2, ALG_CISR = 1 << 3, ALG_ALL = 256 };
enum { OPT_DEVICE = 1000, OPT_QUIRKS, OPT_KERNEL, OPT_USAGE, OPT_DEVCONV, OPT_GPUS, OPT_ITERATE, OPT_NORMALIZE, OPT_TIMING,
       OPT_TJDS_MODE, OPT_EXPAND, OPT_CACHE, OPT_X, OPT_DUMP, OPT_VIRTUAL, OPT_EXCHANGE };

static void usage(FILE *to, const char *prog)
{
    fprintf(to,
            "Usage: %s [-acgt?] [-a|--all-algs] [-c|--csr] [-g|--cisr-gen] [-t|--tjds]\n"
            "        [-n|--number=1000] [-s|--slots=16] [-d|--dir=./] [--device=0] [--gpus=1] [--exchange=auto] [--virtual-gpus]\n"
            "        [--ref-quirks] [--iterate] [--normalize] [--device-convert] [--csr-kernel=auto|vector|stream|stream-carry|colsweep|binned]\n"
            "        [--tjds-mode=auto|row-gather|two-phase|atomic] [--timing=auto|events|device|device-graph]\n"
            "        [--expand-symmetric] [--cache] [--x=ones|random] [--dump-arrays]\n"
            "        [-?|--help] [--usage]\n"
            "        [OPTIONS] <file>\n",
            prog);
}

static void help(const char *prog)
{
    printf("Usage: %s [OPTIONS] <file>\n", prog);
    puts("  -a, --all-algs           Enable all SMVP algorithms.");
    puts("  -c, --csr                Enable CSR SMVP algorithm.");
    puts("  -g, --cisr-gen           Generate CISR COE file.");
    puts("  -t, --tjds               Enable TJDS SMVP algorithm.");
    puts("  -n, --number=1000        Number of computation iterations per-algorithm.");
    puts("  -s, --slots=16           Number of slots for CISR.");
    puts("  -d, --dir=./             Output folder for reports.");
    puts("      --device=0           HIP device ordinal.");
    puts("      --gpus=1             Shard the matrix by row blocks over this many GPUs (all-gather of y after each product;");
    puts("                           so far verified on hardware with one GPU only).");
    puts("      --exchange=auto      How the blocks of y travel: rccl (ncclAllGather), copies (peer hipMemcpyAsync), direct");
    puts("                           (one push kernel per chunk), auto (times each when the blocks are created, keeps the fastest).");
    puts("      --virtual-gpus       --gpus N with the y blocks exchanged by peer pushes instead of RCCL:");
    puts("                           N may exceed the GPUs present (the ranks share them) -- a rehearsal of the N-GPU path.");
    puts("      --iterate            Power iteration: feed each result back as the next operand (x <- A x, n times).");
    puts("      --normalize          --iterate, and divide every iterate by its largest magnitude.");
    puts("      --ref-quirks         Reproduce the reference v0.6.4 TJDS output, defects included.");
    puts("      --device-convert     Build CSR / TJDS from the loaded entries on the GPU instead of the host.");
    puts("      --csr-kernel=auto    CSR kernel family: auto, vector, stream, stream-carry, colsweep, binned.");
    puts("      --tjds-mode=auto     TJDS product: auto (= row-gather, one kernel), two-phase, atomic.");
    puts("      --timing=auto        Per-product window: events (hipEvent pair), device (the kernel times itself; up to 1024");
    puts("                           products per launch), device-graph (the same stamps, one launch per product), auto.");
    puts("      --expand-symmetric   Mirror the stored triangle of a symmetric file (the reference multiplies it as stored).");
    puts("      --cache              Keep / use <file>.smvpbin, a binary copy of the loaded matrix tied to the file's checksum.");
    puts("      --x=ones             Operand: ones (the reference's) or random (uniform [0, 1), seed 67890).");
    puts("      --dump-arrays        Print the converted arrays, the result vector and every product's time like the");
    puts("                           reference's debug build (its shipped binary always does, for CSR).");
    puts("\nHelp options:");
    puts("  -?, --help               Show this help message");
    puts("      --usage              Display brief usage message");
}

static void die(const char *msg)
{
    printf(RED "[ERROR]\t%s\n" RESET, msg);
    exit(1);
}

static int is_dir(const char *path)
{
    struct stat st;
    return stat(path, &st) == 0 && S_ISDIR(st.st_mode);
}

/* popt's POPT_ARG_INT: the whole argument must be a decimal number */
static int parse_int(const char *s, int *out)
{
    char *end = NULL;
    errno = 0;
    long v = strtol(s, &end, 0);
    if (end == s || *end != '\0')
        return 1; /* POPT_ERROR_BADNUMBER */
    if (errno == ERANGE || v > INT_MAX || v < INT_MIN)
        return 2; /* POPT_ERROR_OVERFLOW */
    *out = (int)v;
    return 0;
}

static void mmio_fail(int rc)
{
    /* texts of mmioErrorHandler, main-cli.c:144-166 */
    if (rc == SMVP_MM_PREMATURE_EOF)
        die("Could not process specified Matrix Market input file. Required parameters not present on first line of file.");
    if (rc == SMVP_MM_NO_HEADER)
        die("Could not process specified Matrix Market input file. Required header is missing or file contents may not be Matrix Market formatted.");
    if (rc == SMVP_MM_UNSUPPORTED_TYPE)
        die("Could not process specified Matrix Market input file. Matrix content description not parseable or is absent.");
    die("Could not process specified Matrix Market input file. Unhandled exception occured during file loading .");
}

static void engine_fail(const char *what, int rc)
{
    printf(RED "[ERROR]\t%s failed (status %d): %s\n" RESET, what, rc, smvp_last_error());
    exit(1);
}

static void print_rates(const char *alg, int rows, int cols, int nnz, int diags, int iters, const smvp_time_stats_t *t)
{
    /* 2*nnz flops; 12*nnz + 4*(rows+1 | diags+1) + 8*cols + 8*rows bytes */
    const double flops = 2.0 * nnz;
    const double bytes = 12.0 * nnz + 4.0 * ((diags >= 0 ? diags : rows) + 1.0) + 8.0 * cols + 8.0 * rows;
    if (t->time_avg > 0.0)
        printf(CYAN "[DATA]\t%s average per product: " RESET "%g ms, %.3f GFLOP/s, %.3f GB/s algorithmic\n", alg,
               t->time_avg, flops / t->time_avg * 1e-6, bytes / t->time_avg * 1e-6);
    /* How the window of main-cli.c:408-419 was taken.  The reference's clock brackets the product call on the host; here
     * three figures exist and all are printed: the in-kernel window (what the report file's times are when the kernel
     * times itself: launch and dispatch excluded), the host wall of the whole loop per product (launches, barriers and
     * read-back included), and -- with --timing events -- a hipEvent pair around every launch (which for launches of a
     * few microseconds measures mostly the events).  rocprofv3 reads 4.8 us per dispatch for the sample matrices'
     * kernels inside the graph replay (profiles/r03_cli_n1000.txt). */
    smvp_run_info_t info;
    if (smvp_last_run_info(&info) == SMVP_OK && info.wall_ms > 0.0) {
        printf(CYAN "[DATA]\t%s timing: " RESET "%s; whole loop %g ms of host wall time\n", alg,
               info.timing == SMVP_TIMING_DEVICE
                   ? (info.repeat_launches ? "per product on the device (wall-clock stamps in the kernel), up to 1024 products per launch of the repeating kernel"
                      : info.graph_replays ? "per product on the device (wall-clock stamps in the kernel), products replayed from a hipGraph"
                                           : "per product on the device (wall-clock stamps in the kernel)")
                   : "hipEvent pair around each product",
               info.wall_ms);
        printf(CYAN "[DATA]\t%s per product: " RESET "%.3f us %s (the times in the report file), %.3f us host wall per product over the loop\n",
               alg, t->time_avg * 1e3,
               info.timing == SMVP_TIMING_DEVICE ? "in-kernel window, launch excluded" : "between the events of a pair",
               iters > 0 ? info.wall_ms * 1e3 / iters : 0.0);
    }
}

/* ---- --dump-arrays: the reference's debug dumps, text for text ---------------------------------------------------- */
static void dump_ints(const char *head, const int *a, int n)
{
    fputs(head, stdout);
    for (int i = 0; i < n; ++i)
        printf("%d, ", a[i]);
}

static void dump_vector(const char *head, const double *v, int n)
{
    fputs(head, stdout);
    for (int i = 0; i < n; ++i)
        printf("%g, ", v[i]);
    printf("]\n\n");
}

/* main-cli.c:374-394 (SMVP_CSR_DEBUG): row_ptr, val, col_ind as smvp_csr_compute converts them */
static void dump_csr_arrays(const smvp_coo_t *coo, int rows, int nnz)
{
    int *rp = malloc(sizeof *rp * ((size_t)rows + 1)), *ci = malloc(sizeof *ci * (size_t)(nnz > 0 ? nnz : 1));
    double *v = malloc(sizeof *v * (size_t)(nnz > 0 ? nnz : 1));
    if (!rp || !ci || !v)
        die("Out of memory while staging the matrix.");
    const int rc = smvp_csr_from_coo(coo, rows, nnz, rp, ci, v);
    if (rc != SMVP_OK)
        engine_fail("Converting to CSR", rc);
    dump_ints("[DEBUG]\tCSR JIT row_ptr:\n\t[", rp, rows + 1);
    printf("]\n");
    printf("[DEBUG]\tCSR JIT val:\n\t[");
    for (int i = 0; i < nnz; ++i)
        printf("%g, ", v[i]);
    printf("]\n");
    dump_ints("[DEBUG]\tCSR JIT col_ind:\n\t[", ci, nnz);
    printf("]\n\n");
    free(rp);
    free(ci);
    free(v);
}

/* smvp_csr_debug, main-cli.c:1166-1191 (its "StDev Time" line prints the average: kept as it is) */
static void dump_csr_run(const double *y, const double *each, const smvp_time_stats_t *t, int rows, int nnz, int iters)
{
    printf("[DEBUG]\tCSR Iterations: %d\n", iters);
    printf("[DEBUG]\tCSR fInputRows: %d\n", rows);
    printf("[DEBUG]\tCSR fInputNonZeros: %d\n", nnz);
    printf("[DEBUG]\tCSR Total Time: %g\n", t->time_total);
    printf("[DEBUG]\tCSR Avg Time: %g\n", t->time_avg);
    printf("[DEBUG]\tCSR StDev Time: %g\n", t->time_avg);
    printf("[DEBUG]\tCSR Times:\n");
    printf("\t[");
    for (int i = 0; i < iters; ++i)
        printf("%g, ", each[i]);
    printf("]\n");
    dump_vector("[DEBUG]\tCSR Output Vector:\n\t[", y, rows);
}

/* main-cli.c:870-892 (the reordering table) and :969-992 (val, row_ind, start_pos, num_tjdiag).  start_pos is printed
 * with its terminator over all jagged diagonals; with --ref-quirks over the reference's own count (the length of
 * original column 0, main-cli.c:865) as its loop at :985 does. */
static void dump_tjds_arrays(const smvp_coo_t *coo, int rows, int cols, int nnz, int quirks)
{
    const size_t cap = (size_t)(rows > nnz ? rows : nnz) + 2;
    int *perm = malloc(sizeof *perm * (size_t)(cols > 0 ? cols : 1)), *sp = malloc(sizeof *sp * cap);
    int *ri = malloc(sizeof *ri * (size_t)(nnz > 0 ? nnz : 1)), *len = calloc((size_t)(cols > 0 ? cols : 1), sizeof *len);
    double *v = malloc(sizeof *v * (size_t)(nnz > 0 ? nnz : 1));
    if (!perm || !sp || !ri || !len || !v)
        die("Out of memory while staging the matrix.");
    int nd = 0, ref_nd = 0, last_single = 0;
    const int rc = smvp_tjds_from_coo(coo, rows, cols, nnz, perm, sp, (int)cap, ri, v, &nd, &ref_nd, &last_single);
    if (rc != SMVP_OK)
        engine_fail("Converting to TJDS", rc);
    for (int i = 0; i < nnz; ++i)
        ++len[coo[i].col];
    printf("[DEBUG]\tTJDS PHASE 3: Reordering Table:\n");
    printf("key\t[");
    for (int k = 0; k < cols; ++k)
        printf("%d, ", k);
    printf("]\n");
    dump_ints("origCol\t[", perm, cols);
    printf("]\n");
    printf("colLen\t[");
    for (int k = 0; k < cols; ++k)
        printf("%d, ", len[perm[k]]);
    printf("]\n\n");
    printf("[DEBUG]\tTJDS PHASE 7: Pre-Calc Fields:\n");
    printf("\tval:\t\t[");
    for (int i = 0; i < nnz; ++i)
        printf("%g, ", v[i]);
    printf("]\n");
    dump_ints("\trow_ind:\t[", ri, nnz);
    printf("]\n");
    const int shown = quirks ? (ref_nd < nd ? ref_nd : nd) : nd;
    dump_ints("\tstart_pos:\t[", sp, shown + 1);
    printf("]\n\n");
    printf("\tnum_tjdiag (count, not 0-index):\t%d", shown);
    printf("\n\n");
    free(perm);
    free(sp);
    free(ri);
    free(len);
    free(v);
}

int main(int argc, char *argv[])
{
    static const struct option longopts[] = {
        {"all-algs", no_argument, NULL, 'a'},      {"csr", no_argument, NULL, 'c'},
        {"cisr-gen", no_argument, NULL, 'g'},      {"tjds", no_argument, NULL, 't'},
        {"number", required_argument, NULL, 'n'},  {"slots", required_argument, NULL, 's'},
        {"dir", required_argument, NULL, 'd'},     {"help", no_argument, NULL, '?'},
        {"usage", no_argument, NULL, OPT_USAGE},   {"device", required_argument, NULL, OPT_DEVICE},
        {"ref-quirks", no_argument, NULL, OPT_QUIRKS}, {"csr-kernel", required_argument, NULL, OPT_KERNEL},
        {"device-convert", no_argument, NULL, OPT_DEVCONV}, {"gpus", required_argument, NULL, OPT_GPUS},
        {"iterate", no_argument, NULL, OPT_ITERATE}, {"normalize", no_argument, NULL, OPT_NORMALIZE},
        {"timing", required_argument, NULL, OPT_TIMING}, {"tjds-mode", required_argument, NULL, OPT_TJDS_MODE},
        {"expand-symmetric", no_argument, NULL, OPT_EXPAND}, {"cache", no_argument, NULL, OPT_CACHE},
        {"x", required_argument, NULL, OPT_X}, {"dump-arrays", no_argument, NULL, OPT_DUMP},
        {"virtual-gpus", no_argument, NULL, OPT_VIRTUAL}, {"exchange", required_argument, NULL, OPT_EXCHANGE},
        {NULL, 0, NULL, 0}};
    const char *prog = "smvp-toolkit-cli";
    int alg_mode = ALG_NONE, calc_iter = 1000, cisr_slots = 16, device = 0, quirks = 0;
    int csr_kernel = SMVP_CSR_KERNEL_AUTO, device_convert = 0, ngpus = 1, iterate = 0, normalize = 0;
    int timing = SMVP_TIMING_AUTO, tjds_mode = SMVP_TJDS_MODE_AUTO, expand = 0, use_cache = 0, x_random = 0, dump = 0;
    int virtual_gpus = 0, exchange = SMVP_EXCHANGE_AUTO;
    const char *report_dir = "";

    if (argc < 2) { /* main-cli.c:1267-1271 */
        usage(stderr, prog);
        return 1;
    }

    opterr = 0;
    int c, v;
    while ((c = getopt_long(argc, argv, "+:acgtn:s:d:?", longopts, NULL)) != -1) {
        switch (c) {
        case 'a':
            if (alg_mode != ALG_NONE)
                die("Combining [-a|--all] with other algorithm flags is not supported.");
            alg_mode = ALG_ALL;
            break;
        case 'c':
        case 't':
        case 'g':
            if (alg_mode == ALG_ALL)
                die("Combining [-a|--all] with other algorithm flags is not supported.");
            alg_mode |= (c == 'c') ? ALG_CSR : (c == 't') ? ALG_TJDS : ALG_CISR;
            break;
        case 'n':
        case 's':
            switch (parse_int(optarg, &v)) {
            case 1:
                die("Argument for iteration count contains non-number characters.");
                break;
            case 2:
                die("Argument for iteration count must be between 0 and approximately 1.8E19 (64-bit integer).");
                break;
            default:
                break;
            }
            if (v < 1)
                die(c == 'n' ? "Invalid number of algorithm iterations specified."
                             : "Invalid number of CISR slots specified.");
            if (c == 'n')
                calc_iter = v;
            else
                cisr_slots = v;
            break;
        case 'd':
            if (!is_dir(optarg))
                die("Report output folder not found. Check path and/or create folder if it does not exist.");
            report_dir = optarg;
            break;
        case OPT_DEVICE:
            if (parse_int(optarg, &v) != 0 || v < 0)
                die("Invalid device ordinal specified.");
            device = v;
            break;
        case OPT_QUIRKS:
            quirks = 1;
            break;
        case OPT_DEVCONV:
            device_convert = 1;
            break;
        case OPT_ITERATE:
            iterate = 1;
            break;
        case OPT_NORMALIZE:
            iterate = normalize = 1;
            break;
        case OPT_GPUS:
            if (parse_int(optarg, &v) != 0 || v < 1)
                die("Invalid number of GPUs specified.");
            ngpus = v;
            break;
        case OPT_VIRTUAL:
            virtual_gpus = 1;
            break;
        case OPT_EXCHANGE:
            if (strcmp(optarg, "auto") == 0)
                exchange = SMVP_EXCHANGE_AUTO;
            else if (strcmp(optarg, "rccl") == 0)
                exchange = SMVP_EXCHANGE_RCCL;
            else if (strcmp(optarg, "copies") == 0)
                exchange = SMVP_EXCHANGE_COPIES;
            else if (strcmp(optarg, "direct") == 0)
                exchange = SMVP_EXCHANGE_DIRECT;
            else
                die("Invalid exchange specified (auto, rccl, copies or direct).");
            break;
        case OPT_KERNEL:
            if (strcmp(optarg, "auto") == 0)
                csr_kernel = SMVP_CSR_KERNEL_AUTO;
            else if (strcmp(optarg, "vector") == 0)
                csr_kernel = SMVP_CSR_KERNEL_VECTOR;
            else if (strcmp(optarg, "stream") == 0)
                csr_kernel = SMVP_CSR_KERNEL_STREAM;
            else if (strcmp(optarg, "stream-carry") == 0)
                csr_kernel = SMVP_CSR_KERNEL_STREAM_CARRY;
            else if (strcmp(optarg, "colsweep") == 0)
                csr_kernel = SMVP_CSR_KERNEL_COLSWEEP;
            else if (strcmp(optarg, "binned") == 0)
                csr_kernel = SMVP_CSR_KERNEL_BINNED;
            else
                die("Unknown CSR kernel family (use auto, vector, stream, stream-carry, colsweep or binned).");
            break;
        case OPT_TIMING:
            if (strcmp(optarg, "auto") == 0)
                timing = SMVP_TIMING_AUTO;
            else if (strcmp(optarg, "events") == 0)
                timing = SMVP_TIMING_EVENTS;
            else if (strcmp(optarg, "device") == 0)
                timing = SMVP_TIMING_DEVICE;
            else if (strcmp(optarg, "device-graph") == 0)
                timing = SMVP_TIMING_DEVICE_GRAPH;
            else
                die("Unknown timing method (use auto, events, device or device-graph).");
            break;
        case OPT_TJDS_MODE:
            if (strcmp(optarg, "auto") == 0)
                tjds_mode = SMVP_TJDS_MODE_AUTO;
            else if (strcmp(optarg, "row-gather") == 0)
                tjds_mode = SMVP_TJDS_MODE_ROW_GATHER;
            else if (strcmp(optarg, "two-phase") == 0)
                tjds_mode = SMVP_TJDS_MODE_TWO_PHASE;
            else if (strcmp(optarg, "atomic") == 0)
                tjds_mode = SMVP_TJDS_MODE_ATOMIC;
            else
                die("Unknown TJDS product form (use auto, row-gather, two-phase or atomic).");
            break;
        case OPT_EXPAND:
            expand = 1;
            break;
        case OPT_CACHE:
            use_cache = 1;
            break;
        case OPT_X:
            if (strcmp(optarg, "ones") == 0)
                x_random = 0;
            else if (strcmp(optarg, "random") == 0)
                x_random = 1;
            else
                die("Unknown operand (use ones or random).");
            break;
        case OPT_DUMP:
            dump = 1;
            break;
        case OPT_USAGE:
            usage(stdout, prog);
            return 0;
        case ':':
            die("One or more options missing a required argument.");
            break;
        case '?':
        default:
            if (optopt == 0 || optopt == '?') { /* a literal -? / --help, or an unknown long option */
                if (optind > 0 && optind <= argc && argv[optind - 1] &&
                    (strcmp(argv[optind - 1], "-?") == 0 || strcmp(argv[optind - 1], "--help") == 0)) {
                    help(prog);
                    return 0;
                }
                fprintf(stderr, "%s: unknown option\n", argv[optind - 1] ? argv[optind - 1] : "?");
            } else {
                fprintf(stderr, "-%c: unknown option\n", optopt);
            }
            return 1;
        }
    }

    /* exactly one positional argument, main-cli.c:1389-1393 */
    if (optind != argc - 1) {
        usage(stderr, prog);
        fprintf(stderr, "%s: %s", RED "[ERROR]\tMust specify a single input file", "ex., /path/to/file.mtx\n" RESET);
        return 1;
    }
    const char *input = argv[optind];
    FILE *f = fopen(input, "r");
    if (!f)
        die("Specified input file not found.");

    printf(GREEN "\n[START]\tExecuting smvp-toolbox-cli v%s\n" RESET, smvp_version_string());

    smvp_mm_typecode tc;
    int rc = smvp_mm_read_banner(f, &tc);
    if (rc != SMVP_OK)
        mmio_fail(rc);
    if (tc[1] != 'C')
        die("This application only supports sparse matricies. Specified input file does not appear to contain a sparse matrix.");

    printf(MAGENTA "[FILE]\tInput matrix file name: " RESET "%s\n", input);
    printf(YELLOW "[INFO]\tLoading matrix content from source file.\n" RESET);
    int rows = 0, cols = 0, nnz = 0;
    rc = smvp_mm_read_mtx_crd_size(f, &rows, &cols, &nnz);
    if (rc != SMVP_OK)
        mmio_fail(rc);
    if (rows < 0 || cols < 0 || nnz < 0)
        mmio_fail(SMVP_MM_UNSUPPORTED_TYPE);

    /* --cache: <file>.smvpbin, valid only for the present bytes of <file> and for the same --expand-symmetric choice */
    char *cache_path = NULL;
    int cached = 0;
    smvp_coo_t *coo = NULL;
    if (use_cache) {
        cache_path = malloc(strlen(input) + 16);
        if (!cache_path)
            die("Out of memory while staging the matrix.");
        sprintf(cache_path, "%s.smvpbin", input);
        smvp_mm_typecode ctc;
        int cflags = 0, cr = 0, cc = 0, cn = 0;
        if (smvp_cache_read_header(cache_path, input, &ctc, &cflags, &cr, &cc, &cn) == SMVP_OK && cr == rows && cc == cols &&
            (cflags & 1) == expand) {
            int *rp = malloc(sizeof *rp * ((size_t)cr + 1)), *ci = malloc(sizeof *ci * (size_t)(cn > 0 ? cn : 1));
            double *cv = malloc(sizeof *cv * (size_t)(cn > 0 ? cn : 1));
            coo = malloc(sizeof *coo * (size_t)(cn > 0 ? cn : 1));
            if (!rp || !ci || !cv || !coo)
                die("Out of memory while staging the matrix.");
            if (smvp_cache_read_csr(cache_path, cr, cn, rp, ci, cv) == SMVP_OK && smvp_coo_from_csr(cr, rp, ci, cv, coo) == SMVP_OK) {
                cached = 1;
                nnz = cn;
                printf(YELLOW "[INFO]\tMatrix content taken from the binary cache %s\n" RESET, cache_path);
            } else {
                free(coo);
                coo = NULL;
            }
            free(rp);
            free(ci);
            free(cv);
        }
    }
    if (!cached) {
        coo = malloc(sizeof *coo * (size_t)(nnz > 0 ? nnz : 1)); /* heap, not a stack VLA (main-cli.c:1426) */
        if (!coo)
            die("Out of memory while staging the matrix.");
        rc = smvp_mm_read_coo_entries(f, tc, nnz, coo);
        if (rc != SMVP_OK)
            mmio_fail(rc);
        if (expand) { /* not the reference's behaviour: it multiplies the stored triangle (main-cli.c:1427-1441) */
            int full = 0;
            rc = smvp_mm_expanded_count(tc, coo, nnz, &full);
            smvp_coo_t *all = rc == SMVP_OK ? malloc(sizeof *all * (size_t)(full > 0 ? full : 1)) : NULL;
            if (rc != SMVP_OK || !all)
                engine_fail("Expanding the symmetric storage", rc != SMVP_OK ? rc : SMVP_ERR_ALLOC);
            rc = smvp_mm_expand_symmetric(tc, coo, nnz, rows, cols, all, full, &full);
            if (rc != SMVP_OK)
                engine_fail("Expanding the symmetric storage", rc);
            printf(YELLOW "[INFO]\tSymmetric storage expanded: %d stored entries -> %d.\n" RESET, nnz, full);
            free(coo);
            coo = all;
            nnz = full;
        }
        if (use_cache) {
            int *rp = malloc(sizeof *rp * ((size_t)rows + 1)), *ci = malloc(sizeof *ci * (size_t)(nnz > 0 ? nnz : 1));
            double *cv = malloc(sizeof *cv * (size_t)(nnz > 0 ? nnz : 1));
            if (rp && ci && cv && smvp_csr_from_coo(coo, rows, nnz, rp, ci, cv) == SMVP_OK &&
                smvp_cache_write_csr(cache_path, input, tc, expand, rows, cols, nnz, rp, ci, cv) == SMVP_OK)
                printf(MAGENTA "[FILE]\tBinary cache written: " RESET "%s\n", cache_path);
            else
                printf(YELLOW "[INFO]\tBinary cache not written: %s\n" RESET, smvp_last_error());
            free(rp);
            free(ci);
            free(cv);
        }
    }
    free(cache_path);
    if (f != stdin)
        fclose(f);
    double *y = malloc(sizeof *y * (size_t)(rows > 0 ? rows : 1));
    double *each = malloc(sizeof *each * (size_t)calc_iter);
    if (!y || !each)
        die("Out of memory while staging the matrix.");

    printf(CYAN "[DATA]\tNon-zero numbers contained in matrix: " RESET "%d\n", nnz);
    double *x_operand = NULL;
    if (x_random) { /* additive: the reference multiplies by ones only (main-cli.c:368-369) */
        x_operand = malloc(sizeof *x_operand * (size_t)(cols > 0 ? cols : 1));
        if (!x_operand || smvp_vector_random(x_operand, cols, 67890) != SMVP_OK)
            die("Out of memory while staging the matrix.");
        printf(CYAN "[DATA]\tVector operand in use: " RESET "Random vector (uniform [0, 1), seed 67890) with dimensions [%d, %d]\n", cols, 1);
    } else {
        printf(CYAN "[DATA]\tVector operand in use: " RESET "Ones vector with dimensions [%d, %d]\n", rows, 1);
    }
    if (iterate)
        printf(CYAN "[DATA]\tPower iteration: " RESET "each result is the next operand%s\n",
               normalize ? ", scaled to largest magnitude 1" : "");


    const int run_csr = (alg_mode == ALG_ALL) || (alg_mode & ALG_CSR);
    const int run_tjds = (alg_mode == ALG_ALL) || (alg_mode & ALG_TJDS);
    if (run_csr || run_tjds) {
        char name[256];
        int cus = 0;
        size_t mem = 0;
        rc = smvp_device_info(device, name, sizeof name, &cus, &mem);
        if (rc != SMVP_OK)
            engine_fail("Selecting the GPU", rc);
        printf(CYAN "[DATA]\tCompute device %d: " RESET "%s, %d CUs, %.0f GiB\n", device, name, cus,
               (double)mem / (1024.0 * 1024.0 * 1024.0));
        if (virtual_gpus && (exchange == SMVP_EXCHANGE_AUTO || exchange == SMVP_EXCHANGE_RCCL))
            exchange = SMVP_EXCHANGE_DIRECT; /* ranks that share a GPU: peer pushes (RCCL wants one rank per device) */
        if (ngpus > 1) {
            static const char *how[] = {"RCCL all-gather", "peer copies (hipMemcpyAsync) straight into every GPU's copy",
                                        "peer pushes (one kernel per chunk) straight into every GPU's copy",
                                        "RCCL all-gather or peer pushes, whichever is faster here (timed when the blocks are created)"};
            printf(CYAN "[DATA]\tRow blocks on %d %sGPUs, " RESET "%s of the result vector after each product\n", ngpus,
                   virtual_gpus ? "virtual " : "", how[exchange]);
        }
    }

    smvp_run_opts_t opts;
    smvp_run_opts_default(&opts);
    opts.device = device;
    opts.csr_kernel = csr_kernel;
    opts.tjds_ref_quirks = quirks;
    opts.convert_on_device = device_convert;
    opts.ngpus = ngpus;
    opts.iterate = iterate;
    opts.normalize = normalize;
    opts.timing = timing;
    opts.tjds_mode = tjds_mode;
    opts.shard_exchange = exchange;
    opts.x = x_operand;
    smvp_time_stats_t st;
    char path[4096];

    if (run_csr) {
        printf(YELLOW "[INFO]\tConverting loaded content to CSR format.\n" RESET);
        printf(YELLOW "[INFO]\tCalculating %d iterations of SMVP CSR.\n" RESET, calc_iter);
        if (dump)
            dump_csr_arrays(coo, rows, nnz);
        rc = smvp_csr_compute(coo, rows, cols, nnz, calc_iter, &opts, y, each, &st);
        if (rc != SMVP_OK)
            engine_fail("CSR product", rc);
        if (dump)
            dump_vector("[DEBUG]\tCSR JIT Vector Out:\n\t[", y, rows); /* main-cli.c:458-466 */
        rc = smvp_generate_report_text(input, report_dir, "CSR", nnz, rows, calc_iter, y, &st, 0, path, sizeof path);
        if (rc != SMVP_OK)
            engine_fail("Writing the CSR report", rc);
        printf(MAGENTA "[FILE]\tExecution report file saved as:\n" RESET);
        printf("\t%s\n", path);
        print_rates("CSR", rows, cols, nnz, -1, calc_iter, &st);
        if (dump)
            dump_csr_run(y, each, &st, rows, nnz, calc_iter);
    }
    if (run_tjds) {
        printf(YELLOW "[INFO]\tConverting loaded content to TJDS format.\n" RESET);
        if (dump)
            dump_tjds_arrays(coo, rows, cols, nnz, quirks);
        printf(YELLOW "[INFO]\tCalculating %d iterations of SMVP TJDS.\n" RESET, calc_iter);
        rc = smvp_tjds_compute(coo, rows, cols, nnz, calc_iter, &opts, y, each, &st);
        if (rc != SMVP_OK)
            engine_fail("TJDS product", rc);
        if (dump)
            dump_vector("[DEBUG]\tTJDS PHASE 8: Output Vector:\n\t[", y, rows); /* main-cli.c:1150-1158 */
        rc = smvp_generate_report_text(input, report_dir, "TJDS", nnz, rows, calc_iter, y, &st, 0, path, sizeof path);
        if (rc != SMVP_OK)
            engine_fail("Writing the TJDS report", rc);
        printf(MAGENTA "[FILE]\tExecution report file saved as:\n" RESET);
        printf("\t%s\n", path);
        print_rates("TJDS", rows, cols, nnz, 0, calc_iter, &st);
    }

    if (alg_mode != ALG_ALL && (alg_mode & ALG_CISR)) { /* main-cli.c:1473-1476; host only.  --all-algs = CSR + TJDS */
        printf(YELLOW "[INFO]\tConverting loaded content to CISR format.\n" RESET);
        rc = smvp_cisr_coegen(coo, rows, nnz, cisr_slots, stdout);
        if (rc == SMVP_ERR_UNSUPPORTED) { /* the reference's own exit, main-cli.c:596-600 */
            printf("\n[ERROR]\tslot_group_iter overran fInputNonZeros!\n");
            return 1;
        }
        if (rc != SMVP_OK)
            engine_fail("CISR COE generation", rc);
    }

    free(each);
    free(y);
    free(coo);
    free(x_operand);
    printf(GREEN "[STOP]\tExit smvp-toolbox v%s\n\n" RESET, smvp_version_string());
    return 0;
}
