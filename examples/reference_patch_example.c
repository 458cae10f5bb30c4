/*
 * reference_patch_example.c -- INTEGRATION.md section 2 as a translation unit that compiles.
 *
 * The reference's own types are re-declared here with the layout main-cli.c gives them (MMRawData :42-47,
 * struct _time_data_ :87-95, newResultsData :99-110) so that the dispatch blocks of main() (:1453-1471), rewritten onto
 * libsmvp_amd.so exactly as INTEGRATION.md shows, can be built and run without libpopt.  tests/test_integration_example.py
 * compiles it against include/smvp_amd.h on the CPU and runs it on the GPU box against the reference's committed report.
 *
 *   gcc -O2 -std=c11 examples/reference_patch_example.c -Iinclude -Lsmvp-toolkit_amd/lib -lsmvp_amd \
 *       -Wl,-rpath,$PWD/smvp-toolkit_amd/lib -lm -o /tmp/patch_example
 *   /tmp/patch_example tests/golden/sample-data/ibm32.mtx 100 /tmp
 */
#include "smvp_amd.h"

#include <stdio.h>
#include <stdlib.h>

/* ---- the reference's types (main-cli.c) ---- */
typedef struct _mm_raw_data_ {
    int row;
    int col;
    double val;
} MMRawData; /* :42-47 */

struct _time_data_ {
    double time_total, time_avg, time_stdev, time_min, time_max;
    int _itercount;
    double time_each[]; /* :87-95 flexible array */
};

static struct _time_data_ *newResultsData(struct _time_data_ *t, int iters) /* :99-110 */
{
    t = malloc(sizeof *t + sizeof(double) * (size_t)iters);
    t->_itercount = iters;
    return t;
}

#define ALG_CSR (1 << 1)
#define ALG_TJDS (1 << 2)

/* what makes the cast in INTEGRATION.md legal */
_Static_assert(sizeof(MMRawData) == sizeof(smvp_coo_t), "MMRawData and smvp_coo_t have one size");
_Static_assert(__builtin_offsetof(MMRawData, col) == __builtin_offsetof(smvp_coo_t, col) &&
                   __builtin_offsetof(MMRawData, val) == __builtin_offsetof(smvp_coo_t, val),
               "MMRawData and smvp_coo_t have one layout");

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s file.mtx iterations report_dir\n", argv[0]);
        return 2;
    }
    const char *inputFileName = argv[1], *reportPath = argv[3];
    const int calc_iter = atoi(argv[2]), alg_mode = ALG_CSR | ALG_TJDS;

    /* main-cli.c:1398-1441 with the reader swapped one for one */
    FILE *mmInputFile = fopen(inputFileName, "r");
    if (!mmInputFile)
        return 1;
    smvp_mm_typecode matcode;
    int fInputRows, fInputCols, fInputNonZeros;
    if (smvp_mm_read_banner(mmInputFile, &matcode) != 0 ||
        smvp_mm_read_mtx_crd_size(mmInputFile, &fInputRows, &fInputCols, &fInputNonZeros) != 0)
        return 1;
    MMRawData *mmImportData = malloc(sizeof *mmImportData * (size_t)(fInputNonZeros ? fInputNonZeros : 1));
    if (smvp_mm_read_coo_entries(mmInputFile, matcode, fInputNonZeros, (smvp_coo_t *)mmImportData) != 0)
        return 1;
    fclose(mmInputFile);

    /* main-cli.c:1453-1471 -- the two dispatch blocks */
    smvp_run_opts_t opts;
    smvp_run_opts_default(&opts); /* device 0, AUTO kernel, x = ones */

    if (alg_mode & ALG_CSR) {
        struct _time_data_ *csr_time = NULL;
        csr_time = newResultsData(csr_time, calc_iter);
        double *output_vector_csr = malloc(sizeof(double) * (size_t)fInputRows);
        smvp_time_stats_t st;
        /* was: smvp_csr_compute(mmImportData, fInputRows, fInputNonZeros, calc_iter, csr_time)  (main-cli.c:325) */
        int rc = smvp_csr_compute((const smvp_coo_t *)mmImportData, fInputRows, fInputCols, fInputNonZeros, calc_iter, &opts,
                                  output_vector_csr, csr_time->time_each, &st);
        if (rc != SMVP_OK) {
            printf("[ERROR]\t%s\n", smvp_last_error());
            exit(1);
        }
        csr_time->time_total = st.time_total, csr_time->time_avg = st.time_avg;
        csr_time->time_min = st.time_min, csr_time->time_max = st.time_max, csr_time->time_stdev = st.time_stdev;
        /* generateReportText(inputFileName, reportPath, ALG_CSR, ...) or its replacement: */
        char path[4096];
        rc = smvp_generate_report_text(inputFileName, reportPath, "CSR", fInputNonZeros, fInputRows, calc_iter,
                                       output_vector_csr, &st, 0, path, sizeof path);
        if (rc != SMVP_OK)
            exit(1);
        printf("CSR report: %s (avg %g ms over %d products, first %g ms)\n", path, csr_time->time_avg, calc_iter,
               csr_time->time_each[0]);
    }
    if (alg_mode & ALG_TJDS) {
        struct _time_data_ *tjds_time = NULL;
        tjds_time = newResultsData(tjds_time, calc_iter);
        double *output_vector_tjds = malloc(sizeof(double) * (size_t)fInputRows);
        smvp_time_stats_t st;
        /* was: smvp_tjds_compute(mmImportData, fInputRows, fInputCols, fInputNonZeros, calc_iter, tjds_time)  (main-cli.c:734) */
        int rc = smvp_tjds_compute((const smvp_coo_t *)mmImportData, fInputRows, fInputCols, fInputNonZeros, calc_iter,
                                   &opts, output_vector_tjds, tjds_time->time_each, &st);
        if (rc != SMVP_OK) {
            printf("[ERROR]\t%s\n", smvp_last_error());
            exit(1);
        }
        char path[4096];
        rc = smvp_generate_report_text(inputFileName, reportPath, "TJDS", fInputNonZeros, fInputRows, calc_iter,
                                       output_vector_tjds, &st, 0, path, sizeof path);
        if (rc != SMVP_OK)
            exit(1);
        printf("TJDS report: %s (avg %g ms)\n", path, st.time_avg);
    }
    return 0;
}
