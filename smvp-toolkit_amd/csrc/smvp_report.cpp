// smvp_report.cpp -- timing statistics and the per-algorithm report file.
//
// Replaces main-cli.c:428-456 (total/avg/min/max), calcStDevDouble
// (main-cli.c:114-130) and generateReportText (main-cli.c:246-320).  The report
// text is byte-compatible with the reference's: tests compare it against the
// committed files in output-test/ with the timing and timestamp lines masked.
//
// The five timing lines print the stats of whatever time_each_ms[] the caller hands in.  From smvp_*_compute under
// SMVP_TIMING_AUTO that is, for launches of up to 4096 workgroups (the reference's own sample matrices), the IN-KERNEL
// window of each product -- shorter than the host-side clock_gettime bracket of main-cli.c:408-419 by the launch
// (memplus.mtx: 2.9 us against 4.4 us of loop wall per product).  SMVP_TIMING_EVENTS (CLI: --timing events) gives the
// host-comparable figure; INTEGRATION.md section 2 says which to read when comparing with output-test/*.txt.
#include "smvp_common.h"

#include <cmath>
#include <cstring>
#include <ctime>
#include <string>

extern "C" int smvp_time_stats(const double *ms, int iters, smvp_time_stats_t *out)
{
    if (!out || iters < 0 || (iters > 0 && !ms))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_time_stats: bad argument");
    double total = 0.0, lo = 0.0, hi = 0.0;
    for (int i = 0; i < iters; ++i) {
        total += ms[i];
        lo = (i == 0 || ms[i] < lo) ? ms[i] : lo;
        hi = (i == 0 || ms[i] > hi) ? ms[i] : hi;
    }
    const double mean = iters ? total / iters : 0.0;
    double ss = 0.0;
    for (int i = 0; i < iters; ++i)
        ss += (ms[i] - mean) * (ms[i] - mean);
    out->time_total = total;
    out->time_avg = mean;
    out->time_min = lo;
    out->time_max = hi;
    out->time_stdev = iters ? std::sqrt(ss / iters) : 0.0;
    return SMVP_OK;
}

extern "C" int smvp_generate_report_text(const char *input_file_name, const char *report_dir,
                                         const char *alg_name, int nnz, int rows, int iters,
                                         const double *y, const smvp_time_stats_t *st,
                                         unsigned long unix_time, char *out_path, size_t out_path_cap)
{
    if (!input_file_name || !alg_name || !st || rows < 0 || (rows > 0 && !y))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_generate_report_text: bad argument");
    if (unix_time == 0)
        unix_time = (unsigned long)time(nullptr);

    std::string path;
    if (report_dir && report_dir[0]) {
        path = report_dir;
        if (path.back() != '/')
            path += '/';
    }
    char name[128];
    snprintf(name, sizeof name, "smvp-toolbox_report_%s_%lu.txt", alg_name, unix_time);
    path += name;
    if (out_path && out_path_cap)
        snprintf(out_path, out_path_cap, "%s", path.c_str());

    FILE *f = fopen(path.c_str(), "a+");
    if (!f)
        return smvp::fail(SMVP_ERR_IO, "cannot open report file %s", path.c_str());
    fprintf(f, "Execution results for smvp-toolbox v.%s, %s algorithm\n", smvp_version_string(), alg_name);
    fprintf(f, "Generated on %lu (Unix time)\n\n", unix_time);
    fprintf(f, "Sparse matrix file in use:\n%s\n\n", input_file_name);
    fprintf(f, "Non-zero numbers contained in matrix: %d\n\n", nnz);
    fprintf(f, "Compute times for %d iterations:\n\n", iters);
    fprintf(f, "Total Time: %g ms\n", st->time_total);
    fprintf(f, "Average Time: %g ms\n", st->time_avg);
    fprintf(f, "Fastest Time: %g ms\n", st->time_min);
    fprintf(f, "Slowest Time: %g ms\n", st->time_max);
    fprintf(f, "Time StDev: %g ms\n\n", st->time_stdev);
    fputs("Output vector (one cell per line):\n[\n", f);
    for (int r = 0; r < rows; ++r)
        fprintf(f, "%g\n", y[r]);
    // The reference closes the bracket from inside its loop, so an empty vector
    // never gets one (main-cli.c:306-317).
    if (rows > 0)
        fputs("]\n\n", f);
    const bool bad = ferror(f);
    if (fclose(f) != 0 || bad)
        return smvp::fail(SMVP_ERR_IO, "short write to %s", path.c_str());
    return SMVP_OK;
}
