// host_san_driver.cpp -- exercises the host side of libsmvp_amd under sanitizers (make -C smvp-toolkit_amd host-san).
//
// Not part of the product: a test driver linked against the seven host translation units only (reader, converters,
// report writer, cache, CISR export, synthetic generators, error text -- no HIP), built with
// g++ -fsanitize=address,undefined and, for the parallel Matrix Market tokeniser, -fsanitize=thread (SURVEY 5,
// "race detection / sanitizers": the reference has none, CMakeLists.txt:33-34 even comments -Wall out).
// tests/test_host_sanitizers.py runs it over the reference's sample files, malformed inputs, crafted cache files.
//
//   host_san_driver file <path.mtx> <tmpdir>   everything the host library does with one input file
//   host_san_driver cache <file.smvpbin> <path.mtx>   a cache file somebody else wrote (crafted ones in the tests)
//   host_san_driver synth <tmpdir>              the generators, the partition, the report writer's edge cases
// Exit status 0 unless a call that must succeed fails (an input the library REJECTS is a pass: what is looked for is what the
// sanitizer reports, which ends the process by itself).
#include "smvp_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int g_bad = 0;
#define MUST(expr)                                                                     \
    do {                                                                               \
        const int rc_ = (expr);                                                        \
        if (rc_ != SMVP_OK) {                                                          \
            fprintf(stderr, "FAILED %s -> %d (%s)\n", #expr, rc_, smvp_last_error());  \
            ++g_bad;                                                                   \
        }                                                                              \
    } while (0)

static int run_file(const char *path, const std::string &tmp)
{
    smvp_mm_typecode tc;
    int rows = 0, cols = 0, nnz = 0;
    int rc = smvp_mm_read_header_path(path, &tc, &rows, &cols, &nnz);
    printf("header: rc %d rows %d cols %d nnz %d (%s)\n", rc, rows, cols, nnz, rc ? smvp_last_error() : "ok");
    if (rc != SMVP_OK)
        return 0;  // rejected: fine
    if (nnz < 0 || rows < 0 || cols < 0 || nnz > 50 * 1000 * 1000)
        return 0;
    std::vector<smvp_coo_t> coo((size_t)nnz + 1);
    rc = smvp_mm_read_coo_path(path, coo.data(), nnz, &tc, &rows, &cols, &nnz);
    printf("entries: rc %d (%s)\n", rc, rc ? smvp_last_error() : "ok");
    if (rc != SMVP_OK)
        return 0;
    // the reader through FILE* as well (what the command line does)
    if (FILE *f = fopen(path, "r")) {
        smvp_mm_typecode tc2;
        int r2, c2, n2;
        if (smvp_mm_read_banner(f, &tc2) == SMVP_OK && smvp_mm_read_mtx_crd_size(f, &r2, &c2, &n2) == SMVP_OK && n2 == nnz) {
            std::vector<smvp_coo_t> again((size_t)nnz + 1);
            MUST(smvp_mm_read_coo_entries(f, tc2, nnz, again.data()));
            if (memcmp(again.data(), coo.data(), sizeof(smvp_coo_t) * (size_t)nnz) != 0) {
                fprintf(stderr, "FAILED: the two readers disagree\n");
                ++g_bad;
            }
        }
        fclose(f);
    }
    bool inside = true;
    for (int i = 0; i < nnz; ++i)
        inside = inside && coo[(size_t)i].row >= 0 && coo[(size_t)i].row < rows && coo[(size_t)i].col >= 0 && coo[(size_t)i].col < cols;
    printf("entries inside the matrix: %d\n", (int)inside);
    // ---- COO -> CSR, CSR -> COO
    std::vector<int> row_ptr((size_t)rows + 1), col_ind((size_t)nnz + 1);
    std::vector<double> val((size_t)nnz + 1);
    rc = smvp_csr_from_coo(coo.data(), rows, nnz, row_ptr.data(), col_ind.data(), val.data());
    printf("csr_from_coo: rc %d\n", rc);
    if (rc == SMVP_OK) {
        std::vector<smvp_coo_t> back((size_t)nnz + 1);
        MUST(smvp_coo_from_csr(rows, row_ptr.data(), col_ind.data(), val.data(), back.data()));
        std::vector<int> bounds(9);
        MUST(smvp_partition_rows(row_ptr.data(), rows, 8, bounds.data()));
        // ---- the binary cache: write, read the header and the arrays back, then a cache made from OTHER bytes
        const std::string cache = tmp + "/m.smvpbin";
        MUST(smvp_cache_write_csr(cache.c_str(), path, tc, 0, rows, cols, nnz, row_ptr.data(), col_ind.data(), val.data()));
        smvp_mm_typecode tc3;
        int fl = 0, r3 = 0, c3 = 0, n3 = 0;
        MUST(smvp_cache_read_header(cache.c_str(), path, &tc3, &fl, &r3, &c3, &n3));
        if (r3 == rows && n3 == nnz) {
            std::vector<int> rp2((size_t)rows + 1), ci2((size_t)nnz + 1);
            std::vector<double> v2((size_t)nnz + 1);
            MUST(smvp_cache_read_csr(cache.c_str(), rows, nnz, rp2.data(), ci2.data(), v2.data()));
        }
        // truncated and bit-flipped caches must be refused, not read past
        if (FILE *f = fopen(cache.c_str(), "rb")) {
            std::vector<unsigned char> bytes;
            unsigned char b[4096];
            size_t got;
            while ((got = fread(b, 1, sizeof b, f)) > 0)
                bytes.insert(bytes.end(), b, b + got);
            fclose(f);
            for (int variant = 0; variant < 4 && bytes.size() > 80; ++variant) {
                std::vector<unsigned char> bad = bytes;
                if (variant == 0)
                    bad.resize(bad.size() / 2);
                else if (variant == 1)
                    bad.resize(40);
                else if (variant == 2)
                    bad[bad.size() - 9] ^= 0x40;  // an array byte: the checksum must catch it
                else
                    bad[70 % bad.size()] ^= 0xff;  // (past the 64-byte header: row_ptr)
                const std::string p2 = tmp + "/bad.smvpbin";
                if (FILE *g = fopen(p2.c_str(), "wb")) {
                    fwrite(bad.data(), 1, bad.size(), g);
                    fclose(g);
                    int rr = 0, cc = 0, nn = 0, ff = 0;
                    smvp_mm_typecode t4;
                    int hrc = smvp_cache_read_header(p2.c_str(), path, &t4, &ff, &rr, &cc, &nn);
                    if (hrc == SMVP_OK && rr == rows && nn == nnz) {
                        std::vector<int> rp2((size_t)rows + 1), ci2((size_t)nnz + 1);
                        std::vector<double> v2((size_t)nnz + 1);
                        hrc = smvp_cache_read_csr(p2.c_str(), rows, nnz, rp2.data(), ci2.data(), v2.data());
                    }
                    printf("crafted cache %d: rc %d\n", variant, hrc);
                    if (hrc == SMVP_OK) {
                        fprintf(stderr, "FAILED: a damaged cache file was accepted\n");
                        ++g_bad;
                    }
                }
            }
        }
    }
    // ---- COO -> TJDS
    if (inside) {
        std::vector<int> perm((size_t)cols + 1), sp((size_t)(rows > nnz ? rows : nnz) + 2), ri((size_t)nnz + 1);
        std::vector<double> tv((size_t)nnz + 1);
        int nd = 0, refn = 0, single = 0;
        rc = smvp_tjds_from_coo(coo.data(), rows, cols, nnz, perm.data(), sp.data(), (int)sp.size(), ri.data(), tv.data(), &nd, &refn, &single);
        printf("tjds_from_coo: rc %d diagonals %d ref %d single %d\n", rc, nd, refn, single);
        // too small a start_pos: refused, not overrun
        if (nd > 1) {
            std::vector<int> tiny((size_t)nd);  // needs nd + 1
            int rc2 = smvp_tjds_from_coo(coo.data(), rows, cols, nnz, perm.data(), tiny.data(), nd, ri.data(), tv.data(), &nd, nullptr, nullptr);
            printf("tjds_from_coo with a short start_pos: rc %d\n", rc2);
        }
    }
    // ---- symmetric expansion
    int count = 0;
    rc = smvp_mm_expanded_count(tc, coo.data(), nnz, &count);
    if (rc == SMVP_OK && count >= nnz && count < 120 * 1000 * 1000) {
        std::vector<smvp_coo_t> full((size_t)count + 1);
        int n_out = 0;
        rc = smvp_mm_expand_symmetric(tc, coo.data(), nnz, rows, cols, full.data(), count, &n_out);
        printf("expand: rc %d %d -> %d\n", rc, nnz, n_out);
        if (count > nnz) {  // too small a buffer: refused
            int rc2 = smvp_mm_expand_symmetric(tc, coo.data(), nnz, rows, cols, full.data(), count - 1, &n_out);
            printf("expand into a short buffer: rc %d\n", rc2);
        }
    }
    // ---- CISR export with 1 and 16 slots (1 is where the reference "overruns", main-cli.c:596-600), small matrices only
    if (inside && rows > 0 && nnz <= 400000) {
        for (int slots : {1, 16, 3}) {
            FILE *out = fopen((tmp + "/m.coe").c_str(), "w");
            if (!out)
                continue;
            rc = smvp_cisr_coegen(coo.data(), rows, nnz, slots, out);
            fclose(out);
            printf("cisr %d slots: rc %d\n", slots, rc);
        }
    }
    // ---- report writer
    if (rows > 0) {
        std::vector<double> y((size_t)rows, 1.5), each(7, 0.25);
        smvp_time_stats_t st;
        MUST(smvp_time_stats(each.data(), 7, &st));
        char outp[4096];
        MUST(smvp_generate_report_text(path, tmp.c_str(), "CSR", nnz, rows, 7, y.data(), &st, 1615284655ul, outp, sizeof outp));
        MUST(smvp_generate_report_text(path, (tmp + "/").c_str(), "TJDS", nnz, rows, 7, y.data(), &st, 0ul, outp, 8));  // short out_path
    }
    return 0;
}

static int run_synth(const std::string &tmp)
{
    for (int kind : {SMVP_SYNTH_MEMPLUS_SHAPED, SMVP_SYNTH_UNIFORM}) {
        const int64_t total = 20000, r0 = 3000, r1 = 9000;
        std::vector<int> lens((size_t)(r1 - r0)), rp((size_t)(r1 - r0) + 1, 0);
        MUST(smvp_synth_row_lengths(kind, 12345, total, total, 32, r0, r1, lens.data()));
        for (size_t i = 0; i < lens.size(); ++i)
            rp[i + 1] = rp[i] + lens[i];
        std::vector<int> ci((size_t)rp.back() + 1);
        std::vector<double> v((size_t)rp.back() + 1);
        for (int threads : {1, 4})
            MUST(smvp_synth_fill(kind, 12345, total, total, 32, r0, r1, rp.data(), ci.data(), v.data(), threads));
        for (size_t i = 0; i + 1 < rp.size(); ++i)
            for (int j = rp[i]; j < rp[i + 1]; ++j)
                if (ci[(size_t)j] < 0 || ci[(size_t)j] >= total || (j > rp[i] && ci[(size_t)j] <= ci[(size_t)j - 1])) {
                    fprintf(stderr, "FAILED: synthetic row %zu is not sorted / inside\n", i);
                    return ++g_bad;
                }
        std::vector<int> bounds(6);
        MUST(smvp_partition_rows(rp.data(), (int)(r1 - r0), 5, bounds.data()));
    }
    std::vector<double> x(1001);
    MUST(smvp_vector_random(x.data(), 1001, 67890));
    // empty and degenerate inputs
    smvp_time_stats_t st;
    double one = 3.0;
    MUST(smvp_time_stats(&one, 1, &st));
    (void)smvp_time_stats(nullptr, 0, &st);
    std::vector<int> rp1(1, 0), b(3);
    MUST(smvp_partition_rows(rp1.data(), 0, 2, b.data()));
    int rc = smvp_csr_from_coo(nullptr, 0, 0, rp1.data(), nullptr, nullptr);
    printf("empty csr_from_coo: rc %d\n", rc);
    char outp[64];
    rc = smvp_generate_report_text("x.mtx", (tmp + "/no/such/dir").c_str(), "CSR", 0, 0, 1, nullptr, &st, 1ul, outp, sizeof outp);
    printf("report into a missing directory: rc %d\n", rc);
    printf("version %s\n", smvp_version_string());
    return 0;
}

// a cache file as somebody else wrote it (tests craft ones whose checksum matches but whose arrays do not hold together)
static int run_cache(const char *cache, const char *mtx)
{
    smvp_mm_typecode tc;
    int flags = 0, rows = 0, cols = 0, nnz = 0;
    int rc = smvp_cache_read_header(cache, mtx, &tc, &flags, &rows, &cols, &nnz);
    printf("cache header: rc %d rows %d cols %d nnz %d\n", rc, rows, cols, nnz);
    if (rc != SMVP_OK || rows < 0 || nnz < 0 || nnz > 50 * 1000 * 1000)
        return 0;
    std::vector<int> rp((size_t)rows + 1), ci((size_t)nnz + 1);
    std::vector<double> v((size_t)nnz + 1);
    rc = smvp_cache_read_csr(cache, rows, nnz, rp.data(), ci.data(), v.data());
    printf("cache arrays: rc %d (%s)\n", rc, rc ? smvp_last_error() : "ok");
    if (rc == SMVP_OK) {
        std::vector<smvp_coo_t> coo((size_t)nnz + 1);
        MUST(smvp_coo_from_csr(rows, rp.data(), ci.data(), v.data(), coo.data()));
    }
    return 0;
}

int main(int argc, char **argv)
{
    // (this test driver, not the library, reads the environment: the reader's thread count for the run)
    if (const char *e = getenv("SMVP_MM_THREADS"))
        (void)smvp_set_option("mm_threads", atoi(e));
    if (argc >= 4 && !strcmp(argv[1], "file"))
        run_file(argv[2], argv[3]);
    else if (argc >= 4 && !strcmp(argv[1], "cache"))
        run_cache(argv[2], argv[3]);
    else if (argc >= 3 && !strcmp(argv[1], "synth"))
        run_synth(argv[2]);
    else {
        fprintf(stderr, "usage: %s file <path.mtx> <tmpdir> | cache <file.smvpbin> <path.mtx> | synth <tmpdir>\n", argv[0]);
        return 2;
    }
    if (g_bad)
        fprintf(stderr, "%d check(s) failed\n", g_bad);
    return g_bad ? 1 : 0;
}
