// smvp_mmio.cpp -- Matrix Market coordinate reader for the CLI path.
//
// Fresh implementation of the three input steps the reference CLI performs:
//   banner     mm_read_banner          mmio/mmio.c:96-170
//   size line  mm_read_mtx_crd_size    mmio/mmio.c:180-208
//   entries    the loop in main()      main-cli.c:1426-1441
// Same accept/reject behaviour and return codes; the entry reader slurps the
// rest of the stream and tokenises it in memory instead of one fscanf per entry.
#include "smvp_common.h"

#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr size_t kMaxLine = 1025;  // MM_MAX_LINE_LENGTH, mmio/mmio.h:13

// Splits on blanks like scanf's %s; returns the number of tokens found (<= want).
int split_tokens(const char *line, std::string *tok, int want)
{
    int n = 0;
    const char *p = line;
    while (n < want) {
        while (*p && isspace((unsigned char)*p))
            ++p;
        if (!*p)
            break;
        const char *q = p;
        while (*q && !isspace((unsigned char)*q))
            ++q;
        tok[n++].assign(p, q);
        p = q;
    }
    return n;
}

void to_lower(std::string &s)
{
    for (char &c : s)
        c = (char)tolower((unsigned char)c);
}

// strtol-based "%d": skips blanks, needs at least one digit.
bool take_int(const char *&p, const char *end, int *out)
{
    while (p < end && isspace((unsigned char)*p))
        ++p;
    if (p >= end)
        return false;
    char *stop = nullptr;
    errno = 0;
    long v = strtol(p, &stop, 10);
    if (stop == p)
        return false;
    *out = (int)v;
    p = stop;
    return true;
}

bool take_double(const char *&p, const char *end, double *out)
{
    while (p < end && isspace((unsigned char)*p))
        ++p;
    if (p >= end)
        return false;
    char *stop = nullptr;
    double v = strtod(p, &stop);
    if (stop == p)
        return false;
    *out = v;
    p = stop;
    return true;
}

}  // namespace

extern "C" int smvp_mm_read_banner(FILE *f, smvp_mm_typecode *matcode)
{
    if (!f || !matcode)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_mm_read_banner: null argument");
    char *tc = *matcode;
    tc[0] = tc[1] = tc[2] = ' ';
    tc[3] = 'G';

    char line[kMaxLine];
    if (!fgets(line, sizeof line, f))
        return smvp::fail(SMVP_MM_PREMATURE_EOF, "matrix market: empty file");
    std::string t[5];
    if (split_tokens(line, t, 5) != 5)
        return smvp::fail(SMVP_MM_PREMATURE_EOF, "matrix market: banner needs five fields");
    for (int i = 1; i < 5; ++i)
        to_lower(t[i]);

    if (t[0].compare(0, 14, "%%MatrixMarket") != 0)
        return smvp::fail(SMVP_MM_NO_HEADER, "matrix market: missing %%%%MatrixMarket banner");
    if (t[1] != "matrix")
        return smvp::fail(SMVP_MM_UNSUPPORTED_TYPE, "matrix market: object '%s'", t[1].c_str());
    tc[0] = 'M';

    if (t[2] == "coordinate")
        tc[1] = 'C';
    else if (t[2] == "array")
        tc[1] = 'A';
    else
        return smvp::fail(SMVP_MM_UNSUPPORTED_TYPE, "matrix market: format '%s'", t[2].c_str());

    static const struct { const char *name; char code; } fields[] = {
        {"real", 'R'}, {"complex", 'C'}, {"pattern", 'P'}, {"integer", 'I'}};
    static const struct { const char *name; char code; } symm[] = {
        {"general", 'G'}, {"symmetric", 'S'}, {"hermitian", 'H'}, {"skew-symmetric", 'K'}};
    char fcode = 0, scode = 0;
    for (auto &e : fields)
        if (t[3] == e.name)
            fcode = e.code;
    if (!fcode)
        return smvp::fail(SMVP_MM_UNSUPPORTED_TYPE, "matrix market: field '%s'", t[3].c_str());
    tc[2] = fcode;
    for (auto &e : symm)
        if (t[4] == e.name)
            scode = e.code;
    if (!scode)
        return smvp::fail(SMVP_MM_UNSUPPORTED_TYPE, "matrix market: symmetry '%s'", t[4].c_str());
    tc[3] = scode;
    return SMVP_OK;
}

extern "C" int smvp_mm_read_mtx_crd_size(FILE *f, int *rows, int *cols, int *nnz)
{
    if (!f || !rows || !cols || !nnz)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_mm_read_mtx_crd_size: null argument");
    *rows = *cols = *nnz = 0;
    char line[kMaxLine];
    do {
        if (!fgets(line, sizeof line, f))
            return smvp::fail(SMVP_MM_PREMATURE_EOF, "matrix market: no size line");
    } while (line[0] == '%');

    const char *p = line, *end = line + strlen(line);
    int v[3];
    if (take_int(p, end, &v[0]) && take_int(p, end, &v[1]) && take_int(p, end, &v[2])) {
        *rows = v[0], *cols = v[1], *nnz = v[2];
        return SMVP_OK;
    }
    // A blank (or short) line after the comments: mmio keeps pulling integers
    // off the stream until it has three (mmio/mmio.c:200-205).
    int have = 0;
    for (;;) {
        int c = fgetc(f);
        if (c == EOF)
            return smvp::fail(SMVP_MM_PREMATURE_EOF, "matrix market: no size line");
        if (isspace(c))
            continue;
        ungetc(c, f);
        int val;
        if (fscanf(f, "%d", &val) == 1) {
            v[have++] = val;
            if (have == 3)
                break;
        } else {
            have = 0;  // mmio restarts its three-field scan after a mismatch
            fgetc(f);
        }
    }
    *rows = v[0], *cols = v[1], *nnz = v[2];
    return SMVP_OK;
}

namespace {

// The serial tokeniser: fscanf semantics (a number ends where it stops parsing, not at the next blank).
int parse_entries_serial(const char *p, const char *end, bool pattern, int nnz, smvp_coo_t *out)
{
    for (int i = 0; i < nnz; ++i) {
        int r, c;
        double v = 1.0;  // main-cli.c:1432
        if (!take_int(p, end, &r) || !take_int(p, end, &c) || (!pattern && !take_double(p, end, &v)))
            return smvp::fail(SMVP_MM_PREMATURE_EOF, "matrix market: entry %d of %d is missing or malformed", i + 1, nnz);
        out[i].row = r - 1;  // main-cli.c:1439-1440
        out[i].col = c - 1;
        out[i].val = v;
    }
    return SMVP_OK;
}

inline bool blank(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// Plain decimal tokens without a trip through strtol / strtod (the parallel tokeniser's hot loop: a 17-digit value costs strtod
// ~150 ns, the common 6-15 digit ones of Matrix Market files far less this way).  Both return false for anything they do not
// handle EXACTLY as the C library would -- the caller then asks the C library.
//   integers: [+-]digits, at most 9 digits (no overflow question), ended by a blank or the end of the text;
//   doubles:  [+-]digits[.digits][(e|E)[+-]digits] with at most 19 significant digits, ended the same way, and only where the
//             result is exact by construction (Clinger's fast path): the digits as an integer m <= 2^53 and a power of ten
//             10^|e| with |e| <= 22 are both exact doubles, so m * 10^e or m / 10^e is ONE correctly rounded operation --
//             the same double strtod returns (glibc's is correctly rounded too).  "inf", "nan", hex floats, longer
//             mantissas, larger exponents: not here.
inline bool token_end(const char *q, const char *end) { return q >= end || blank(*q); }

bool fast_int(const char *q, const char *end, const char **after, long *out)
{
    bool neg = false;
    if (q < end && (*q == '+' || *q == '-'))
        neg = *q++ == '-';
    const char *d0 = q;
    long v = 0;
    while (q < end && *q >= '0' && *q <= '9' && q - d0 < 10)
        v = v * 10 + (*q++ - '0');
    if (q == d0 || q - d0 > 9 || !token_end(q, end))
        return false;
    *out = neg ? -v : v;
    *after = q;
    return true;
}

bool fast_double(const char *q, const char *end, const char **after, double *out)
{
    static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    bool neg = false;
    if (q < end && (*q == '+' || *q == '-'))
        neg = *q++ == '-';
    unsigned long long m = 0;
    int digits = 0, sig = 0, frac = 0;  // digits seen, significant ones (from the first non-zero on), digits behind the point
    bool point = false;
    for (; q < end; ++q) {
        if (*q >= '0' && *q <= '9') {
            ++digits;
            if (sig > 0 || *q != '0') {
                if (++sig > 19)
                    return false;
                m = m * 10 + (unsigned)(*q - '0');
            }
            frac += point;
        } else if (*q == '.' && !point) {
            point = true;
        } else {
            break;
        }
    }
    if (digits == 0)
        return false;
    int e = 0;
    if (q < end && (*q == 'e' || *q == 'E')) {
        const char *r = q + 1;
        bool eneg = false;
        if (r < end && (*r == '+' || *r == '-'))
            eneg = *r++ == '-';
        const char *e0 = r;
        while (r < end && *r >= '0' && *r <= '9' && r - e0 < 5)
            e = e * 10 + (*r++ - '0');
        if (r == e0 || r - e0 > 4)
            return false;
        e = eneg ? -e : e;
        q = r;
    }
    if (!token_end(q, end))
        return false;
    e -= frac;
    if (m > (1ull << 53) || e < -22 || e > 22)
        return false;
    double v = (double)m;
    v = e < 0 ? v / p10[-e] : v * p10[e];
    *out = neg ? -v : v;
    *after = q;
    return true;
}

// Parallel tokeniser for large files.  The text is cut into one chunk per thread at
// blanks; pass 1 counts each chunk's tokens, a prefix sum gives every chunk its first global token number,
// pass 2 parses token g into entry g / per_entry, field g % per_entry.  A token that is not one complete
// number makes the caller fall back to the serial tokeniser, which reproduces fscanf on such input.
bool parse_entries_parallel(const char *begin, const char *end, bool pattern, int nnz, smvp_coo_t *out, int threads)
{
    const int per_entry = pattern ? 2 : 3;
    const long long want = (long long)nnz * per_entry;
    std::vector<const char *> cut((size_t)threads + 1);
    cut[0] = begin;
    cut[(size_t)threads] = end;
    for (int t = 1; t < threads; ++t) {
        const char *q = begin + (end - begin) * t / threads;
        while (q < end && !blank(*q))
            ++q;  // never split a token
        cut[(size_t)t] = q;
    }
    std::vector<long long> count((size_t)threads, 0), first((size_t)threads + 1, 0);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            long long n = 0;
            bool in = false;
            for (const char *q = cut[(size_t)t]; q < cut[(size_t)t + 1]; ++q) {
                const bool b = blank(*q);
                n += (!b && !in);
                in = !b;
            }
            count[(size_t)t] = n;
        });
    for (auto &th : pool)
        th.join();
    pool.clear();
    for (int t = 0; t < threads; ++t)
        first[(size_t)t + 1] = first[(size_t)t] + count[(size_t)t];
    if (first[(size_t)threads] < want)
        return false;  // short file: let the serial pass produce the error with its entry number

    std::vector<char> ok((size_t)threads, 1);
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            long long g = first[(size_t)t];
            const char *q = cut[(size_t)t], *stop = cut[(size_t)t + 1];
            while (g < want) {
                while (q < stop && blank(*q))
                    ++q;
                if (q >= stop)
                    break;
                const long long entry = g / per_entry;
                const int field = (int)(g % per_entry);
                const char *after = nullptr;
                if (field < 2) {
                    long v;
                    if (!fast_int(q, end, &after, &v)) {
                        char *a2 = nullptr;
                        v = strtol(q, &a2, 10);
                        after = a2;
                    }
                    (field == 0 ? out[entry].row : out[entry].col) = (int)v - 1;
                    if (pattern && field == 1)
                        out[entry].val = 1.0;
                } else if (!fast_double(q, end, &after, &out[entry].val)) {
                    char *a2 = nullptr;
                    out[entry].val = strtod(q, &a2);
                    after = a2;
                }
                if (after == q || (after < end && !blank(*after))) {
                    ok[(size_t)t] = 0;  // e.g. "1.5" in an index column: not this tokeniser's business
                    return;
                }
                q = after;
                ++g;
            }
        });
    for (auto &th : pool)
        th.join();
    for (char o : ok)
        if (!o)
            return false;
    return true;
}

}  // namespace

extern "C" int smvp_mm_read_coo_entries(FILE *f, const smvp_mm_typecode matcode, int nnz, smvp_coo_t *out)
{
    if (!f || !matcode || nnz < 0 || (nnz > 0 && !out))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_mm_read_coo_entries: bad argument");
    // Everything after the size line, in one buffer.
    std::vector<char> buf;
    {
        const long here = ftell(f);
        long size = -1;
        if (here >= 0 && fseek(f, 0, SEEK_END) == 0) {
            size = ftell(f);
            fseek(f, here, SEEK_SET);
        }
        if (size > here) {
            buf.resize((size_t)(size - here) + 1);
            const size_t got = fread(buf.data(), 1, (size_t)(size - here), f);
            buf.resize(got + 1);
        } else {  // not seekable: read in pieces
            char chunk[1 << 16];
            size_t got;
            while ((got = fread(chunk, 1, sizeof chunk, f)) > 0)
                buf.insert(buf.end(), chunk, chunk + got);
            buf.push_back('\0');
        }
        buf.back() = '\0';
    }
    const char *p = buf.data(), *end = buf.data() + buf.size() - 1;
    const bool pattern = (matcode[2] == 'P');
    // the plan option "mm_threads" overrides the thread count (1 = always serial); small files are not worth the threads
    int threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    const int asked = smvp::option("mm_threads", 0);
    if (asked > 0)
        threads = asked;
    const bool big = (end - p) >= (8 << 20) || asked > 0;
    if (threads > 1 && big && nnz > 0 && parse_entries_parallel(p, end, pattern, nnz, out, threads))
        return SMVP_OK;
    return parse_entries_serial(p, end, pattern, nnz, out);
}

extern "C" int smvp_mm_read_header_path(const char *path, smvp_mm_typecode *matcode, int *rows, int *cols, int *nnz)
{
    if (!path)
        return smvp::fail(SMVP_ERR_INVALID, "null path");
    FILE *f = fopen(path, "r");
    if (!f)
        return smvp::fail(SMVP_ERR_IO, "cannot open %s", path);
    int rc = smvp_mm_read_banner(f, matcode);
    if (rc == SMVP_OK)
        rc = smvp_mm_read_mtx_crd_size(f, rows, cols, nnz);
    fclose(f);
    return rc;
}

extern "C" int smvp_mm_read_coo_path(const char *path, smvp_coo_t *out, int capacity,
                                     smvp_mm_typecode *matcode, int *rows, int *cols, int *nnz)
{
    if (!path)
        return smvp::fail(SMVP_ERR_INVALID, "null path");
    FILE *f = fopen(path, "r");
    if (!f)
        return smvp::fail(SMVP_ERR_IO, "cannot open %s", path);
    int rc = smvp_mm_read_banner(f, matcode);
    if (rc == SMVP_OK)
        rc = smvp_mm_read_mtx_crd_size(f, rows, cols, nnz);
    if (rc == SMVP_OK && *nnz > capacity)
        rc = smvp::fail(SMVP_ERR_INVALID, "capacity %d < nnz %d", capacity, *nnz);
    if (rc == SMVP_OK)
        rc = smvp_mm_read_coo_entries(f, *matcode, *nnz, out);
    fclose(f);
    return rc;
}
