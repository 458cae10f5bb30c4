// smvp_cache.cpp -- two optional steps between the Matrix Market reader and the format converters
// (SURVEY 8(f) row 2); neither exists in the reference and both are off unless asked for.
//
//   symmetric expansion   the reference multiplies the stored triangle of a symmetric file as it stands
//                         (main-cli.c:1427-1441 never looks at the symmetry field), and so does this engine by
//                         default; smvp_mm_expand_symmetric mirrors the off-diagonal entries so that A_full x is
//                         computed instead.
//   binary cache          parsing the text is the slowest step for a large file; <file>.smvpbin keeps the matrix
//                         as CSR arrays behind a header that names the .mtx it was made from (size + FNV-1a 64 of
//                         its bytes), so a stale or foreign cache is never used.
#include "smvp_common.h"

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr char kMagic[8] = {'S', 'M', 'V', 'P', 'B', 'I', 'N', '1'};

struct CacheHeader {
    char magic[8];
    uint32_t version;       // 1
    uint32_t flags;         // bit 0: symmetric storage was expanded before conversion
    int32_t rows, cols, nnz;
    char typecode[4];       // of the source file
    uint64_t mtx_bytes;     // size of the source .mtx
    uint64_t mtx_fnv1a;     // FNV-1a 64 of its contents
    uint64_t payload_fnv1a; // of row_ptr | col_ind | val as stored
    uint64_t reserved;      // 0
};
static_assert(sizeof(CacheHeader) == 64, "the cache header is 64 bytes");

inline uint64_t fnv1a(uint64_t h, const void *data, size_t n)
{
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
constexpr uint64_t kFnvSeed = 0xcbf29ce484222325ull;

int file_digest(const char *path, uint64_t *bytes, uint64_t *digest)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return smvp::fail(SMVP_ERR_IO, "cannot open %s", path);
    std::vector<unsigned char> buf(1 << 20);
    uint64_t h = kFnvSeed, total = 0;
    size_t got;
    while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
        h = fnv1a(h, buf.data(), got);
        total += got;
    }
    fclose(f);
    *bytes = total;
    *digest = h;
    return SMVP_OK;
}

int mirror_sign(const smvp_mm_typecode tc)  // +1 symmetric / hermitian (real field), -1 skew-symmetric, 0 general
{
    return tc[3] == 'S' || tc[3] == 'H' ? 1 : (tc[3] == 'K' ? -1 : 0);
}

}  // namespace

extern "C" int smvp_mm_expanded_count(const smvp_mm_typecode matcode, const smvp_coo_t *coo, int nnz, int *count)
{
    if (!matcode || nnz < 0 || (nnz > 0 && !coo) || !count)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_mm_expanded_count: bad argument");
    long long n = nnz;
    if (mirror_sign(matcode) != 0)
        for (int i = 0; i < nnz; ++i)
            n += coo[i].row != coo[i].col;
    if (n > 2147483647ll)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "the expanded matrix has %lld entries: beyond 32-bit indices", n);
    *count = (int)n;
    return SMVP_OK;
}

extern "C" int smvp_mm_expand_symmetric(const smvp_mm_typecode matcode, const smvp_coo_t *coo, int nnz, int rows, int cols,
                                        smvp_coo_t *out, int capacity, int *nnz_out)
{
    int want = 0;
    if (int rc = smvp_mm_expanded_count(matcode, coo, nnz, &want))
        return rc;
    if (!out || capacity < want || !nnz_out)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_mm_expand_symmetric: %d entries need capacity >= %d", nnz, want);
    const int sign = mirror_sign(matcode);
    if (sign != 0 && rows != cols)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_mm_expand_symmetric: a %d x %d matrix cannot be symmetric", rows, cols);
    // stored entries first, in file order, then the mirrored ones in file order: the converters sort anyway
    int n = 0;
    for (int i = 0; i < nnz; ++i)
        out[n++] = coo[i];
    if (sign != 0)
        for (int i = 0; i < nnz; ++i)
            if (coo[i].row != coo[i].col) {
                out[n].row = coo[i].col;
                out[n].col = coo[i].row;
                out[n].val = sign > 0 ? coo[i].val : -coo[i].val;
                ++n;
            }
    *nnz_out = n;
    return SMVP_OK;
}

extern "C" int smvp_cache_write_csr(const char *cache_path, const char *mtx_path, const smvp_mm_typecode matcode, int flags,
                                    int rows, int cols, int nnz, const int *row_ptr, const int *col_ind, const double *val)
{
    if (!cache_path || !mtx_path || !matcode || rows < 0 || cols < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!col_ind || !val)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_cache_write_csr: bad argument");
    CacheHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, kMagic, 8);
    h.version = 1;
    h.flags = (uint32_t)flags;
    h.rows = rows, h.cols = cols, h.nnz = nnz;
    memcpy(h.typecode, matcode, 4);
    if (int rc = file_digest(mtx_path, &h.mtx_bytes, &h.mtx_fnv1a))
        return rc;
    uint64_t p = fnv1a(kFnvSeed, row_ptr, sizeof(int) * ((size_t)rows + 1));
    p = fnv1a(p, col_ind, sizeof(int) * (size_t)nnz);
    h.payload_fnv1a = fnv1a(p, val, sizeof(double) * (size_t)nnz);
    FILE *f = fopen(cache_path, "wb");
    if (!f)
        return smvp::fail(SMVP_ERR_IO, "cannot create %s", cache_path);
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(row_ptr, sizeof(int), (size_t)rows + 1, f) == (size_t)rows + 1 &&
              fwrite(col_ind, sizeof(int), (size_t)nnz, f) == (size_t)nnz && fwrite(val, sizeof(double), (size_t)nnz, f) == (size_t)nnz;
    ok = fclose(f) == 0 && ok;
    if (!ok) {
        remove(cache_path);
        return smvp::fail(SMVP_ERR_IO, "short write to %s", cache_path);
    }
    return SMVP_OK;
}

// Header of a cache file, checked against the .mtx it claims to come from (mtx_path NULL = no such check).
// SMVP_ERR_IO: no readable cache; SMVP_ERR_INVALID: not a cache, another version, or made from other bytes.
extern "C" int smvp_cache_read_header(const char *cache_path, const char *mtx_path, smvp_mm_typecode *matcode, int *flags,
                                      int *rows, int *cols, int *nnz)
{
    if (!cache_path)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_cache_read_header: bad argument");
    FILE *f = fopen(cache_path, "rb");
    if (!f)
        return smvp::fail(SMVP_ERR_IO, "cannot open %s", cache_path);
    CacheHeader h;
    const bool got = fread(&h, sizeof h, 1, f) == 1;
    fclose(f);
    if (!got || memcmp(h.magic, kMagic, 8) != 0 || h.version != 1 || h.rows < 0 || h.cols < 0 || h.nnz < 0)
        return smvp::fail(SMVP_ERR_INVALID, "%s is not a version-1 smvpbin cache", cache_path);
    if (mtx_path) {
        uint64_t bytes = 0, digest = 0;
        if (int rc = file_digest(mtx_path, &bytes, &digest))
            return rc;
        if (bytes != h.mtx_bytes || digest != h.mtx_fnv1a)
            return smvp::fail(SMVP_ERR_INVALID, "%s was not made from the present contents of %s", cache_path, mtx_path);
    }
    if (matcode)
        memcpy(*matcode, h.typecode, 4);
    if (flags)
        *flags = (int)h.flags;
    if (rows)
        *rows = h.rows;
    if (cols)
        *cols = h.cols;
    if (nnz)
        *nnz = h.nnz;
    return SMVP_OK;
}

extern "C" int smvp_cache_read_csr(const char *cache_path, int rows, int nnz, int *row_ptr, int *col_ind, double *val)
{
    if (!cache_path || rows < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!col_ind || !val)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_cache_read_csr: bad argument");
    FILE *f = fopen(cache_path, "rb");
    if (!f)
        return smvp::fail(SMVP_ERR_IO, "cannot open %s", cache_path);
    CacheHeader h;
    bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, kMagic, 8) == 0 && h.version == 1 && h.rows == rows && h.nnz == nnz;
    ok = ok && fread(row_ptr, sizeof(int), (size_t)rows + 1, f) == (size_t)rows + 1 &&
         fread(col_ind, sizeof(int), (size_t)nnz, f) == (size_t)nnz && fread(val, sizeof(double), (size_t)nnz, f) == (size_t)nnz;
    fclose(f);
    if (!ok)
        return smvp::fail(SMVP_ERR_INVALID, "%s is truncated or does not hold a %d-row, %d-entry matrix", cache_path, rows, nnz);
    uint64_t p = fnv1a(kFnvSeed, row_ptr, sizeof(int) * ((size_t)rows + 1));
    p = fnv1a(p, col_ind, sizeof(int) * (size_t)nnz);
    p = fnv1a(p, val, sizeof(double) * (size_t)nnz);
    if (p != h.payload_fnv1a)
        return smvp::fail(SMVP_ERR_INVALID, "%s is damaged (payload checksum)", cache_path);
    // The checksum is no integrity guarantee (a crafted file carries a matching one): what the callers index with --
    // smvp_coo_from_csr writes out[row_ptr[r] .. row_ptr[r + 1]), the kernels gather x[col_ind[j]] -- is checked itself.
    if (row_ptr[0] != 0 || row_ptr[rows] != nnz)
        return smvp::fail(SMVP_ERR_INVALID, "%s holds an inconsistent row pointer", cache_path);
    for (int r = 0; r < rows; ++r)
        if (row_ptr[r] > row_ptr[r + 1])
            return smvp::fail(SMVP_ERR_INVALID, "%s: row pointer decreases at row %d", cache_path, r);
    for (int j = 0; j < nnz; ++j)
        if (col_ind[j] < 0 || col_ind[j] >= h.cols)
            return smvp::fail(SMVP_ERR_INVALID, "%s: column index %d of entry %d lies outside [0, %d)", cache_path, col_ind[j], j, h.cols);
    return SMVP_OK;
}

// CSR -> entries in (row, col) order, for the entry points that take COO like the reference's do
extern "C" int smvp_coo_from_csr(int rows, const int *row_ptr, const int *col_ind, const double *val, smvp_coo_t *out)
{
    if (rows < 0 || !row_ptr || (row_ptr[rows] > 0 && (!col_ind || !val || !out)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_coo_from_csr: bad argument");
    for (int r = 0; r < rows; ++r)  // out[] has row_ptr[rows] entries: nothing may be written outside it
        if (row_ptr[r] < 0 || row_ptr[r] > row_ptr[r + 1] || row_ptr[r + 1] > row_ptr[rows])
            return smvp::fail(SMVP_ERR_INVALID, "smvp_coo_from_csr: row_ptr is not a non-decreasing sequence ending at row_ptr[rows]");
    for (int r = 0; r < rows; ++r)
        for (int j = row_ptr[r]; j < row_ptr[r + 1]; ++j) {
            out[j].row = r;
            out[j].col = col_ind[j];
            out[j].val = val[j];
        }
    return SMVP_OK;
}
