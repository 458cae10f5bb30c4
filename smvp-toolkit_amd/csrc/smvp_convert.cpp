// smvp_convert.cpp -- COO -> CSR and COO -> TJDS on the host, O(nnz + rows + cols).
//
// Replaces the conversion halves of the reference's two compute functions:
//   CSR   main-cli.c:340-365   qsort by (row, col), then a branchy row_ptr fill
//   TJDS  main-cli.c:766-967   three qsorts, an O(nnz * cols) column renumbering
//                              (:894-904) and an O(rows * cols) operand permute
// Here both are stable counting sorts; the integer outputs are the same arrays
// (tests/ pin them against the oracle, which is pinned against the reference's
// committed reports).  Nothing on this path touches the GPU.
#include "smvp_common.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace {

int check_coo(const smvp_coo_t *coo, int rows, int cols, int nnz, const char *who)
{
    if (rows < 0 || cols < 0 || nnz < 0 || (nnz > 0 && !coo))
        return smvp::fail(SMVP_ERR_INVALID, "%s: bad dimensions", who);
    for (int i = 0; i < nnz; ++i)
        if (coo[i].row < 0 || coo[i].row >= rows || coo[i].col < 0 || coo[i].col >= cols)
            return smvp::fail(SMVP_ERR_INVALID, "%s: entry %d = (%d, %d) lies outside %d x %d", who, i,
                              coo[i].row, coo[i].col, rows, cols);
    return SMVP_OK;
}

// order[] = indices of coo sorted by (major, minor), input order kept for ties.
// Two stable counting passes: minor key first, then major key.
template <class Major, class Minor>
void sort_two_keys(int nnz, int n_major, int n_minor, Major major, Minor minor, std::vector<int> &order)
{
    std::vector<int> tmp((size_t)nnz), head((size_t)(n_minor > n_major ? n_minor : n_major) + 1);
    std::fill(head.begin(), head.end(), 0);
    for (int i = 0; i < nnz; ++i)
        head[(size_t)minor(i) + 1]++;
    for (int k = 0; k < n_minor; ++k)
        head[(size_t)k + 1] += head[(size_t)k];
    for (int i = 0; i < nnz; ++i)
        tmp[(size_t)head[(size_t)minor(i)]++] = i;

    std::fill(head.begin(), head.end(), 0);
    for (int i = 0; i < nnz; ++i)
        head[(size_t)major(i) + 1]++;
    for (int k = 0; k < n_major; ++k)
        head[(size_t)k + 1] += head[(size_t)k];
    order.resize((size_t)nnz);
    for (int t = 0; t < nnz; ++t) {
        int i = tmp[(size_t)t];
        order[(size_t)head[(size_t)major(i)]++] = i;
    }
}

}  // namespace

extern "C" int smvp_csr_from_coo(const smvp_coo_t *coo, int rows, int nnz,
                                 int *row_ptr, int *col_ind, double *val)
{
    if (!row_ptr || (nnz > 0 && (!col_ind || !val)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_from_coo: null output");
    int max_col = 0;
    for (int i = 0; i < nnz; ++i)
        if (coo && coo[i].col >= max_col)
            max_col = coo[i].col + 1;
    if (int rc = check_coo(coo, rows, max_col, nnz, "smvp_csr_from_coo"))
        return rc;

    std::vector<int> order;
    sort_two_keys(nnz, rows, max_col, [&](int i) { return coo[i].row; }, [&](int i) { return coo[i].col; }, order);

    memset(row_ptr, 0, sizeof(int) * ((size_t)rows + 1));
    for (int t = 0; t < nnz; ++t) {
        const smvp_coo_t &e = coo[order[(size_t)t]];
        col_ind[t] = e.col;
        val[t] = e.val;
        row_ptr[e.row + 1]++;
    }
    for (int r = 0; r < rows; ++r)
        row_ptr[r + 1] += row_ptr[r];
    return SMVP_OK;
}

extern "C" int smvp_tjds_from_coo(const smvp_coo_t *coo, int rows, int cols, int nnz,
                                  int *perm, int *start_pos, int start_pos_capacity,
                                  int *row_ind, double *val,
                                  int *num_diag, int *ref_num_tjdiag, int *last_diag_single)
{
    if ((cols > 0 && !perm) || !start_pos || !num_diag || (nnz > 0 && (!row_ind || !val)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_from_coo: null output");
    if (int rc = check_coo(coo, rows, cols, nnz, "smvp_tjds_from_coo"))
        return rc;

    // Column-major order (main-cli.c:766): entry t of `order` is the rank-th
    // stored entry of its column, and that rank IS its jagged-diagonal number
    // (the "vertical compression" of main-cli.c:789-826).
    std::vector<int> order;
    sort_two_keys(nnz, cols, rows, [&](int i) { return coo[i].col; }, [&](int i) { return coo[i].row; }, order);

    std::vector<int> col_len((size_t)cols + 1, 0);
    for (int i = 0; i < nnz; ++i)
        col_len[(size_t)coo[i].col]++;
    int longest = 0;
    for (int c = 0; c < cols; ++c)
        if (col_len[(size_t)c] > longest)
            longest = col_len[(size_t)c];
    if (longest + 1 > start_pos_capacity)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_from_coo: %d diagonals need start_pos_capacity >= %d",
                          longest, longest + 1);

    // Permutation (main-cli.c:845-868): columns by length descending, equal
    // lengths by original index ascending == one stable counting pass on length.
    std::vector<int> bucket((size_t)longest + 2, 0);
    for (int c = 0; c < cols; ++c)
        bucket[(size_t)(longest - col_len[(size_t)c]) + 1]++;  // key 0 = longest
    for (int k = 0; k <= longest; ++k)
        bucket[(size_t)k + 1] += bucket[(size_t)k];
    std::vector<int> where((size_t)cols + 1);
    for (int c = 0; c < cols; ++c) {
        int k = bucket[(size_t)(longest - col_len[(size_t)c])]++;
        perm[k] = c;
        where[(size_t)c] = k;
    }

    // Diagonal d holds one entry from each column longer than d; sorted as they
    // are, those are exactly permuted columns 0 .. count_d-1, so the entry of
    // (column c, rank d) lands at start_pos[d] + where[c]  (main-cli.c:926-967).
    std::vector<int> count((size_t)longest + 1, 0);
    for (int c = 0; c < cols; ++c)
        if (col_len[(size_t)c] > 0)
            count[(size_t)col_len[(size_t)c] - 1]++;  // columns of length exactly len
    // suffix sums: count_d = number of columns with length > d
    for (int d = longest - 2; d >= 0; --d)
        count[(size_t)d] += count[(size_t)d + 1];
    start_pos[0] = 0;
    for (int d = 0; d < longest; ++d)
        start_pos[d + 1] = start_pos[d] + count[(size_t)d];

    int t = 0;
    for (int c = 0; c < cols; ++c) {          // `order` is grouped by column, ascending
        const int k = where[(size_t)c];
        for (int d = 0; d < col_len[(size_t)c]; ++d, ++t) {
            const smvp_coo_t &e = coo[order[(size_t)t]];
            const int j = start_pos[d] + k;
            row_ind[j] = e.row;
            val[j] = e.val;
        }
    }

    *num_diag = longest;
    if (ref_num_tjdiag)
        *ref_num_tjdiag = cols > 0 ? col_len[0] : 0;  // main-cli.c:865, before the sort
    if (last_diag_single)
        *last_diag_single = (longest > 0 && count[(size_t)longest - 1] == 1) ? 1 : 0;
    return SMVP_OK;
}

extern "C" int smvp_partition_rows(const int *row_ptr, int rows, int parts, int *bounds)
{
    if (!row_ptr || !bounds || rows < 0 || parts < 1)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_partition_rows: bad argument");
    // Cut where the running cost (12 B per entry + 20 B per row, the algorithmic
    // bytes of SURVEY 8(d)) crosses p/parts of the total.
    const double total = 12.0 * row_ptr[rows] + 20.0 * rows;
    bounds[0] = 0;
    int r = 0;
    for (int p = 1; p < parts; ++p) {
        const double want = total * p / parts;
        while (r < rows && 12.0 * row_ptr[r] + 20.0 * r < want)
            ++r;
        bounds[p] = r;
    }
    bounds[parts] = rows;
    return SMVP_OK;
}
