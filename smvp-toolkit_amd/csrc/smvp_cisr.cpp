// smvp_cisr.cpp -- CISR .coe generator: host-only, no GPU work in it.
//
// Replaces smvp_cisr_coegen, main-cli.c:473-729: the matrix in CSR order is dealt onto `slots` channels -- every
// channel walks one row at a time and picks up the next unassigned row when its row runs out -- and the resulting
// slot groups are written as a Xilinx Vivado block-RAM initialisation file: 36-bit words, a control code (00 start,
// 01 value, 02 row lengths, 03 end) in front of a 32-bit payload (main-cli.c:672-687).
//
// The schedule is kept as one cursor per channel and run twice (count the groups, then print them) instead of the
// reference's nnz x slots table.  Behaviour kept on purpose, because it defines the output:
//   - the group after the last entry (all padding) is written too (main-cli.c:586-594);
//   - a channel that is handed an EMPTY row emits the entry that follows it (its cursor starts at row_ptr[r], which
//     for an empty row is the next row's first entry, main-cli.c:541,571);
//   - when the schedule needs as many groups as there are entries (always, with one slot) the reference prints
//     "slot_group_iter overran fInputNonZeros!" and exits (main-cli.c:596-600): SMVP_ERR_UNSUPPORTED here;
//   - the value is cast to int before it is packed (main-cli.c:703), so fractional values pack as 0.
// PARITY UNPINNED: the reference holds no .coe output and cannot be built in this image (libpopt), see DESIGN.md.
#include "smvp_common.h"

#include <vector>

namespace {

struct Channel {
    int at = 0;       // entry index this channel emits in the current group (>= nnz: padding)
    int row_end = 0;  // one past the last entry of the row it walks
};

// Moves every channel to group g (g = 0: the first); hands out rows in order.  Returns true while some channel still
// points at a real entry.
bool next_group(std::vector<Channel> &ch, bool first, const int *row_ptr, int rows, int nnz, int *next_row)
{
    bool live = false;
    for (Channel &c : ch) {
        if (first || c.at >= c.row_end - 1) {
            if (*next_row < rows) {
                c.at = row_ptr[*next_row];
                c.row_end = row_ptr[*next_row + 1];
                ++*next_row;
            } else {
                c.at = nnz + 1;
            }
        } else {
            ++c.at;
        }
        live = live || c.at < nnz;
    }
    return live;
}

}  // namespace

extern "C" int smvp_cisr_coegen(const smvp_coo_t *coo, int rows, int nnz, int slots, FILE *out)
{
    if (rows < 0 || nnz < 0 || slots < 1 || (nnz > 0 && !coo) || !out)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_cisr_coegen: bad argument");
    std::vector<int> row_ptr((size_t)rows + 1), col_ind((size_t)std::max(nnz, 1));
    std::vector<double> val((size_t)std::max(nnz, 1));
    if (int rc = smvp_csr_from_coo(coo, rows, nnz, row_ptr.data(), col_ind.data(), val.data()))
        return rc;

    // pass 1: how many groups, and does the reference give up
    int groups = 0, handed_out = 0;
    {
        std::vector<Channel> ch((size_t)slots);
        bool live = true;
        while (live) {
            live = next_group(ch, groups == 0, row_ptr.data(), rows, nnz, &handed_out);
            ++groups;
            if (groups >= nnz)
                return smvp::fail(SMVP_ERR_UNSUPPORTED, "slot_group_iter overran fInputNonZeros!");
        }
    }

    fputs("\n;*********************************************", out);
    fputs("\n;* CISR COE File for Vivado Single-Port BRAM *", out);
    fputs("\n;*********************************************\n", out);
    fprintf(out, "\n;Generated with a slot/channel count of: %d\n\n", slots);
    fputs("memory_initialization_radix=16;\n", out);
    fputs("memory_initialization_vector=\n", out);
    fputs("00aaaaaaaa,\n", out);

    // pass 2: the value words, two row lengths behind each while rows remain (rows never handed out count as 0)
    std::vector<Channel> ch((size_t)slots);
    int next_row = 0, length_row = 0;
    for (int g = 0; g < groups; ++g) {
        next_group(ch, g == 0, row_ptr.data(), rows, nnz, &next_row);
        for (int s = 0; s < slots; ++s) {
            const int at = ch[(size_t)s].at;
            uint32_t value = 0, column = 0;
            if (at < nnz) {
                const double v = val[(size_t)at];
                value = (uint32_t)((v >= -2147483648.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN);
                column = (uint32_t)col_ind[(size_t)at];
            }
            fprintf(out, "01%08x,\n", (unsigned)((value << 20) | (column << 8) | (uint32_t)s));
            if (length_row < rows) {
                auto len = [&](int r) { return r < handed_out ? (uint32_t)(row_ptr[(size_t)r + 1] - row_ptr[(size_t)r]) : 0u; };
                uint32_t word = (1u << 28) | (len(length_row) << 16);
                ++length_row;
                if (length_row < rows) {
                    word |= (1u << 12) | len(length_row);
                    ++length_row;
                }
                fprintf(out, "02%08x,\n", (unsigned)word);
            }
        }
    }
    fputs("03ffffffff;\n\n", out);
    return ferror(out) ? smvp::fail(SMVP_ERR_IO, "smvp_cisr_coegen: write failed") : SMVP_OK;
}

extern "C" int smvp_cisr_coegen_path(const smvp_coo_t *coo, int rows, int nnz, int slots, const char *path)
{
    if (!path)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_cisr_coegen_path: bad argument");
    FILE *f = fopen(path, "w");
    if (!f)
        return smvp::fail(SMVP_ERR_IO, "cannot create %s", path);
    const int rc = smvp_cisr_coegen(coo, rows, nnz, slots, f);
    if (fclose(f) != 0 && rc == SMVP_OK)
        return smvp::fail(SMVP_ERR_IO, "cannot write %s", path);
    return rc;
}
