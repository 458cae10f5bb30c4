// smvp_error.cpp -- thread-local last-error text, version string, and the plan options for experiments and tests.
#include "smvp_common.h"

#include <atomic>
#include <cstring>

namespace {
thread_local char g_err[512] = "";
}

namespace smvp {
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
void clear_error() { g_err[0] = '\0'; }
}  // namespace smvp

extern "C" const char *smvp_last_error(void) { return g_err; }

extern "C" const char *smvp_version_string(void)
{
    static char v[32];
    snprintf(v, sizeof v, "%d.%d.%d", SMVP_VERSION_MAJOR, SMVP_VERSION_MINOR, SMVP_VERSION_REVISION);
    return v;
}

// ---- plan options (smvp_set_option): what used to be environment switches.  Read where a plan or a handle is built or a file is
// parsed, never on a launch path; -1 = not set (the library's default applies).  Process-wide, lock-free.
namespace {
struct Option { const char *name; std::atomic<int> value; };
Option g_options[] = {
    {"csr_col16", {-1}},        // 0: the CSR tile kernel keeps col_ind's 32-bit columns (default: 16-bit offsets where a tile's columns lie close)
    {"csr_rowrel", {-1}},       // 0: the tile kernel's second phase reads row_ptr (default: its own 16-bit row offsets)
    {"binned_near", {-1}},      // 1: the binned plan's near part runs on the tile kernel (default 0: csr_near_window where it suits)
    {"binned_overlap", {-1}},   // 0: pass A behind the near part on one stream (default 1: beside it on a stream of its own)
    {"tjds_index", {-1}},       // the one-kernel TJDS product's index stream: 0 = 16-bit position words (default), 1 = 32-bit sorted, 2 = 32-bit columns
    {"sharded_threads", {-1}},  // 1: smvp_sharded_* uses its issuing threads with ONE GPU too (default 0: the caller's thread)
    {"mm_threads", {-1}},       // threads of the Matrix Market tokeniser (default: up to 16 for files of 8 MB and more; 1 = serial; n > 1 forces the parallel path)
};
Option *find_option(const char *name)
{
    if (name)
        for (Option &o : g_options)
            if (!strcmp(o.name, name))
                return &o;
    return nullptr;
}
}  // namespace

namespace smvp {
int option(const char *name, int fallback)
{
    const Option *o = find_option(name);
    const int v = o ? o->value.load(std::memory_order_relaxed) : -1;
    return v < 0 ? fallback : v;
}
}  // namespace smvp

extern "C" int smvp_set_option(const char *name, int value)
{
    Option *o = find_option(name);
    if (!o)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_set_option: no option called '%s'", name ? name : "(null)");
    o->value.store(value < 0 ? -1 : value, std::memory_order_relaxed);
    return SMVP_OK;
}

extern "C" int smvp_get_option(const char *name, int *value)
{
    const Option *o = find_option(name);
    if (!o || !value)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_get_option: no option called '%s'", name ? name : "(null)");
    *value = o->value.load(std::memory_order_relaxed);
    return SMVP_OK;
}
