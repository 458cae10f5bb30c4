// smvp_error.cpp -- thread-local last-error text and version string.
#include "smvp_common.h"

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
