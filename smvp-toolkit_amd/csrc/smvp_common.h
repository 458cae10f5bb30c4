// smvp_common.h -- internal helpers shared by the host-side translation units.
#pragma once
#include "smvp_amd.h"

#include <cstdarg>
#include <cstdio>

namespace smvp {
// Records a message for smvp_last_error() and hands back `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();
}  // namespace smvp
