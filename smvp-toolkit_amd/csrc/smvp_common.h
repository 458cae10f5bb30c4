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

// internal, device side (smvp_convert_device.hip): see its definition
struct ihipStream_t;
namespace smvp {
int build_row_inverse(const int *d_row_ind, int nnz, int rows, int *d_inv_ptr, int *d_inv_pos, ihipStream_t *stream);
}
