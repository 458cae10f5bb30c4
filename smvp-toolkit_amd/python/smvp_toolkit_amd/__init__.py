"""ctypes binding for libsmvp_amd.so (include/smvp_amd.h).

The product is the C-ABI library plus the C command-line program; this module
only exposes that ABI to Python for the tests and for bench.py.  It holds no
compute of its own and there is no fallback: if the library is missing, or no
HIP device is visible, the calls fail.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))          # smvp-toolkit_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
LIB_PATH = os.environ.get("SMVP_LIB_PATH") or os.path.join(PKG_ROOT, "lib", "libsmvp_amd.so")   # (the override: A/B of two builds in one gpurun call)
CLI_PATH = os.path.join(PKG_ROOT, "bin", "smvp-toolkit-cli")

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_IO, ERR_UNSUPPORTED = 1, 2, 3, 4, 5, 6
MM_COULD_NOT_READ_FILE, MM_PREMATURE_EOF, MM_NOT_MTX, MM_NO_HEADER, MM_UNSUPPORTED_TYPE = 11, 12, 13, 14, 15
CSR_KERNEL_AUTO, CSR_KERNEL_VECTOR, CSR_KERNEL_STREAM, CSR_KERNEL_STREAM_CARRY, CSR_KERNEL_COLSWEEP = 0, 1, 2, 3, 4
CSR_KERNEL_BINNED = 5
MEM_HOST, MEM_DEVICE = 0, 1
SYNTH_MEMPLUS_SHAPED, SYNTH_UNIFORM = 1, 2
TJDS_MODE_AUTO, TJDS_MODE_ATOMIC, TJDS_MODE_TWO_PHASE, TJDS_MODE_ROW_GATHER = 0, 1, 2, 3
TIMING_AUTO, TIMING_EVENTS, TIMING_DEVICE, TIMING_DEVICE_GRAPH = 0, 1, 2, 3

COO_DTYPE = np.dtype([("row", "<i4"), ("col", "<i4"), ("val", "<f8")], align=True)


def set_option(name, value):
    """smvp_set_option: a plan option for experiments and tests (include/smvp_amd.h); value < 0 or None = the library's default."""
    _check(lib().smvp_set_option(name.encode(), -1 if value is None else int(value)), "smvp_set_option")


def get_option(name):
    v = C.c_int()
    _check(lib().smvp_get_option(name.encode(), C.byref(v)), "smvp_get_option")
    return v.value


class option:
    """with sm.option("csr_col16", 0): ...  -- sets a plan option for the block and restores what was there."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def sweep_parts(rows_per_block, parts, chunks=0):
    """SMVP_CSR_SWEEP_PARTS / SMVP_CSR_SWEEP_PARAM: the column sweep's kernel parameter with 2 or 4 column parts per strip and,
    for experiments, the chunks a wavefront keeps in flight forced to 1, 2 or 4 (include/smvp_amd.h)."""
    return int(rows_per_block) | ({1: 0, 2: 1, 4: 2, 8: 3}[int(parts)] << 24) | ({0: 0, 1: 1, 2: 2, 4: 3}[int(chunks)] << 26)


class TimeStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("time_total", "time_avg", "time_stdev", "time_min", "time_max")]


class RunOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint), ("device", C.c_int), ("csr_kernel", C.c_int), ("csr_param", C.c_int),
                ("tjds_ref_quirks", C.c_int), ("convert_on_device", C.c_int), ("ngpus", C.c_int),
                ("iterate", C.c_int), ("normalize", C.c_int), ("tjds_mode", C.c_int), ("timing", C.c_int),
                ("shard_exchange", C.c_int), ("repeat_patience_us", C.c_int), ("x", C.c_void_p)]


class PlanInfo(C.Structure):
    _fields_ = [("matrix_bytes", C.c_double), ("plan_bytes", C.c_double), ("build_ms", C.c_double)]


class RunInfo(C.Structure):
    _fields_ = [("timing", C.c_int), ("graph_replays", C.c_int), ("wall_ms", C.c_double), ("device_clock_khz", C.c_double),
                ("repeat_launches", C.c_int), ("repeat_gave_up", C.c_int)]


class ShardOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint), ("chunks", C.c_int), ("balance", C.c_int), ("exchange", C.c_int)]


GATHER_NONE, GATHER_OVERLAPPED, GATHER_AFTER = 0, 1, 2
EXCHANGE_RCCL, EXCHANGE_COPIES, EXCHANGE_DIRECT, EXCHANGE_AUTO = 0, 1, 2, 3
EXCHANGE_NAMES = {EXCHANGE_RCCL: "rccl", EXCHANGE_COPIES: "copies", EXCHANGE_DIRECT: "direct", EXCHANGE_AUTO: "auto"}


class SmvpError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().smvp_last_error().decode(errors="replace")
        super().__init__("%s failed with status %d: %s" % (where, code, msg))


_lib = None

# every symbol include/smvp_amd.h declares; tests check that all of them resolve
EXPORTS = [
    "smvp_last_error", "smvp_version_string", "smvp_set_option", "smvp_get_option",
    "smvp_mm_read_banner", "smvp_mm_read_mtx_crd_size", "smvp_mm_read_coo_entries",
    "smvp_mm_read_header_path", "smvp_mm_read_coo_path", "smvp_mm_expanded_count", "smvp_mm_expand_symmetric",
    "smvp_cache_write_csr", "smvp_cache_read_header", "smvp_cache_read_csr", "smvp_coo_from_csr",
    "smvp_csr_from_coo", "smvp_tjds_from_coo", "smvp_csr_from_coo_device", "smvp_tjds_from_coo_device",
    "smvp_device_count", "smvp_device_info", "smvp_csr_plan_info", "smvp_tjds_plan_info",
    "smvp_csr_create", "smvp_csr_create_block", "smvp_csr_far_share", "smvp_csr_set_kernel", "smvp_csr_get_kernel", "smvp_csr_gather_spread", "smvp_csr_spmv",
    "smvp_csr_describe", "smvp_csr_plan_launches", "smvp_csr_destroy",
    "smvp_tjds_create", "smvp_tjds_set_x", "smvp_tjds_zero_y", "smvp_tjds_spmv",
    "smvp_tjds_set_ref_quirks", "smvp_tjds_set_mode", "smvp_tjds_set_tile", "smvp_tjds_set_value_cache", "smvp_tjds_get_value_cache", "smvp_tjds_describe", "smvp_tjds_destroy",
    "smvp_shard_opts_default", "smvp_csr_sharded_create", "smvp_csr_sharded_create_ex", "smvp_tjds_sharded_create",
    "smvp_tjds_sharded_create_ex", "smvp_sharded_layout", "smvp_sharded_set_csr_kernel", "smvp_sharded_probe_exchange", "smvp_sharded_exchange_info", "smvp_sharded_set_exchange", "smvp_sharded_set_x", "smvp_sharded_spmv",
    "smvp_sharded_synchronize", "smvp_sharded_feed_back", "smvp_sharded_get_y", "smvp_sharded_info", "smvp_sharded_destroy",
    "smvp_run_opts_default", "smvp_csr_compute", "smvp_tjds_compute", "smvp_last_run_info",
    "smvp_time_stats", "smvp_generate_report_text", "smvp_cisr_coegen", "smvp_cisr_coegen_path",
    "smvp_synth_row_lengths", "smvp_synth_fill", "smvp_partition_rows", "smvp_vector_random",
]


def lib():
    """The loaded library.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C smvp-toolkit_amd` first" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.smvp_last_error.restype = C.c_char_p
        L.smvp_version_string.restype = C.c_char_p
        L.smvp_set_option.argtypes = [C.c_char_p, C.c_int]
        L.smvp_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        vp, ci = C.c_void_p, C.c_int
        L.smvp_csr_create.argtypes = [C.POINTER(vp), ci, ci, ci, ci, vp, vp, vp, ci, vp]
        L.smvp_csr_create_block.argtypes = [C.POINTER(vp), ci, ci, ci, ci, vp, vp, vp, ci, vp, C.c_longlong]
        L.smvp_csr_far_share.argtypes = [vp, C.POINTER(C.c_double)]
        L.smvp_csr_set_kernel.argtypes = [vp, ci, ci]
        L.smvp_csr_get_kernel.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
        L.smvp_csr_gather_spread.argtypes = [vp, C.POINTER(C.c_double)]
        L.smvp_csr_spmv.argtypes = [vp, vp, vp, vp]
        L.smvp_csr_describe.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_double)]
        L.smvp_csr_plan_launches.argtypes = [vp, C.POINTER(ci)]
        L.smvp_csr_plan_info.argtypes = [vp, C.POINTER(PlanInfo)]
        L.smvp_tjds_plan_info.argtypes = [vp, C.POINTER(PlanInfo)]
        L.smvp_csr_destroy.argtypes = [vp]
        L.smvp_csr_destroy.restype = None
        L.smvp_tjds_create.argtypes = [C.POINTER(vp), ci, ci, ci, ci, ci, vp, vp, vp, vp, ci]
        L.smvp_tjds_set_x.argtypes = [vp, vp, vp]
        L.smvp_tjds_zero_y.argtypes = [vp, vp, vp]
        L.smvp_tjds_spmv.argtypes = [vp, vp, vp]
        L.smvp_tjds_set_ref_quirks.argtypes = [vp, ci, ci, ci]
        L.smvp_tjds_set_mode.argtypes = [vp, ci]
        L.smvp_tjds_set_tile.argtypes = [vp, ci]
        L.smvp_tjds_set_value_cache.argtypes = [vp, ci]
        L.smvp_tjds_get_value_cache.argtypes = [vp, C.POINTER(ci), C.POINTER(C.c_longlong)]
        L.smvp_last_run_info.argtypes = [C.POINTER(RunInfo)]
        L.smvp_tjds_describe.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_double)]
        L.smvp_tjds_destroy.argtypes = [vp]
        L.smvp_tjds_destroy.restype = None
        L.smvp_csr_from_coo.argtypes = [vp, ci, ci, vp, vp, vp]
        L.smvp_tjds_from_coo.argtypes = [vp, ci, ci, ci, vp, vp, ci, vp, vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
        L.smvp_csr_sharded_create.argtypes = [C.POINTER(vp), ci, vp, ci, ci, ci, vp, vp, vp]
        L.smvp_tjds_sharded_create.argtypes = [C.POINTER(vp), ci, vp, vp, ci, ci, ci]
        L.smvp_csr_sharded_create_ex.argtypes = [C.POINTER(vp), ci, vp, ci, ci, ci, vp, vp, vp, C.POINTER(ShardOpts)]
        L.smvp_tjds_sharded_create_ex.argtypes = [C.POINTER(vp), ci, vp, vp, ci, ci, ci, C.POINTER(ShardOpts)]
        L.smvp_shard_opts_default.argtypes = [C.POINTER(ShardOpts)]
        L.smvp_shard_opts_default.restype = None
        L.smvp_sharded_layout.argtypes = [vp, C.POINTER(ci), vp, vp]
        L.smvp_sharded_set_csr_kernel.argtypes = [vp, ci, ci]
        L.smvp_sharded_probe_exchange.argtypes = [vp, ci]
        L.smvp_sharded_exchange_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), vp, C.POINTER(ci)]
        L.smvp_sharded_set_exchange.argtypes = [vp, ci]
        L.smvp_sharded_set_x.argtypes = [vp, vp]
        L.smvp_sharded_spmv.argtypes = [vp, ci, ci]
        L.smvp_sharded_synchronize.argtypes = [vp, C.POINTER(C.c_double)]
        L.smvp_sharded_get_y.argtypes = [vp, ci, ci, vp]
        L.smvp_sharded_feed_back.argtypes = [vp, ci]
        L.smvp_sharded_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
        L.smvp_sharded_destroy.argtypes = [vp]
        L.smvp_sharded_destroy.restype = None
        L.smvp_csr_from_coo_device.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp]
        L.smvp_tjds_from_coo_device.argtypes = [vp, ci, ci, ci, vp, vp, ci, vp, vp, C.POINTER(ci), C.POINTER(ci),
                                                C.POINTER(ci), vp]
        L.smvp_csr_compute.argtypes = [vp, ci, ci, ci, ci, C.POINTER(RunOpts), vp, vp, C.POINTER(TimeStats)]
        L.smvp_tjds_compute.argtypes = [vp, ci, ci, ci, ci, C.POINTER(RunOpts), vp, vp, C.POINTER(TimeStats)]
        L.smvp_run_opts_default.argtypes = [C.POINTER(RunOpts)]
        L.smvp_run_opts_default.restype = None
        L.smvp_time_stats.argtypes = [vp, ci, C.POINTER(TimeStats)]
        L.smvp_generate_report_text.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, ci, ci, ci, vp,
                                                C.POINTER(TimeStats), C.c_ulong, C.c_char_p, C.c_size_t]
        i64, u64 = C.c_int64, C.c_uint64
        L.smvp_synth_row_lengths.argtypes = [ci, u64, i64, i64, ci, i64, i64, vp]
        L.smvp_synth_fill.argtypes = [ci, u64, i64, i64, ci, i64, i64, vp, vp, vp, ci]
        L.smvp_partition_rows.argtypes = [vp, ci, ci, vp]
        L.smvp_vector_random.argtypes = [vp, C.c_int64, C.c_uint64]
        L.smvp_mm_read_header_path.argtypes = [C.c_char_p, vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
        L.smvp_mm_read_coo_path.argtypes = [C.c_char_p, vp, ci, vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
        L.smvp_mm_expanded_count.argtypes = [vp, vp, ci, C.POINTER(ci)]
        L.smvp_mm_expand_symmetric.argtypes = [vp, vp, ci, ci, ci, vp, ci, C.POINTER(ci)]
        L.smvp_cache_write_csr.argtypes = [C.c_char_p, C.c_char_p, vp, ci, ci, ci, ci, vp, vp, vp]
        L.smvp_cache_read_header.argtypes = [C.c_char_p, C.c_char_p, vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
        L.smvp_cache_read_csr.argtypes = [C.c_char_p, ci, ci, vp, vp, vp]
        L.smvp_coo_from_csr.argtypes = [ci, vp, vp, vp, vp]
        L.smvp_cisr_coegen_path.argtypes = [vp, ci, ci, ci, C.c_char_p]
        L.smvp_device_count.argtypes = [C.POINTER(ci)]
        L.smvp_device_info.argtypes = [ci, C.c_char_p, C.c_size_t, C.POINTER(ci), C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def _check(code, where):
    if code != OK:
        raise SmvpError(code, where)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a if a.size else np.zeros(1, dtype=dtype)


# ---------------------------------------------------------------- Matrix Market
def mm_read_header(path):
    """-> (status, typecode str, rows, cols, nnz); status is an mmio code, not raised."""
    tc = C.create_string_buffer(4)
    m, n, nz = C.c_int(), C.c_int(), C.c_int()
    rc = lib().smvp_mm_read_header_path(os.fsencode(path), C.cast(tc, C.c_void_p), C.byref(m), C.byref(n), C.byref(nz))
    return rc, tc.raw.decode(), m.value, n.value, nz.value


def mm_read_coo(path):
    """-> (typecode, rows, cols, coo structured array).  Raises SmvpError on a bad file."""
    rc, tc, m, n, nz = mm_read_header(path)
    _check(rc, "smvp_mm_read_header_path")
    coo = np.zeros(max(nz, 1), dtype=COO_DTYPE)
    tcb = C.create_string_buffer(4)
    mm, nn, nzz = C.c_int(), C.c_int(), C.c_int()
    _check(lib().smvp_mm_read_coo_path(os.fsencode(path), _p(coo), nz, C.cast(tcb, C.c_void_p), C.byref(mm),
                                       C.byref(nn), C.byref(nzz)), "smvp_mm_read_coo_path")
    return tcb.raw.decode(), mm.value, nn.value, coo[:nz]


def mm_expand_symmetric(typecode, coo, rows, cols):
    """Mirror the off-diagonal entries of symmetric / skew-symmetric storage (NOT the reference's behaviour) -> coo."""
    nnz = len(coo)
    coo = _arr(coo, COO_DTYPE)
    tc = C.create_string_buffer(typecode.encode()[:4].ljust(4), 4)
    want = C.c_int()
    _check(lib().smvp_mm_expanded_count(C.cast(tc, C.c_void_p), _p(coo), nnz, C.byref(want)), "smvp_mm_expanded_count")
    out = np.zeros(max(want.value, 1), dtype=COO_DTYPE)
    got = C.c_int()
    _check(lib().smvp_mm_expand_symmetric(C.cast(tc, C.c_void_p), _p(coo), nnz, rows, cols, _p(out), want.value,
                                          C.byref(got)), "smvp_mm_expand_symmetric")
    return out[:got.value]


def cache_write_csr(cache_path, mtx_path, typecode, rows, cols, row_ptr, col_ind, val, expanded=False):
    rp, ci, v = _arr(row_ptr, np.int32), _arr(col_ind, np.int32), _arr(val, np.float64)
    tc = C.create_string_buffer(typecode.encode()[:4].ljust(4), 4)
    _check(lib().smvp_cache_write_csr(os.fsencode(cache_path), os.fsencode(mtx_path), C.cast(tc, C.c_void_p), int(expanded),
                                      rows, cols, int(rp[rows]), _p(rp), _p(ci), _p(v)), "smvp_cache_write_csr")


def cache_read_csr(cache_path, mtx_path=None):
    """-> (typecode, expanded, rows, cols, row_ptr, col_ind, val); raises SmvpError (ERR_IO: no cache, ERR_INVALID: stale)."""
    tc = C.create_string_buffer(4)
    fl, m, n, nz = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _check(lib().smvp_cache_read_header(os.fsencode(cache_path), None if mtx_path is None else os.fsencode(mtx_path),
                                        C.cast(tc, C.c_void_p), C.byref(fl), C.byref(m), C.byref(n), C.byref(nz)),
           "smvp_cache_read_header")
    rp = np.zeros(m.value + 1, dtype=np.int32)
    ci = np.zeros(max(nz.value, 1), dtype=np.int32)
    v = np.zeros(max(nz.value, 1), dtype=np.float64)
    _check(lib().smvp_cache_read_csr(os.fsencode(cache_path), m.value, nz.value, _p(rp), _p(ci), _p(v)), "smvp_cache_read_csr")
    return tc.raw.decode(), bool(fl.value & 1), m.value, n.value, rp, ci[:nz.value], v[:nz.value]


def coo_from_csr(rows, row_ptr, col_ind, val):
    rp, ci, v = _arr(row_ptr, np.int32), _arr(col_ind, np.int32), _arr(val, np.float64)
    out = np.zeros(max(int(rp[rows]), 1), dtype=COO_DTYPE)
    _check(lib().smvp_coo_from_csr(rows, _p(rp), _p(ci), _p(v), _p(out)), "smvp_coo_from_csr")
    return out[:int(rp[rows])]


def cisr_coegen(coo, rows, slots, path):
    """CISR .coe file of the matrix (smvp_cisr_coegen_path); raises SmvpError (ERR_UNSUPPORTED: the reference's overrun exit)."""
    n = len(coo)
    _check(lib().smvp_cisr_coegen_path(_p(_arr(coo, COO_DTYPE)), rows, n, slots, os.fsencode(path)), "smvp_cisr_coegen_path")
    return open(path).read()


def make_coo(rows_idx, cols_idx, vals):
    coo = np.zeros(len(rows_idx), dtype=COO_DTYPE)
    coo["row"], coo["col"], coo["val"] = rows_idx, cols_idx, vals
    return coo


# ------------------------------------------------------------------ conversion
def csr_from_coo(coo, rows):
    coo = np.ascontiguousarray(coo, dtype=COO_DTYPE)
    nnz = len(coo)
    row_ptr = np.zeros(rows + 1, dtype=np.int32)
    col_ind = np.zeros(max(nnz, 1), dtype=np.int32)
    val = np.zeros(max(nnz, 1), dtype=np.float64)
    _check(lib().smvp_csr_from_coo(_p(_arr(coo, COO_DTYPE)), rows, nnz, _p(row_ptr), _p(col_ind), _p(val)),
           "smvp_csr_from_coo")
    return row_ptr, col_ind[:nnz], val[:nnz]


class TjdsArrays:
    """perm / start_pos / row_ind / val plus the two reference-quirk scalars."""


def tjds_from_coo(coo, rows, cols):
    coo = np.ascontiguousarray(coo, dtype=COO_DTYPE)
    nnz = len(coo)
    t = TjdsArrays()
    t.rows, t.cols, t.nnz = rows, cols, nnz
    perm = np.zeros(max(cols, 1), dtype=np.int32)
    cap = max(rows, nnz) + 2
    sp = np.zeros(cap, dtype=np.int32)
    row_ind = np.zeros(max(nnz, 1), dtype=np.int32)
    val = np.zeros(max(nnz, 1), dtype=np.float64)
    nd, rn, ls = C.c_int(), C.c_int(), C.c_int()
    _check(lib().smvp_tjds_from_coo(_p(_arr(coo, COO_DTYPE)), rows, cols, nnz, _p(perm), _p(sp), cap, _p(row_ind),
                                    _p(val), C.byref(nd), C.byref(rn), C.byref(ls)), "smvp_tjds_from_coo")
    t.num_diag, t.ref_num_tjdiag, t.last_diag_single = nd.value, rn.value, ls.value
    t.perm, t.start_pos = perm[:cols], sp[:t.num_diag + 1].copy()
    t.row_ind, t.val = row_ind[:nnz], val[:nnz]
    return t


def csr_from_coo_device(d_coo, rows, cols, nnz, stream=None):
    """COO -> CSR on the GPU.  d_coo: torch uint8 CUDA tensor holding nnz smvp_coo_t (16 B each).
    Returns (row_ptr, col_ind, val) as torch CUDA tensors."""
    import torch

    row_ptr = torch.empty(rows + 1, dtype=torch.int32, device=d_coo.device)
    col_ind = torch.empty(max(nnz, 1), dtype=torch.int32, device=d_coo.device)
    val = torch.empty(max(nnz, 1), dtype=torch.float64, device=d_coo.device)
    _check(lib().smvp_csr_from_coo_device(_dev_ptr(d_coo), rows, cols, nnz, _dev_ptr(row_ptr), _dev_ptr(col_ind),
                                          _dev_ptr(val), _stream_ptr(stream)), "smvp_csr_from_coo_device")
    return row_ptr, col_ind[:nnz], val[:nnz]


def tjds_from_coo_device(d_coo, rows, cols, nnz, stream=None):
    """COO -> TJDS on the GPU -> TjdsArrays whose arrays are torch CUDA tensors."""
    import torch

    t = TjdsArrays()
    t.rows, t.cols, t.nnz = rows, cols, nnz
    cap = max(rows, nnz) + 2
    perm = torch.empty(max(cols, 1), dtype=torch.int32, device=d_coo.device)
    sp = torch.empty(cap, dtype=torch.int32, device=d_coo.device)
    row_ind = torch.empty(max(nnz, 1), dtype=torch.int32, device=d_coo.device)
    val = torch.empty(max(nnz, 1), dtype=torch.float64, device=d_coo.device)
    nd, rn, ls = C.c_int(), C.c_int(), C.c_int()
    _check(lib().smvp_tjds_from_coo_device(_dev_ptr(d_coo), rows, cols, nnz, _dev_ptr(perm), _dev_ptr(sp), cap,
                                           _dev_ptr(row_ind), _dev_ptr(val), C.byref(nd), C.byref(rn), C.byref(ls),
                                           _stream_ptr(stream)), "smvp_tjds_from_coo_device")
    t.num_diag, t.ref_num_tjdiag, t.last_diag_single = nd.value, rn.value, ls.value
    t.perm, t.start_pos = perm[:cols], sp[:t.num_diag + 1]
    t.row_ind, t.val = row_ind[:nnz], val[:nnz]
    return t


def partition_rows(row_ptr, parts):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    bounds = np.zeros(parts + 1, dtype=np.int32)
    _check(lib().smvp_partition_rows(_p(row_ptr), len(row_ptr) - 1, parts, _p(bounds)), "smvp_partition_rows")
    return bounds


def vector_random(n, seed=67890):
    """The operand of `--x random`: uniform [0, 1), a pure function of (seed, index)."""
    x = np.zeros(max(n, 1), dtype=np.float64)
    _check(lib().smvp_vector_random(_p(x), n, seed), "smvp_vector_random")
    return x[:n]


# ------------------------------------------------------------------- synthetic
def synth_csr(kind, seed, rows_total, cols_total, param=0, row_begin=0, row_end=None, threads=None):
    """Row block [row_begin, row_end) of a synthetic matrix -> (row_ptr, col_ind, val), local row numbering."""
    row_end = rows_total if row_end is None else row_end
    n = row_end - row_begin
    lens = np.zeros(max(n, 1), dtype=np.int32)
    _check(lib().smvp_synth_row_lengths(kind, seed, rows_total, cols_total, param, row_begin, row_end, _p(lens)),
           "smvp_synth_row_lengths")
    row_ptr64 = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens[:n], out=row_ptr64[1:])
    if row_ptr64[-1] >= 2 ** 31:
        raise ValueError("row block holds %d entries: 32-bit indices (the reference's) cannot address it" % row_ptr64[-1])
    row_ptr = row_ptr64.astype(np.int32)
    nnz = int(row_ptr[-1])
    col_ind = np.empty(max(nnz, 1), dtype=np.int32)
    val = np.empty(max(nnz, 1), dtype=np.float64)
    if threads is None:
        threads = max(1, min(16, (os.cpu_count() or 1)))
    _check(lib().smvp_synth_fill(kind, seed, rows_total, cols_total, param, row_begin, row_end, _p(row_ptr),
                                 _p(col_ind), _p(val), threads), "smvp_synth_fill")
    return row_ptr, col_ind[:nnz], val[:nnz]


# ---------------------------------------------------------------------- device
def device_count():
    n = C.c_int()
    _check(lib().smvp_device_count(C.byref(n)), "smvp_device_count")
    return n.value


def device_info(device=0):
    name = C.create_string_buffer(256)
    cus, mem = C.c_int(), C.c_size_t()
    _check(lib().smvp_device_info(device, name, 256, C.byref(cus), C.byref(mem)), "smvp_device_info")
    return name.value.decode(), cus.value, mem.value


def _dev_ptr(t):
    """Device address of a torch tensor (or a raw int address)."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


def _stream_ptr(stream):
    if stream is None:
        return None
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)   # torch.cuda.Stream


class CsrMatrix:
    """Device-resident CSR matrix (smvp_csr_t).  Arrays may be numpy (copied to HBM) or torch CUDA tensors (adopted)."""

    def __init__(self, rows, cols, row_ptr, col_ind, val, device=0, first_row=0):
        """first_row: global number of the first row when this is a row block of a larger matrix (smvp_csr_create_block)."""
        self.rows, self.cols = rows, cols
        self._h = C.c_void_p()
        self._keep = None
        if isinstance(row_ptr, np.ndarray):
            rp, ci, v = _arr(row_ptr, np.int32), _arr(col_ind, np.int32), _arr(val, np.float64)
            self.nnz = int(rp[rows]) if rows >= 0 else 0
            _check(lib().smvp_csr_create_block(C.byref(self._h), device, rows, cols, self.nnz, _p(rp), _p(ci), _p(v),
                                               MEM_HOST, None, int(first_row)), "smvp_csr_create_block")
        else:
            host_rp = _arr(row_ptr.cpu().numpy(), np.int32)
            self.nnz = int(host_rp[rows])
            self._keep = (row_ptr, col_ind, val)
            _check(lib().smvp_csr_create_block(C.byref(self._h), device, rows, cols, self.nnz, _dev_ptr(row_ptr),
                                               _dev_ptr(col_ind), _dev_ptr(val), MEM_DEVICE, _p(host_rp), int(first_row)),
                   "smvp_csr_create_block")

    def set_kernel(self, kernel, param=0):
        _check(lib().smvp_csr_set_kernel(self._h, kernel, param), "smvp_csr_set_kernel")

    def get_kernel(self):
        k, p = C.c_int(), C.c_int()
        _check(lib().smvp_csr_get_kernel(self._h, C.byref(k), C.byref(p)), "smvp_csr_get_kernel")
        return k.value, p.value

    def launches(self):
        """Kernel launches per product of the current plan."""
        n = C.c_int()
        _check(lib().smvp_csr_plan_launches(self._h, C.byref(n)), "smvp_csr_plan_launches")
        return n.value

    def plan_info(self):
        """{matrix_bytes, plan_bytes, build_ms} of the current launch plan."""
        i = PlanInfo()
        _check(lib().smvp_csr_plan_info(self._h, C.byref(i)), "smvp_csr_plan_info")
        return {"matrix_bytes": i.matrix_bytes, "plan_bytes": i.plan_bytes, "build_ms": i.build_ms}

    def far_share(self):
        """Share of the entries further than 4096 from the diagonal of the whole matrix (AUTO's choice of the binned plan); -1: not measured."""
        v = C.c_double()
        _check(lib().smvp_csr_far_share(self._h, C.byref(v)), "smvp_csr_far_share")
        return v.value

    def gather_spread(self):
        """Share of the gathers that pull their own line of x (what AUTO's choice of the column sweep rests on); -1: not sampled."""
        v = C.c_double()
        _check(lib().smvp_csr_gather_spread(self._h, C.byref(v)), "smvp_csr_gather_spread")
        return v.value

    def spmv(self, x, y, stream=None):
        """y = A x, asynchronous on `stream`; x, y are torch CUDA float64 tensors."""
        _check(lib().smvp_csr_spmv(self._h, _dev_ptr(x), _dev_ptr(y), _stream_ptr(stream)), "smvp_csr_spmv")

    def describe(self):
        name = C.create_string_buffer(256)
        b = C.c_double()
        _check(lib().smvp_csr_describe(self._h, name, 256, C.byref(b)), "smvp_csr_describe")
        return name.value.decode(), b.value

    def close(self):
        if self._h:
            lib().smvp_csr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TjdsMatrix:
    """Device-resident TJDS matrix (smvp_tjds_t) built from TjdsArrays."""

    def __init__(self, t, device=0):
        self.rows, self.cols, self.nnz = t.rows, t.cols, t.nnz
        self._t = t
        self._h = C.c_void_p()
        if isinstance(t.perm, np.ndarray):
            _check(lib().smvp_tjds_create(C.byref(self._h), device, t.rows, t.cols, t.nnz, t.num_diag,
                                          _p(_arr(t.perm, np.int32)), _p(_arr(t.start_pos, np.int32)),
                                          _p(_arr(t.row_ind, np.int32)), _p(_arr(t.val, np.float64)), MEM_HOST),
                   "smvp_tjds_create")
        else:       # torch CUDA tensors (smvp_tjds_from_coo_device): adopted in place
            _check(lib().smvp_tjds_create(C.byref(self._h), device, t.rows, t.cols, t.nnz, t.num_diag,
                                          _dev_ptr(t.perm), _dev_ptr(t.start_pos), _dev_ptr(t.row_ind),
                                          _dev_ptr(t.val), MEM_DEVICE), "smvp_tjds_create")

    def set_x(self, x, stream=None):
        _check(lib().smvp_tjds_set_x(self._h, _dev_ptr(x), _stream_ptr(stream)), "smvp_tjds_set_x")

    def zero_y(self, y, stream=None):
        _check(lib().smvp_tjds_zero_y(self._h, _dev_ptr(y), _stream_ptr(stream)), "smvp_tjds_zero_y")

    def spmv(self, y, stream=None):
        _check(lib().smvp_tjds_spmv(self._h, _dev_ptr(y), _stream_ptr(stream)), "smvp_tjds_spmv")

    def set_mode(self, mode):
        _check(lib().smvp_tjds_set_mode(self._h, mode), "smvp_tjds_set_mode")

    def set_tile(self, entries_per_tile):
        _check(lib().smvp_tjds_set_tile(self._h, entries_per_tile), "smvp_tjds_set_tile")

    def set_value_cache(self, min_tiles):
        """Keep a second copy of the values of val lines shared by `min_tiles` tiles or more (0 = none)."""
        _check(lib().smvp_tjds_set_value_cache(self._h, int(min_tiles)), "smvp_tjds_set_value_cache")

    def get_value_cache(self):
        """(min_tiles, cached entries)."""
        m, n = C.c_int(), C.c_longlong()
        _check(lib().smvp_tjds_get_value_cache(self._h, C.byref(m), C.byref(n)), "smvp_tjds_get_value_cache")
        return m.value, n.value

    def plan_info(self):
        """{matrix_bytes, plan_bytes, build_ms}: the plans of the modes selected so far."""
        i = PlanInfo()
        _check(lib().smvp_tjds_plan_info(self._h, C.byref(i)), "smvp_tjds_plan_info")
        return {"matrix_bytes": i.matrix_bytes, "plan_bytes": i.plan_bytes, "build_ms": i.build_ms}

    def set_ref_quirks(self, enable=True):
        _check(lib().smvp_tjds_set_ref_quirks(self._h, int(enable), self._t.ref_num_tjdiag,
                                              self._t.last_diag_single), "smvp_tjds_set_ref_quirks")

    def describe(self):
        name = C.create_string_buffer(128)
        b = C.c_double()
        _check(lib().smvp_tjds_describe(self._h, name, 128, C.byref(b)), "smvp_tjds_describe")
        return name.value.decode(), b.value

    def close(self):
        if self._h:
            lib().smvp_tjds_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedMatrix:
    """Row blocks of one matrix on several GPUs of this process (smvp_sharded_t), RCCL all-gather of y."""

    def __init__(self, fmt, ngpus, rows, cols, coo=None, csr=None, devices=None, chunks=0, balance=True, exchange=EXCHANGE_AUTO):
        self._h = C.c_void_p()
        devs = None if devices is None else (C.c_int * ngpus)(*devices)
        o = ShardOpts()
        lib().smvp_shard_opts_default(C.byref(o))
        o.chunks, o.balance, o.exchange = int(chunks), int(balance), int(exchange)
        if fmt == "csr":
            rp, ci, v = csr
            nnz = int(rp[rows])
            _check(lib().smvp_csr_sharded_create_ex(C.byref(self._h), ngpus, devs, rows, cols, nnz, _p(_arr(rp, np.int32)),
                                                    _p(_arr(ci, np.int32)), _p(_arr(v, np.float64)), C.byref(o)),
                   "smvp_csr_sharded_create_ex")
        else:
            nnz = len(coo)
            _check(lib().smvp_tjds_sharded_create_ex(C.byref(self._h), ngpus, devs, _p(_arr(coo, COO_DTYPE)), rows, cols,
                                                     nnz, C.byref(o)), "smvp_tjds_sharded_create_ex")
        self.rows, self.cols = rows, cols

    def layout(self):
        """(chunks per GPU, block bounds[ngpus + 1], chunk bounds[ngpus][chunks + 1]) in global rows."""
        n, _ = self.info()
        c = C.c_int()
        _check(lib().smvp_sharded_layout(self._h, C.byref(c), None, None), "smvp_sharded_layout")
        bounds = np.zeros(n + 1, dtype=np.int32)
        cb = np.zeros(n * (c.value + 1), dtype=np.int32)
        _check(lib().smvp_sharded_layout(self._h, C.byref(c), _p(bounds), _p(cb)), "smvp_sharded_layout")
        return c.value, bounds, cb.reshape(n, c.value + 1)

    def set_csr_kernel(self, kernel, param=0):
        _check(lib().smvp_sharded_set_csr_kernel(self._h, kernel, param), "smvp_sharded_set_csr_kernel")

    def probe_exchange(self, reps=3):
        """Time one product's exchange of y under every available form (AUTO handles then keep the fastest)."""
        _check(lib().smvp_sharded_probe_exchange(self._h, int(reps)), "smvp_sharded_probe_exchange")
        return self.exchange_info()

    def exchange_info(self):
        """{"active": EXCHANGE_*, "available": [EXCHANGE_*...], "ms": {name: ms of the last probe}, "rccl_ranks": n}."""
        act, av, rk = C.c_int(), C.c_int(), C.c_int()
        ms = (C.c_double * 3)()
        _check(lib().smvp_sharded_exchange_info(self._h, C.byref(act), C.byref(av), ms, C.byref(rk)), "smvp_sharded_exchange_info")
        return {"active": act.value, "active_name": EXCHANGE_NAMES[act.value],
                "available": [e for e in range(3) if av.value >> e & 1],
                "ms": {EXCHANGE_NAMES[e]: ms[e] for e in range(3) if ms[e] >= 0}, "rccl_ranks": rk.value}

    def set_exchange(self, exchange):
        _check(lib().smvp_sharded_set_exchange(self._h, int(exchange)), "smvp_sharded_set_exchange")

    def set_x(self, x=None):
        keep = None if x is None else _arr(x, np.float64)
        _check(lib().smvp_sharded_set_x(self._h, None if keep is None else _p(keep)), "smvp_sharded_set_x")

    def spmv(self, allgather=GATHER_OVERLAPPED, timed=True):
        """allgather: GATHER_NONE / GATHER_OVERLAPPED (True) / GATHER_AFTER."""
        _check(lib().smvp_sharded_spmv(self._h, int(allgather), int(timed)), "smvp_sharded_spmv")

    def feed_back(self, normalize=False):
        _check(lib().smvp_sharded_feed_back(self._h, int(normalize)), "smvp_sharded_feed_back")

    def synchronize(self):
        ms = C.c_double()
        _check(lib().smvp_sharded_synchronize(self._h, C.byref(ms)), "smvp_sharded_synchronize")
        return ms.value

    def get_y(self, slot=0, gathered=True):
        y = np.zeros(max(self.rows, 1), dtype=np.float64)
        _check(lib().smvp_sharded_get_y(self._h, slot, int(gathered), _p(y)), "smvp_sharded_get_y")
        return y[:self.rows]

    def info(self):
        n, b = C.c_int(), C.c_int()
        _check(lib().smvp_sharded_info(self._h, C.byref(n), C.byref(b)), "smvp_sharded_info")
        return n.value, b.value

    def close(self):
        if self._h:
            lib().smvp_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------- reference-shaped entry points
def _run_opts(device, csr_kernel, csr_param, ref_quirks, x, device_convert=False, ngpus=0, iterate=False,
              normalize=False, tjds_mode=TJDS_MODE_AUTO, timing=TIMING_AUTO, exchange=EXCHANGE_AUTO, repeat_patience_us=0):
    o = RunOpts()
    lib().smvp_run_opts_default(C.byref(o))
    o.device, o.csr_kernel, o.csr_param, o.tjds_ref_quirks = device, csr_kernel, csr_param, int(ref_quirks)
    o.convert_on_device, o.ngpus = int(device_convert), int(ngpus)
    o.iterate, o.normalize = int(iterate), int(normalize)
    o.tjds_mode, o.timing, o.shard_exchange = int(tjds_mode), int(timing), int(exchange)
    o.repeat_patience_us = int(repeat_patience_us)
    keep = None
    if x is not None:
        keep = _arr(x, np.float64)
        o.x = keep.ctypes.data
    return o, keep


def last_run_info():
    """How the last *_compute on this thread timed its products -> RunInfo (timing, graph_replays, wall_ms, ...)."""
    r = RunInfo()
    _check(lib().smvp_last_run_info(C.byref(r)), "smvp_last_run_info")
    return r


def csr_compute(coo, rows, cols, iters=1, device=0, kernel=CSR_KERNEL_AUTO, param=0, x=None, device_convert=False,
                ngpus=0, iterate=False, normalize=False, timing=TIMING_AUTO, exchange=EXCHANGE_AUTO, repeat_patience_us=0):
    """smvp_csr_compute: COO in, (y, per-iteration ms, TimeStats) out."""
    nnz = len(coo)
    coo = _arr(coo, COO_DTYPE)
    y = np.zeros(max(rows, 1), dtype=np.float64)
    ms = np.zeros(iters, dtype=np.float64)
    st = TimeStats()
    o, keep = _run_opts(device, kernel, param, False, x, device_convert, ngpus, iterate, normalize, timing=timing, exchange=exchange,
                        repeat_patience_us=repeat_patience_us)
    _check(lib().smvp_csr_compute(_p(coo), rows, cols, nnz, iters, C.byref(o), _p(y), _p(ms), C.byref(st)),
           "smvp_csr_compute")
    return y[:rows], ms, st


def tjds_compute(coo, rows, cols, iters=1, device=0, ref_quirks=False, x=None, device_convert=False, ngpus=0,
                 iterate=False, normalize=False, mode=TJDS_MODE_AUTO, timing=TIMING_AUTO, exchange=EXCHANGE_AUTO, repeat_patience_us=0):
    """smvp_tjds_compute: COO in, (y, per-iteration ms, TimeStats) out."""
    nnz = len(coo)
    coo = _arr(coo, COO_DTYPE)
    y = np.zeros(max(rows, 1), dtype=np.float64)
    ms = np.zeros(iters, dtype=np.float64)
    st = TimeStats()
    o, keep = _run_opts(device, CSR_KERNEL_AUTO, 0, ref_quirks, x, device_convert, ngpus, iterate, normalize,
                        tjds_mode=mode, timing=timing, exchange=exchange, repeat_patience_us=repeat_patience_us)
    _check(lib().smvp_tjds_compute(_p(coo), rows, cols, nnz, iters, C.byref(o), _p(y), _p(ms), C.byref(st)),
           "smvp_tjds_compute")
    return y[:rows], ms, st


# -------------------------------------------------------------- stats + report
def time_stats(ms):
    ms = _arr(ms, np.float64)
    st = TimeStats()
    _check(lib().smvp_time_stats(_p(ms), len(ms), C.byref(st)), "smvp_time_stats")
    return st


def generate_report_text(input_name, report_dir, alg_name, nnz, y, iters, stats, unix_time=0):
    rows = len(y)
    y = _arr(y, np.float64)
    out = C.create_string_buffer(4096)
    _check(lib().smvp_generate_report_text(os.fsencode(input_name), os.fsencode(report_dir or ""),
                                           alg_name.encode(), nnz, rows, iters, _p(y), C.byref(stats),
                                           unix_time, out, 4096), "smvp_generate_report_text")
    return out.value.decode()
