"""ctypes binding for oracle/liboracle.so -- the CPU checker.

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (smvp-toolkit_amd/) never touches it.
"""
import ctypes as C
import gzip
import os
import shutil
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")


class Coo(C.Structure):
    _fields_ = [("row", C.c_int), ("col", C.c_int), ("val", C.c_double)]


COO_DTYPE = np.dtype([("row", "<i4"), ("col", "<i4"), ("val", "<f8")], align=True)
assert COO_DTYPE.itemsize == C.sizeof(Coo) == 16


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("total", "avg", "stdev", "min", "max")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        _lib = C.CDLL(so)
        _lib.orc_write_report.argtypes = [C.c_char_p, C.c_char_p, C.c_ulong, C.c_char_p, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p, C.POINTER(Stats)]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fixture_path(name):
    """Path of a sample matrix; .gz fixtures are inflated into a temp dir once."""
    plain = os.path.join(GOLDEN, "sample-data", name)
    if os.path.exists(plain):
        return plain
    gz = plain + ".gz"
    if not os.path.exists(gz):
        raise FileNotFoundError(name)
    cache = os.path.join(tempfile.gettempdir(), "smvp_fixture_cache_%d" % os.getuid())
    os.makedirs(cache, exist_ok=True)
    out = os.path.join(cache, name)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(gz):
        # several ranks may get here together: each inflates into its own temp file, the rename is atomic
        fd, tmp = tempfile.mkstemp(prefix=name + ".", dir=cache)
        with gzip.open(gz, "rb") as src, os.fdopen(fd, "wb") as dst:
            shutil.copyfileobj(src, dst)
        os.replace(tmp, out)
    return out


def read_report(name):
    """Text of a committed reference report (plain or .gz)."""
    p = os.path.join(GOLDEN, "reports", name)
    if os.path.exists(p):
        return open(p).read()
    return gzip.open(p + ".gz", "rt").read()


def report_y_lines(text):
    lines = text.split("\n")
    a = lines.index("[")
    b = lines.index("]")
    return lines[a + 1:b]


def mm_read_header(path):
    tc = C.create_string_buffer(4)
    m, n, nz = C.c_int(), C.c_int(), C.c_int()
    rc = lib().orc_mm_read_header(path.encode(), tc, C.byref(m), C.byref(n), C.byref(nz))
    return rc, tc.raw.decode(), m.value, n.value, nz.value


def mm_read_coo(path):
    rc, tc, m, n, nz = mm_read_header(path)
    if rc != 0:
        return rc, tc, m, n, None
    coo = np.zeros(max(nz, 1), dtype=COO_DTYPE)
    tcb = C.create_string_buffer(4)
    mm, nn, nzz = C.c_int(), C.c_int(), C.c_int()
    rc = lib().orc_mm_read_coo(path.encode(), _p(coo), nz, tcb, C.byref(mm), C.byref(nn), C.byref(nzz))
    return rc, tcb.raw.decode(), mm.value, nn.value, coo[:nz]


def csr_build(coo, rows, literal=False):
    nnz = len(coo)
    coo = np.ascontiguousarray(coo)
    row_ptr = np.zeros(rows + 1, dtype=np.int32)
    col_ind = np.zeros(max(nnz, 1), dtype=np.int32)
    val = np.zeros(max(nnz, 1), dtype=np.float64)
    fn = lib().orc_csr_build_literal if literal else lib().orc_csr_build
    fn(_p(coo), rows, nnz, _p(row_ptr), _p(col_ind), _p(val))
    return row_ptr, col_ind[:nnz], val[:nnz]


def csr_spmv(row_ptr, col_ind, val, x):
    rows = len(row_ptr) - 1
    y = np.zeros(rows, dtype=np.float64)
    ci = np.ascontiguousarray(col_ind, dtype=np.int32)
    v = np.ascontiguousarray(val, dtype=np.float64)
    if len(ci) == 0:
        ci = np.zeros(1, np.int32)
        v = np.zeros(1, np.float64)
    lib().orc_csr_spmv(rows, _p(np.ascontiguousarray(row_ptr, dtype=np.int32)), _p(ci), _p(v),
                       _p(np.ascontiguousarray(x, dtype=np.float64)), _p(y))
    return y


def csr_iterate(row_ptr, col_ind, val, x0, iters, normalize=False):
    rows = len(row_ptr) - 1
    y = np.zeros(rows, dtype=np.float64)
    lib().orc_csr_iterate(rows, _p(np.ascontiguousarray(row_ptr, dtype=np.int32)),
                          _p(np.ascontiguousarray(col_ind, dtype=np.int32)), _p(np.ascontiguousarray(val, dtype=np.float64)),
                          _p(np.ascontiguousarray(x0, dtype=np.float64)), iters, int(normalize), _p(y))
    return y


class Tjds:
    pass


def tjds_build(coo, rows, cols):
    nnz = len(coo)
    coo = np.ascontiguousarray(coo)
    t = Tjds()
    t.rows, t.cols, t.nnz = rows, cols, nnz
    t.perm = np.zeros(max(cols, 1), dtype=np.int32)
    sp = np.zeros(max(rows, nnz) + 2, dtype=np.int32)
    t.row_ind = np.zeros(max(nnz, 1), dtype=np.int32)
    t.val = np.zeros(max(nnz, 1), dtype=np.float64)
    nd, rn, ls = C.c_int(), C.c_int(), C.c_int()
    rc = lib().orc_tjds_build(_p(coo), rows, cols, nnz, _p(t.perm), _p(sp), _p(t.row_ind), _p(t.val),
                              C.byref(nd), C.byref(rn), C.byref(ls))
    assert rc == 0
    t.num_diag, t.ref_num_tjdiag, t.last_diag_single = nd.value, rn.value, ls.value
    t.start_pos = sp[:t.num_diag + 1].copy()
    t.perm = t.perm[:cols]
    t.row_ind = t.row_ind[:nnz]
    t.val = t.val[:nnz]
    return t


def _pad(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a if len(a) else np.zeros(1, dtype)


def tjds_spmv(t, x, refquirks=False):
    y = np.zeros(t.rows, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    args = [_p(_pad(t.perm, np.int32)), _p(_pad(t.start_pos, np.int32)), _p(_pad(t.row_ind, np.int32)),
            _p(_pad(t.val, np.float64)), _p(x), _p(y)]
    if refquirks:
        lib().orc_tjds_spmv_refquirks(t.rows, t.cols, t.num_diag, t.ref_num_tjdiag, t.last_diag_single, *args)
    else:
        lib().orc_tjds_spmv(t.rows, t.cols, t.num_diag, *args)
    return y


def time_stats(ms):
    ms = np.ascontiguousarray(ms, dtype=np.float64)
    st = Stats()
    lib().orc_time_stats(_p(ms), len(ms), C.byref(st))
    return st


def write_report(path, alg, unix_time, input_name, nnz, y, iters, st):
    y = np.ascontiguousarray(y, dtype=np.float64)
    return lib().orc_write_report(path.encode(), alg.encode(), unix_time, input_name.encode(), nnz, len(y),
                                  iters, _p(y), C.byref(st))


def csr_timed(row_ptr, col_ind, val, x, iters):
    rows = len(row_ptr) - 1
    y = np.zeros(rows, dtype=np.float64)
    ms = np.zeros(iters, dtype=np.float64)
    lib().orc_csr_timed(rows, _p(row_ptr), _p(col_ind), _p(val), _p(x), _p(y), iters, _p(ms))
    return y, ms


def tjds_timed(t, x, iters):
    y = np.zeros(t.rows, dtype=np.float64)
    ms = np.zeros(iters, dtype=np.float64)
    lib().orc_tjds_timed(t.rows, t.cols, t.num_diag, _p(t.perm), _p(t.start_pos), _p(t.row_ind), _p(t.val),
                         _p(np.ascontiguousarray(x, dtype=np.float64)), _p(y), iters, _p(ms))
    return y, ms


def cisr_coegen(coo, rows, slots, path):
    """Oracle restatement of main-cli.c:473-729 -> (0 written | 1 the reference's overrun exit, text)."""
    coo = np.ascontiguousarray(coo)
    lib().orc_cisr_coegen_path.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    rc = lib().orc_cisr_coegen_path(_p(coo if len(coo) else np.zeros(1, dtype=COO_DTYPE)), rows, len(coo), slots, path.encode())
    return rc, open(path).read()


def fmt_g(y):
    """'%g' of every element, through C printf (what main-cli.c:308 prints)."""
    libc = C.CDLL(None)
    buf = C.create_string_buffer(64)
    out = []
    for v in np.asarray(y, dtype=np.float64):
        libc.snprintf(buf, 64, b"%g", C.c_double(float(v)))
        out.append(buf.value.decode())
    return out
