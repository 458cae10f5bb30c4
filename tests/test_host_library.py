"""CPU-side checks of the product library: no compute entry point is called here.

Covers the host half of the path (Matrix Market reader, COO->CSR, COO->TJDS,
stats, report writer, row partition) against the oracle and the committed
reports, and that the C-ABI library exports every symbol include/smvp_amd.h
declares.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import smvp_toolkit_amd as sm
from conftest import REPORTS, ROOT, SAMPLES
from test_oracle_golden import _mask


# ----------------------------------------------------------------------- ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "smvp_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(smvp_[a-z0-9_]+)\s*\(", header))
    assert declared == set(sm.EXPORTS), declared ^ set(sm.EXPORTS)
    lib = sm.lib()
    for name in sorted(declared):
        assert getattr(lib, name) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", sm.LIB_PATH], text=True)
    exported = set(re.findall(r" T (smvp_[a-z0-9_]+)", out))
    assert declared <= exported


def test_no_torch_or_oracle_in_the_abi_library():
    out = subprocess.check_output(["ldd", sm.LIB_PATH], text=True)
    assert "torch" not in out and "oracle" not in out
    assert "amdhip64" in out


def test_struct_layouts_match_header():
    assert sm.COO_DTYPE.itemsize == 16 and sm.COO_DTYPE.fields["val"][1] == 8
    assert C.sizeof(sm.TimeStats) == 40


def test_compute_fails_loudly_without_a_device():
    """On a box without a GPU the entry points must refuse, not fall back."""
    if sm.device_count() > 0:
        pytest.skip("a GPU is visible")
    coo = sm.make_coo([0], [0], [1.0])
    with pytest.raises(sm.SmvpError) as e:
        sm.csr_compute(coo, 1, 1)
    assert e.value.code == sm.ERR_NO_DEVICE
    with pytest.raises(sm.SmvpError) as e:
        sm.tjds_compute(coo, 1, 1)
    assert e.value.code == sm.ERR_NO_DEVICE
    with pytest.raises(sm.SmvpError):
        sm.CsrMatrix(1, 1, np.array([0, 1], np.int32), np.array([0], np.int32), np.array([1.0]))


# -------------------------------------------------------------- Matrix Market
@pytest.mark.parametrize("name", SAMPLES)
def test_reader_matches_oracle(name):
    path = ob.fixture_path(name)
    tc, m, n, coo = sm.mm_read_coo(path)
    rc, tc2, m2, n2, coo2 = ob.mm_read_coo(path)
    assert (tc, m, n) == (tc2, m2, n2)
    assert coo.tobytes() == coo2.tobytes()


BANNERS = [
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n1 1 3.5\n", 0, "MCRG"),
    ("%%MatrixMarket MATRIX Coordinate PATTERN Symmetric\n2 2 1\n2 1\n", 0, "MCPS"),
    ("%%MatrixMarket matrix coordinate integer skew-symmetric\n% c\n%c2\n3 3 1\n2 1 4\n", 0, "MCIK"),
    ("%%MatrixMarket matrix coordinate complex hermitian\n2 2 1\n2 1 1 0\n", 0, "MCCH"),
    ("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n", 0, "MARG"),
    ("%%MatrixMarket matrix coordinate real general\n% only comments\n\n   \n4 5 0\n", 0, "MCRG"),
    ("%%matrixmarket matrix coordinate real general\n1 1 0\n", 14, None),
    ("%MatrixMarket matrix coordinate real general\n1 1 0\n", 14, None),
    ("%%MatrixMarket matrix coordinate real\n1 1 0\n", 12, None),
    ("%%MatrixMarket vector coordinate real general\n1 1 0\n", 15, None),
    ("%%MatrixMarket matrix sparse real general\n1 1 0\n", 15, None),
    ("%%MatrixMarket matrix coordinate double general\n1 1 0\n", 15, None),
    ("%%MatrixMarket matrix coordinate real diagonal\n1 1 0\n", 15, None),
    ("", 12, None),
    ("%%MatrixMarket matrix coordinate real general\n% comments then nothing\n", 12, None),
]


def _ref_mmio():
    so = os.path.join(ROOT, "oracle", "_ref", "libmmio_ref.so")
    return C.CDLL(so) if os.path.exists(so) else None


@pytest.mark.parametrize("text,code,tc", BANNERS)
def test_banner_and_size_codes(tmp_path, text, code, tc):
    p = tmp_path / "m.mtx"
    p.write_text(text)
    got = sm.mm_read_header(str(p))
    assert got[0] == code
    if code == 0:
        assert got[1] == tc
    assert ob.mm_read_header(str(p))[0] == code
    if code == 0:
        assert ob.mm_read_header(str(p))[1:] == got[1:]
    ref = _ref_mmio()
    if ref is not None:                      # the reference's own mmio.c, compiled by oracle/Makefile
        libc = C.CDLL(None)
        libc.fopen.restype = C.c_void_p
        libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        libc.fclose.argtypes = [C.c_void_p]
        ref.mm_read_banner.argtypes = [C.c_void_p, C.c_char_p]
        ref.mm_read_mtx_crd_size.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        f = libc.fopen(str(p).encode(), b"r")
        buf = C.create_string_buffer(4)
        rc = ref.mm_read_banner(f, buf)
        m, n, nz = C.c_int(), C.c_int(), C.c_int()
        if rc == 0 and buf.raw[1:2] == b"C":
            rc = ref.mm_read_mtx_crd_size(f, C.byref(m), C.byref(n), C.byref(nz))
        libc.fclose(f)
        if tc is None or tc[1] == "C":
            assert rc == code
        if code == 0 and tc[1] == "C":
            assert (buf.raw.decode(), m.value, n.value, nz.value) == got[1:]


@pytest.mark.parametrize("name", SAMPLES + ["badfile.mtx"])
def test_reference_mmio_agrees_on_samples(name):
    ref = _ref_mmio()
    if ref is None:
        pytest.skip("oracle/_ref/libmmio_ref.so not built (reference checkout absent)")
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    ref.mm_read_banner.argtypes = [C.c_void_p, C.c_char_p]
    ref.mm_read_mtx_crd_size.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
    path = ob.fixture_path(name)
    f = libc.fopen(path.encode(), b"r")
    buf = C.create_string_buffer(4)
    rc = ref.mm_read_banner(f, buf)
    m, n, nz = C.c_int(), C.c_int(), C.c_int()
    if rc == 0:
        rc = ref.mm_read_mtx_crd_size(f, C.byref(m), C.byref(n), C.byref(nz))
    libc.fclose(f)
    got = sm.mm_read_header(path)
    assert got[0] == rc
    if rc == 0:
        assert got[1:] == (buf.raw.decode(), m.value, n.value, nz.value)


def test_entries_may_span_lines_like_fscanf(tmp_path):
    p = tmp_path / "m.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1\n 2.5 2\n2 -1e3\n3 3 7\n")
    tc, m, n, coo = sm.mm_read_coo(str(p))
    assert coo["row"].tolist() == [0, 1, 2] and coo["col"].tolist() == [0, 1, 2]
    assert coo["val"].tolist() == [2.5, -1000.0, 7.0]


def test_truncated_entries_are_an_error(tmp_path):
    p = tmp_path / "m.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1 1.0\n2 2\n")
    with pytest.raises(sm.SmvpError) as e:
        sm.mm_read_coo(str(p))
    assert e.value.code == sm.MM_PREMATURE_EOF


# ------------------------------------------------------------------ converters
@pytest.mark.parametrize("name", SAMPLES)
def test_csr_arrays_bit_exact(name):
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path(name))
    got = sm.csr_from_coo(coo, m)
    want = ob.csr_build(coo, m)
    for g, w in zip(got, want):
        assert g.dtype == w.dtype and g.tobytes() == w.tobytes()


@pytest.mark.parametrize("name", SAMPLES)
def test_tjds_arrays_bit_exact(name):
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path(name))
    got = sm.tjds_from_coo(coo, m, n)
    want = ob.tjds_build(coo, m, n)
    assert (got.num_diag, got.ref_num_tjdiag, got.last_diag_single) == \
        (want.num_diag, want.ref_num_tjdiag, want.last_diag_single)
    for f in ("perm", "start_pos", "row_ind", "val"):
        assert getattr(got, f).tobytes() == getattr(want, f).tobytes(), f


def random_coo(rng, rows, cols, nnz, empty_rows=False):
    """Distinct (row, col) pairs in shuffled order; optionally leave every third row empty."""
    nnz = min(nnz, rows * cols)
    flat = rng.choice(rows * cols, size=nnz, replace=False)
    r, c = flat // cols, flat % cols
    if empty_rows:
        keep = (r % 3) != 1
        r, c = r[keep], c[keep]
    return sm.make_coo(r, c, rng.uniform(-1, 1, len(r)))


@pytest.mark.parametrize("seed", range(12))
def test_converters_on_random_matrices(seed):
    rng = np.random.default_rng(seed)
    rows, cols = int(rng.integers(1, 60)), int(rng.integers(1, 60))
    coo = random_coo(rng, rows, cols, int(rng.integers(0, rows * cols + 1)), empty_rows=bool(seed % 2))
    got = sm.csr_from_coo(coo, rows)
    want = ob.csr_build(coo, rows)
    for g, w in zip(got, want):
        assert g.tobytes() == w.tobytes()
    t, u = sm.tjds_from_coo(coo, rows, cols), ob.tjds_build(coo, rows, cols)
    assert (t.num_diag, t.ref_num_tjdiag, t.last_diag_single) == (u.num_diag, u.ref_num_tjdiag, u.last_diag_single)
    for f in ("perm", "start_pos", "row_ind", "val"):
        assert getattr(t, f).tobytes() == getattr(u, f).tobytes(), f


def test_converters_reject_out_of_range_entries():
    with pytest.raises(sm.SmvpError):
        sm.csr_from_coo(sm.make_coo([5], [0], [1.0]), 3)
    with pytest.raises(sm.SmvpError):
        sm.tjds_from_coo(sm.make_coo([0], [9], [1.0]), 3, 3)


def test_empty_matrix_conversions():
    coo = sm.make_coo([], [], [])
    row_ptr, col_ind, val = sm.csr_from_coo(coo, 4)
    assert row_ptr.tolist() == [0] * 5 and len(col_ind) == 0
    t = sm.tjds_from_coo(coo, 4, 3)
    assert t.num_diag == 0 and t.start_pos.tolist() == [0] and t.perm.tolist() == [0, 1, 2]


def test_input_is_not_modified():
    """The reference sorts the caller's COO array in place (main-cli.c:340,766); the ABI must not."""
    rng = np.random.default_rng(3)
    coo = random_coo(rng, 20, 20, 100)
    before = coo.tobytes()
    sm.csr_from_coo(coo, 20)
    sm.tjds_from_coo(coo, 20, 20)
    assert coo.tobytes() == before


# ------------------------------------------------------------- stats + report
def test_time_stats_match_oracle():
    ms = np.random.default_rng(0).random(1000)
    a, b = sm.time_stats(ms), ob.time_stats(ms)
    assert (a.time_total, a.time_avg, a.time_min, a.time_max, a.time_stdev) == (b.total, b.avg, b.min, b.max, b.stdev)


@pytest.mark.parametrize("name", SAMPLES)
@pytest.mark.parametrize("alg", ["CSR", "TJDS"])
def test_report_text_matches_committed_report(name, alg, tmp_path):
    stamp = REPORTS[name][0 if alg == "CSR" else 1]
    if stamp is None:
        pytest.skip("the reference crashed before writing this report")
    ref = ob.read_report("smvp-toolbox_report_%s_%s.txt" % (alg, stamp))
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path(name))
    # y comes from the oracle here (host-only test); the GPU tests repeat this with the kernels' y
    if alg == "CSR":
        y = ob.csr_spmv(*ob.csr_build(coo, m), np.ones(n))
    else:
        y = ob.tjds_spmv(ob.tjds_build(coo, m, n), np.ones(n), refquirks=True)
    iters = int(re.search(r"Compute times for (\d+) iterations", ref).group(1))
    path = sm.generate_report_text(ref.split("\n")[4], str(tmp_path), alg, len(coo), y, iters,
                                   sm.time_stats(np.full(iters, 0.25)), unix_time=1615284655)
    assert os.path.basename(path) == "smvp-toolbox_report_%s_1615284655.txt" % alg
    assert _mask(open(path).read()) == _mask(ref)


def test_report_dir_handling(tmp_path):
    st = sm.time_stats(np.ones(2))
    p1 = sm.generate_report_text("a.mtx", str(tmp_path) + "/", "CSR", 1, np.ones(1), 2, st, unix_time=5)
    p2 = sm.generate_report_text("a.mtx", str(tmp_path), "CSR", 1, np.ones(1), 2, st, unix_time=5)
    assert p1 == p2 == str(tmp_path / "smvp-toolbox_report_CSR_5.txt")
    assert open(p1).read().count("Execution results") == 2       # opened "a+" like main-cli.c:293
    with pytest.raises(sm.SmvpError):
        sm.generate_report_text("a.mtx", str(tmp_path / "missing"), "CSR", 1, np.ones(1), 2, st, unix_time=5)


# ------------------------------------------------------------------- partition
def test_partition_rows_balances_bytes():
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path("memplus.mtx"))
    row_ptr, _, _ = sm.csr_from_coo(coo, m)
    for parts in (1, 2, 4, 8):
        b = sm.partition_rows(row_ptr, parts)
        assert b[0] == 0 and b[-1] == m and np.all(np.diff(b) >= 0)
        cost = 12.0 * np.diff(row_ptr[b]) + 20.0 * np.diff(b)
        assert cost.max() <= cost.sum() / parts + 12.0 * 574 + 20.0


# ------------------------------------------------- parallel Matrix Market tokeniser
def _write_mtx(path, rows, cols, r, c, v, pattern=False, sep="\n"):
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s general\n%% generated\n%d %d %d\n" %
                ("pattern" if pattern else "real", rows, cols, len(r)))
        if pattern:
            f.write(sep.join("%d %d" % (a + 1, b + 1) for a, b in zip(r, c)))
        else:
            f.write(sep.join("%d %d %.17g" % (a + 1, b + 1, x) for a, b, x in zip(r, c, v)))
        f.write("\n")


@pytest.mark.parametrize("threads", ["1", "2", "7", "16"])
@pytest.mark.parametrize("pattern", [False, True])
def test_parallel_tokeniser_equals_serial(tmp_path, monkeypatch, threads, pattern):
    rng = np.random.default_rng(42)
    n = 20000
    r, c = rng.integers(0, 5000, n), rng.integers(0, 7000, n)
    v = rng.uniform(-1e3, 1e3, n) * 10.0 ** rng.integers(-20, 20, n)
    p = str(tmp_path / "m.mtx")
    _write_mtx(p, 5000, 7000, r, c, v, pattern=pattern, sep="\n" if not pattern else "  \t\n ")
    sm.set_option("mm_threads", int(threads))        # (conftest.py resets the plan options after every test)
    tc, m, k, coo = sm.mm_read_coo(p)
    assert (m, k, len(coo)) == (5000, 7000, n)
    assert np.array_equal(coo["row"], r) and np.array_equal(coo["col"], c)
    assert np.array_equal(coo["val"], np.ones(n) if pattern else v)


def test_parallel_tokeniser_number_spellings(tmp_path, monkeypatch):
    """The parallel tokeniser parses plain decimal tokens itself where the result is exact by construction and asks strtol /
    strtod for everything else: every spelling %lg accepts must give the double strtod gives (Python's float is correctly
    rounded like glibc's strtod)."""
    rng = np.random.default_rng(7)
    toks = ["0", "-0", "+3", "5.", ".5", "-.25", "1e22", "1e23", "1E-22", "1e-23", "9007199254740992", "9007199254740993",
            "0.1", "0.30000000000000004", "123456789012345678", "1.7976931348623157e308", "4.9e-324", "2.2250738585072014e-308",
            "00012.500", "1e+05", "1e0005", "12345678901234567890", "0.000000000000000000001234", "1e400", "-1e-400",
            "inf", "-Infinity", "0x1p3", "0x.8p1", "1.5e3", "3.0E+2", "7e-1", "100000000000000000000000", "0.0", "000"]
    for fmt in ("%.3g", "%.6g", "%.12g", "%.15g", "%.16g", "%.17g", "%e", "%f", "%.0f"):
        for _ in range(400):
            toks.append(fmt % (rng.uniform(-1, 1) * 10.0 ** int(rng.integers(-30, 30))))
    toks += [str(int(v)) for v in rng.integers(-10**9, 10**9, 300)]
    n = len(toks)
    r, c = rng.integers(1, 1000, n), rng.integers(1, 1000, n)
    p = tmp_path / "spell.mtx"
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write("1000 1000 %d\n" % n)
        for a, b, t in zip(r, c, toks):
            f.write("%s%d \t+%d %s\r\n" % ("000" if a % 3 == 0 else "", a, b, t))
    want = np.array([float.fromhex(t) if "0x" in t else float(t) for t in toks])
    for threads in ("1", "5"):
        sm.set_option("mm_threads", int(threads))
        tc, m, k, coo = sm.mm_read_coo(str(p))
        assert np.array_equal(coo["row"], r - 1) and np.array_equal(coo["col"], c - 1)
        got = coo["val"]
        assert got.tobytes() == want.tobytes(), [(t, g, w) for t, g, w in zip(toks, got, want) if not (g == w or (g != g and w != w))][:5]


def test_parallel_tokeniser_on_sample_and_fallbacks(tmp_path, monkeypatch):
    sm.set_option("mm_threads", 8)
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path("memplus.mtx"))
    rc, tc2, m2, n2, coo2 = ob.mm_read_coo(ob.fixture_path("memplus.mtx"))
    assert coo.tobytes() == coo2.tobytes()
    # a token the strict tokeniser refuses ("2.5" where an index belongs) goes through the fscanf-like serial path
    p = tmp_path / "odd.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 4.0\n2 2.5 7\n")
    tc, m, n, coo = sm.mm_read_coo(str(p))
    assert coo["row"].tolist() == [0, 1] and coo["col"].tolist() == [0, 1] and coo["val"].tolist() == [4.0, 0.5]
    # a short file still reports the premature end
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1 4.0\n2 2 7\n")
    with pytest.raises(sm.SmvpError) as e:
        sm.mm_read_coo(str(p))
    assert e.value.code == sm.MM_PREMATURE_EOF


def test_round3_entry_points_reject_bad_arguments_without_a_device():
    """The plan controls added in round 3 validate their arguments before anything touches a GPU."""
    import ctypes as C

    L = sm.lib()
    v, n, c = C.c_double(), C.c_int(), C.c_longlong()
    assert L.smvp_csr_gather_spread(None, C.byref(v)) == sm.ERR_INVALID
    assert L.smvp_csr_plan_launches(None, C.byref(n)) == sm.ERR_INVALID
    assert L.smvp_tjds_set_value_cache(None, 4) == sm.ERR_INVALID
    assert L.smvp_tjds_get_value_cache(None, C.byref(n), C.byref(c)) == sm.ERR_INVALID
    assert L.smvp_vector_random(None, 5, 1) == sm.ERR_INVALID
    assert L.smvp_vector_random(None, 0, 1) == sm.OK                       # nothing to write
    assert "bad argument" in L.smvp_last_error().decode() or L.smvp_last_error().decode() == ""
    x = sm.vector_random(7, 3)
    assert np.array_equal(x, sm.vector_random(9, 3)[:7]) and not np.array_equal(x, sm.vector_random(7, 4))
