"""Pin the CPU oracle against everything the reference commits for this path.

The reference holds no unit tests; its de-facto golden vectors are the eleven
report files under output-test/ and build/ (copied to tests/golden/reports/),
each with the full y vector printed with "%g" (main-cli.c:308).  goodwin.mtx is
missing from the reference checkout, so its two reports cannot be checked.
"""
import os
import re

import numpy as np
import pytest

import oracle_binding as ob
from conftest import REPORTS, SAMPLES

# recorded by the survey from the reference's own SMVP_TJDS_DEBUG dump of ibm32
IBM32_PERM = [8, 9, 0, 2, 1, 6, 7, 10, 17, 24, 28, 3, 4, 22, 25, 26, 29, 5, 11, 14, 15, 16, 18, 19, 20, 23,
              30, 31, 12, 13, 21, 27]
IBM32_START_POS = [0, 32, 64, 92, 109, 120, 124, 126]
# diagonal lengths quoted in SURVEY.md 8(a) row a11
MEMPLUS_DIAG_HEAD = [17758, 17758, 17706, 12670, 5347, 2889, 2750]
PWT_DIAGS = [36519, 36328, 31010, 30588, 29152, 11232, 5770, 619, 94, 1]
# (D, length of original column 0) from SURVEY.md 8(a) row a9
DIAG_COUNTS = {"ibm32.mtx": (7, 6), "curtis54.mtx": (16, 3), "memplus.mtx": (574, 9), "pwt.mtx": (10, 9),
               "pdp08-pg4.mtx": (3, 3)}


def load(name):
    rc, tc, m, n, coo = ob.mm_read_coo(ob.fixture_path(name))
    assert rc == 0
    return tc, m, n, coo


@pytest.mark.parametrize("name", SAMPLES)
def test_csr_y_matches_committed_report(name):
    tc, m, n, coo = load(name)
    row_ptr, col_ind, val = ob.csr_build(coo, m)
    y = ob.csr_spmv(row_ptr, col_ind, val, np.ones(n))
    ref = ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_%s.txt" % REPORTS[name][0]))
    assert ob.fmt_g(y) == ref


@pytest.mark.parametrize("name", [s for s in SAMPLES if REPORTS[s][1]])
def test_tjds_refquirks_y_matches_committed_report(name):
    """The reference's TJDS output is wrong on three of four matrices; the quirk model reproduces all four."""
    tc, m, n, coo = load(name)
    t = ob.tjds_build(coo, m, n)
    y = ob.tjds_spmv(t, np.ones(n), refquirks=True)
    ref = ob.report_y_lines(ob.read_report("smvp-toolbox_report_TJDS_%s.txt" % REPORTS[name][1]))
    assert ob.fmt_g(y) == ref


@pytest.mark.parametrize("name", SAMPLES)
def test_literal_row_ptr_equals_prefix_sum(name):
    """main-cli.c:348-365 as written (on zeroed memory) gives the standard row_ptr on every sample."""
    tc, m, n, coo = load(name)
    a = ob.csr_build(coo, m)
    b = ob.csr_build(coo, m, literal=True)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("name", SAMPLES)
def test_tjds_structure(name):
    tc, m, n, coo = load(name)
    t = ob.tjds_build(coo, m, n)
    assert (t.num_diag, t.ref_num_tjdiag) == DIAG_COUNTS[name]
    assert sorted(t.perm.tolist()) == list(range(n))
    lens = np.diff(t.start_pos)
    assert t.start_pos[0] == 0 and t.start_pos[-1] == len(coo)
    assert np.all(lens[:-1] >= lens[1:]) and lens[0] <= n
    # a correct TJDS product equals the CSR product
    row_ptr, col_ind, val = ob.csr_build(coo, m)
    x = np.random.default_rng(7).random(n)
    y_csr = ob.csr_spmv(row_ptr, col_ind, val, x)
    y_tj = ob.tjds_spmv(t, x)
    scale = ob.csr_spmv(row_ptr, col_ind, np.abs(val), np.abs(x))
    assert np.all(np.abs(y_csr - y_tj) <= 1e-13 * scale)


def test_ibm32_integer_arrays_from_reference_dump():
    tc, m, n, coo = load("ibm32.mtx")
    t = ob.tjds_build(coo, m, n)
    assert t.perm.tolist() == IBM32_PERM
    assert t.start_pos.tolist() == IBM32_START_POS


def test_diagonal_lengths_quoted_in_survey():
    tc, m, n, coo = load("memplus.mtx")
    t = ob.tjds_build(coo, m, n)
    assert np.diff(t.start_pos)[:7].tolist() == MEMPLUS_DIAG_HEAD and np.diff(t.start_pos)[-3:].tolist() == [1, 1, 1]
    tc, m, n, coo = load("pwt.mtx")
    t = ob.tjds_build(coo, m, n)
    assert np.diff(t.start_pos).tolist() == PWT_DIAGS
    assert t.last_diag_single == 1


def test_symmetric_storage_is_not_mirrored():
    """pwt is 'pattern symmetric'; the reference multiplies the stored triangle only (main-cli.c:1427-1441)."""
    tc, m, n, coo = load("pwt.mtx")
    assert tc == "MCPS" and len(coo) == 181313
    assert np.all(coo["row"] >= coo["col"]) and np.all(coo["val"] == 1.0)


def _mask(text):
    """Drop what legitimately differs between two runs: timestamp and the five timing lines."""
    text = re.sub(r"Generated on \d+", "Generated on T", text)
    return re.sub(r"(Total|Average|Fastest|Slowest) Time: \S+ ms|Time StDev: \S+ ms", "TIME", text)


@pytest.mark.parametrize("name", SAMPLES)
def test_report_writer_bytes(name, tmp_path):
    tc, m, n, coo = load(name)
    row_ptr, col_ind, val = ob.csr_build(coo, m)
    y = ob.csr_spmv(row_ptr, col_ind, val, np.ones(n))
    ref = ob.read_report("smvp-toolbox_report_CSR_%s.txt" % REPORTS[name][0])
    input_name = ref.split("\n")[4]
    iters = int(re.search(r"Compute times for (\d+) iterations", ref).group(1))
    st = ob.time_stats(np.full(iters, 0.5))
    out = str(tmp_path / "r.txt")
    assert ob.write_report(out, "CSR", 123, input_name, len(coo), y, iters, st) == 0
    assert _mask(open(out).read()) == _mask(ref)


def test_time_stats_definition():
    ms = np.array([1.0, 2.0, 4.0, 5.0])
    st = ob.time_stats(ms)
    assert (st.total, st.avg, st.min, st.max) == (12.0, 3.0, 1.0, 5.0)
    assert st.stdev == pytest.approx(np.std(ms))


def test_bad_file_codes(tmp_path):
    assert ob.mm_read_header(ob.fixture_path("badfile.mtx"))[0] == 12          # MM_PREMATURE_EOF
    p = tmp_path / "x.mtx"
    p.write_text("%%NotMatrixMarket matrix coordinate real general\n1 1 1\n1 1 1\n")
    assert ob.mm_read_header(str(p))[0] == 14                                   # MM_NO_HEADER
    p.write_text("%%MatrixMarket tensor coordinate real general\n1 1 1\n1 1 1\n")
    assert ob.mm_read_header(str(p))[0] == 15                                   # MM_UNSUPPORTED_TYPE
