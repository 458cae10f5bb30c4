"""GPU parity: the HIP kernels, called through the C ABI, against the CPU oracle.

Tolerances (BASELINE.json north_star):
  - integer arrays (row_ptr/col_ind, perm/start_pos/row_ind): bit-exact -- checked on the host
    in test_host_library.py; here the same arrays feed the kernels.
  - fp64 y: |y_gpu[r] - y_ref[r]| <= 1e-9 * sum_j |a_rj x_j| for every row (row-normwise relative;
    memplus has ~200 rows whose true sum cancels to ~1e-15 of their terms, so an element-wise
    relative bound cannot hold for ANY summation order other than the serial one).
  - pattern matrices (ibm32, curtis54, pwt) and pdp08-pg4 have small-integer sums: exact.
  - rows the stream kernel sums with one lane are bit-identical to the serial oracle.
"""
import os
import re
import subprocess
import zlib

import numpy as np
import pytest

import oracle_binding as ob
import smvp_toolkit_amd as sm
from conftest import REPORTS, SAMPLES
from test_oracle_golden import _mask

pytestmark = pytest.mark.gpu
TOL = 1e-9
EXACT = {"ibm32.mtx", "curtis54.mtx", "pwt.mtx", "pdp08-pg4.mtx"}

CSR_VARIANTS = [(sm.CSR_KERNEL_STREAM, 256), (sm.CSR_KERNEL_STREAM, 1024), (sm.CSR_KERNEL_STREAM, 2048),
                (sm.CSR_KERNEL_STREAM_CARRY, 1024), (sm.CSR_KERNEL_STREAM_CARRY, 2048)] + \
               [(sm.CSR_KERNEL_VECTOR, t) for t in (2, 4, 8, 16, 32, 64)] + \
               [(sm.CSR_KERNEL_COLSWEEP, rb) for rb in (0, 1024, 8192)] + \
               [(sm.CSR_KERNEL_BINNED, band) for band in (0, 1, 3, 100)]   # band: |column - row| beyond it is "far"


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def row_scale(row_ptr, col_ind, val, x):
    """sum_j |a_rj x_j| per row -- the yardstick of the normwise bound."""
    return ob.csr_spmv(row_ptr, col_ind, np.abs(val), np.abs(x))


def assert_close(y, ref, scale, exact=False):
    assert y.shape == ref.shape
    if exact:
        assert np.array_equal(y, ref)
    else:
        bad = np.abs(y - ref) > TOL * scale
        assert not bad.any(), "%d rows beyond %g * sum|a x|; worst %g" % (
            bad.sum(), TOL, (np.abs(y - ref) / np.maximum(scale, 1e-300)).max())


def gpu_csr(torch, rows, cols, row_ptr, col_ind, val, x, kernel, param):
    A = sm.CsrMatrix(rows, cols, row_ptr, col_ind, val)
    A.set_kernel(kernel, param)
    assert A.get_kernel()[0] == kernel and (param == 0 or A.get_kernel()[1] == param)
    dx = dev(torch, x)
    dy = torch.full((max(rows, 1),), float("nan"), dtype=torch.float64, device="cuda")
    A.spmv(dx, dy)
    torch.cuda.synchronize()
    A.close()
    return dy.cpu().numpy()[:rows]


def load(name):
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path(name))
    return m, n, coo


# --------------------------------------------------------------- sample matrices
@pytest.mark.parametrize("name", SAMPLES)
@pytest.mark.parametrize("kernel,param", CSR_VARIANTS)
def test_csr_sample_matrices_ones(torch, name, kernel, param):
    m, n, coo = load(name)
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    x = np.ones(n)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    y = gpu_csr(torch, m, n, row_ptr, col_ind, val, x, kernel, param)
    assert_close(y, ref, row_scale(row_ptr, col_ind, val, x), exact=name in EXACT)
    if name in EXACT or kernel in (sm.CSR_KERNEL_STREAM, sm.CSR_KERNEL_STREAM_CARRY):
        # the %g text of the report must equal the reference's committed report
        want = ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_%s.txt" % REPORTS[name][0]))
        got = ob.fmt_g(y)
        if name in EXACT:
            assert got == want
        else:
            # memplus: the "%g" text of every row the kernel sums in the serial order (one lane, up to 32 entries) and
            # of every well-conditioned longer row must be the committed report's; only rows whose sum cancels to
            # less than 1e-6 of its terms may print differently after a re-ordered sum
            lens = np.diff(row_ptr)
            scale = row_scale(row_ptr, col_ind, val, x)
            serial = lens <= (32 if kernel == sm.CSR_KERNEL_STREAM else 0)
            if kernel == sm.CSR_KERNEL_STREAM_CARRY:      # rows inside one tile are summed by one lane there too
                serial = (lens <= 32) & (row_ptr[:-1] // param == (np.maximum(row_ptr[1:], 1) - 1) // param)
            well = np.abs(ref) > 1e-6 * scale
            assert all(got[i] == want[i] for i in np.flatnonzero(serial))
            assert sum(got[i] != want[i] for i in np.flatnonzero(well & ~serial)) <= 2    # a last-digit rounding tie at most


@pytest.mark.parametrize("name", SAMPLES)
@pytest.mark.parametrize("kernel,param", CSR_VARIANTS)
def test_csr_sample_matrices_general_x(torch, name, kernel, param):
    m, n, coo = load(name)
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    x = np.random.default_rng(67890).random(n)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    y = gpu_csr(torch, m, n, row_ptr, col_ind, val, x, kernel, param)
    assert_close(y, ref, row_scale(row_ptr, col_ind, val, x), exact=kernel == sm.CSR_KERNEL_COLSWEEP)   # sweep: serial order


def test_stream_kernel_is_bitwise_serial_on_short_rows(torch):
    """Every row of up to 32 entries is summed left to right by one lane: bit for bit the serial loop."""
    m, n, coo = load("memplus.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    x = np.random.default_rng(1).random(n)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    lens = np.diff(row_ptr)
    y = gpu_csr(torch, m, n, row_ptr, col_ind, val, x, sm.CSR_KERNEL_STREAM, 256)
    assert np.array_equal(y[lens <= 32], ref[lens <= 32])
    for tile in (1024, 2048):
        # owner form: wherever the row lies, also across a tile edge
        y = gpu_csr(torch, m, n, row_ptr, col_ind, val, x, sm.CSR_KERNEL_STREAM, tile)
        short = lens <= 32
        assert short.mean() > 0.98
        assert np.array_equal(y[short], ref[short])
        # carry form: rows that fit one tile
        y = gpu_csr(torch, m, n, row_ptr, col_ind, val, x, sm.CSR_KERNEL_STREAM_CARRY, tile)
        one_tile = short & (row_ptr[:-1] // tile == (np.maximum(row_ptr[1:], 1) - 1) // tile)
        assert one_tile.mean() > 0.9
        assert np.array_equal(y[one_tile], ref[one_tile])


def test_csr_16_bit_column_offsets(torch):
    """A tile whose columns span less than 65536 reads 16-bit offsets from its smallest column (csr_stream_owner<., 5, .>:
    10 instead of 12 bytes per entry); a wider tile of the same matrix reads col_ind itself; a matrix in which most tiles
    are wide keeps the plain kernel (<., 0, .>).  Same bits every way, and as the serial loop on rows of up to 32 entries."""
    rng = np.random.default_rng(21)
    rows = 3000
    for wide_rows, width, flavor in (([1500], 65535, 5), ([1500], 65536, 5), ([7, 1500, 2990], 3_000_000, 5),
                                     (range(rows), 65536, 0)):
        lens = rng.integers(2, 9, rows)
        row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        col_ind = np.concatenate([np.sort(rng.choice(2000, size=n, replace=False)) for n in lens]).astype(np.int32)
        for r in wide_rows:                                            # rows from column 0 to column `width`: their tiles' span
            j = int(row_ptr[r])
            col_ind[j], col_ind[j + lens[r] - 1] = 0, width
        cols = int(col_ind.max()) + 1
        val = rng.uniform(-1, 1, len(col_ind))
        x = rng.random(cols)
        ref = ob.csr_spmv(row_ptr, col_ind, val, x)
        A = sm.CsrMatrix(rows, cols, row_ptr, col_ind, val)
        dx = dev(torch, x)
        for tile in (1024, 2048):
            A.set_kernel(sm.CSR_KERNEL_STREAM, tile)
            assert A.describe()[0] == "csr_stream_owner<%d, %d, false>" % (tile // 256, flavor)
            dy = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
            A.spmv(dx, dy)
            torch.cuda.synchronize()
            assert np.array_equal(dy.cpu().numpy(), ref)
        A.set_kernel(sm.CSR_KERNEL_STREAM, 256)                     # 256-entry tiles always read col_ind
        assert A.describe()[0] == "csr_stream_owner<1, 0, false>"
        A.close()
    # the plan option (smvp_set_option): same bits without the offsets
    m, n, coo = load("memplus.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    x = rng.random(n)
    got = []
    for col16 in (1, 0):
        with sm.option("csr_col16", col16):
            A = sm.CsrMatrix(m, n, row_ptr, col_ind, val)
            A.set_kernel(sm.CSR_KERNEL_STREAM, 1024)
            assert A.describe()[0] == "csr_stream_owner<4, %d, false>" % (5 if col16 else 0)
            dy = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
            A.spmv(dev(torch, x), dy)
            torch.cuda.synchronize()
            got.append(dy.cpu().numpy())
            A.close()
    assert sm.get_option("csr_col16") == -1
    with pytest.raises(sm.SmvpError):
        sm.set_option("no_such_option", 1)
    assert np.array_equal(got[0], got[1])
    # round 5: tiles of 1024 / 2048 entries read every row's start as a 16-bit offset from the tile's first entry instead of
    # row_ptr (2 instead of 4 bytes per row).  Same bits with the development switch that keeps row_ptr -- CSR and TJDS, rows
    # without entries at the start, in the middle, at the end, a matrix whose entry count is a multiple of the tile
    lens = rng.integers(0, 9, 9000)
    lens[:5] = 0
    lens[4000:4100] = 0
    lens[-7:] = 0
    lens[2000] = 3000                                   # a row that runs past its tile's overflow area
    extra = (-int(lens.sum())) % 2048
    lens[100] += extra                                  # entries: a multiple of 2048, so the trailing empty rows sit at a tile edge
    rp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    assert rp2[-1] % 2048 == 0
    ci2 = np.concatenate([np.sort(rng.choice(5000, size=int(k), replace=False)) for k in lens if k]).astype(np.int32)
    v2 = rng.uniform(-1, 1, len(ci2))
    x2 = rng.random(5000)
    ref2 = ob.csr_spmv(rp2, ci2, v2, x2)
    sc2 = row_scale(rp2, ci2, v2, x2)
    coo2 = sm.make_coo(np.repeat(np.arange(len(lens)), lens), ci2, v2)
    got = {}
    for env in (None, "0"):
        with sm.option("csr_rowrel", None if env is None else 0):
            for tile in (1024, 2048):
                A = sm.CsrMatrix(len(lens), 5000, rp2, ci2, v2)
                A.set_kernel(sm.CSR_KERNEL_STREAM, tile)
                dy = torch.full((len(lens),), float("nan"), dtype=torch.float64, device="cuda")
                A.spmv(dev(torch, x2), dy)
                torch.cuda.synchronize()
                got["csr", tile, env] = dy.cpu().numpy()
                assert_close(got["csr", tile, env], ref2, sc2)
                A.close()
                T = sm.TjdsMatrix(sm.tjds_from_coo(coo2, len(lens), 5000))
                T.set_tile(tile)
                T.set_x(dev(torch, x2))
                dy = torch.full((len(lens),), float("nan"), dtype=torch.float64, device="cuda")
                T.spmv(dy)
                torch.cuda.synchronize()
                got["tjds", tile, env] = dy.cpu().numpy()
                assert_close(got["tjds", tile, env], ref2, sc2)
                T.close()
    for fmt in ("csr", "tjds"):
        for tile in (1024, 2048):
            assert np.array_equal(got[fmt, tile, None], got[fmt, tile, "0"])
    # and through the reference-shaped entry point with the kernel timing itself (the STAMPED instantiation of <4, 5, .>)
    n = 300_000
    band = np.clip(np.arange(n, dtype=np.int64)[:, None] + np.arange(-3, 4), 0, n - 1).astype(np.int32)   # no wrap: every tile is narrow
    vals = rng.uniform(-1, 1, band.size)
    coo = sm.make_coo(np.repeat(np.arange(n), 7), band.ravel(), vals)
    A = sm.CsrMatrix(n, n, (np.arange(n + 1, dtype=np.int64) * 7).astype(np.int32), band.ravel(), vals)
    assert A.describe()[0] == "csr_stream_owner<4, 5, false>"
    A.close()
    y, ms, st = sm.csr_compute(coo, n, n, iters=5, timing=sm.TIMING_DEVICE)
    assert sm.last_run_info().timing == sm.TIMING_DEVICE and 0 < st.time_min < 1.0
    assert np.array_equal(y, ob.csr_spmv((np.arange(n + 1, dtype=np.int64) * 7).astype(np.int32), band.ravel(), vals, np.ones(n)))


def test_colsweep_on_scattered_columns(torch):
    """The column-swept kernel on a config-4-shaped matrix (uniform columns over an operand larger than L2): AUTO picks
    it from its create-time estimate of the gather spread; every row is summed in ascending column order, so the result
    is bit for bit the serial loop's (main-cli.c:410-416) and the same from run to run, whatever the strip height."""
    rows, cols = 700_000, 3_000_000
    row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, cols, cols, 32, 0, rows)
    x = np.random.default_rng(9).random(cols)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    A = sm.CsrMatrix(rows, cols, row_ptr, col_ind, val)
    assert A.get_kernel() == (sm.CSR_KERNEL_COLSWEEP, 2736)      # 700 K rows: 256 workgroups of 2736 rows = ONE full generation (round 5)
    dx = dev(torch, x)
    dy = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    A.spmv(dx, dy)
    torch.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy(), ref)
    A.set_kernel(sm.CSR_KERNEL_STREAM, 0)
    dy.fill_(float("nan"))
    A.spmv(dx, dy)
    torch.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy(), ref)                  # 32 entries per row: the tile kernel is serial too
    # (round 6: strips of up to 5120 rows -- four of them fill a CU's 160 KB of LDS; 13 bits of the row word, turns capped at 7)
    for rb, want in ((0, 2736), (8192, 8192), (2048, 2048), (4096, 4096), (1024, 1024), (3000, 3000), (1028, 1028), (256, 256),
                     (9768, 9768), (20480, 20480)):
        A.set_kernel(sm.CSR_KERNEL_COLSWEEP, rb)
        assert A.get_kernel() == (sm.CSR_KERNEL_COLSWEEP, want) and A.describe()[0].startswith("csr_colsweep<")
        for _ in range(2):
            dy.fill_(float("nan"))
            A.spmv(dx, dy)
            torch.cuda.synchronize()
            assert np.array_equal(dy.cpu().numpy(), ref)
    # column parts (round 6, never AUTO): the workgroup's wavefronts share two strips / one strip, each taking a half / quarter of the
    # columns into partial sums of its own; a row's sum is its partial sums added part by part -- the same from run to run and
    # inside the rounding bound, not the serial loop's bits
    scale = ob.csr_spmv(row_ptr, col_ind, np.abs(val), np.abs(x))
    # ... and parts = 8: one column part per XCD (workgroup b of a launch takes eighth b % 8 of the columns for the four whole strips of row
    # group b / 8; the eight partial sums meet in scratch and are added by sweep_combine)
    for rb, parts in ((2736, 2), (2736, 4), (1024, 4), (5120, 4), (10240, 2), (0, 4), (2736, 8), (20480, 8), (256, 8), (0, 8)):
        A.set_kernel(sm.CSR_KERNEL_COLSWEEP, sm.sweep_parts(rb, parts))
        got_rb, got_parts = A.get_kernel()[1] & 0xffffff, 1 << (A.get_kernel()[1] >> 24)
        assert got_parts == parts and "%d column parts" % parts in A.describe()[0]
        assert got_rb == (rb or (2736 if parts < 8 else 10940))        # (parts = 8, height chosen: two generations of 32 row groups)
        assert A.launches() == (-(-(-(-rows // got_rb) * 8) // 256) + 1 if parts == 8 else A.launches())
        runs = []
        for _ in range(2):
            dy.fill_(float("nan"))
            A.spmv(dx, dy)
            torch.cuda.synchronize()
            runs.append(dy.cpu().numpy())
        assert np.array_equal(runs[0], runs[1]) and np.all(np.abs(runs[0] - ref) <= 1e-12 * scale)
        assert not np.array_equal(runs[0], ref) or parts == 1            # (association differs: some row's last bit does)
    for bad in (sm.sweep_parts(5124, 4), sm.sweep_parts(10244, 2), 1 << 28, sm.sweep_parts(2734, 2), sm.sweep_parts(20484, 8), sm.sweep_parts(252, 8)):
        with pytest.raises(sm.SmvpError):
            A.set_kernel(sm.CSR_KERNEL_COLSWEEP, bad)
    with pytest.raises(sm.SmvpError):
        A.set_kernel(sm.CSR_KERNEL_COLSWEEP, 3001)            # four strips per workgroup: a multiple of 4
    with pytest.raises(sm.SmvpError):
        A.set_kernel(sm.CSR_KERNEL_COLSWEEP, 20484)
    A.set_kernel(sm.CSR_KERNEL_AUTO, 0)                       # AUTO again: the sweep, plan rebuilt
    assert A.get_kernel() == (sm.CSR_KERNEL_COLSWEEP, 2736)
    dy.fill_(float("nan"))
    A.spmv(dx, dy)
    torch.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy(), ref)
    A.close()
    # through the reference-shaped entry point too
    coo = sm.make_coo(np.repeat(np.arange(2000), 32), col_ind[:64000] % 2000, val[:64000])
    y1, _, _ = sm.csr_compute(coo, 2000, 2000, iters=3, kernel=sm.CSR_KERNEL_COLSWEEP)
    y0, _, _ = sm.csr_compute(coo, 2000, 2000, iters=3)
    assert np.array_equal(y1, y0)


def test_colsweep_rows_that_meet_in_a_chunk(torch):
    """Entries of one row that arrive in the same 256-entry chunk of a strip's stream take turns; from the 31st on
    they are added lane by lane.  Dense rows in a narrow band of columns (and a few duplicate columns, which Matrix
    Market allows and the serial loop adds in input order) put whole chunks on that path: still the serial bits."""
    rng = np.random.default_rng(77)
    lens = [300, 1, 0, 257, 64, 31, 32, 33, 700] * 40 + [2] * 3000
    rows, cols = len(lens), 1024
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col_ind = np.concatenate([np.sort(rng.integers(0, cols, n)) for n in lens]).astype(np.int32)   # with repeats
    val = rng.uniform(-1, 1, len(col_ind))
    x = rng.random(cols)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    for rb in (0, 1024, 8192, 20480):                    # (20480: strips of 5120 rows, the turns capped at 7 since round 6)
        y = gpu_csr(torch, rows, cols, row_ptr, col_ind, val, x, sm.CSR_KERNEL_COLSWEEP, rb)
        assert np.array_equal(y, ref)
    # column parts: every part's chunks take their own turns; the sums are the serial ones up to the association part by part
    scale = row_scale(row_ptr, col_ind, val, x)
    for rb, parts in ((1024, 2), (1024, 4), (5120, 4), (1024, 8), (256, 8)):
        y = gpu_csr(torch, rows, cols, row_ptr, col_ind, val, x, sm.CSR_KERNEL_COLSWEEP, sm.sweep_parts(rb, parts))
        assert_close(y, ref, scale)
        assert np.array_equal(y, gpu_csr(torch, rows, cols, row_ptr, col_ind, val, x, sm.CSR_KERNEL_COLSWEEP, sm.sweep_parts(rb, parts)))


def test_binned_plan_on_a_row_block_of_a_sharded_matrix(torch):
    """A rank's row block [r0, r1) of the SURVEY 8(d) random model, as smvp_sharded.hip and bench.py create it
    (smvp_csr_create_block, first_row = r0): the diagonal lies at column r0 + local row, so the near / far split, the far
    share AUTO decides on and K6's windows of x are those of the whole matrix.  The same arrays created WITHOUT the offset
    -- what round 4 did at N > 1 -- look all far: every entry takes the 28-byte far path and K6 loads windows nothing points
    into.  Both give the right product; only the first is the plan that was measured at N = 1."""
    rows = 1 << 21
    r0, r1 = rows // 2, rows
    row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, rows, rows, 0, r0, r1)
    n_loc = r1 - r0
    x = sm.vector_random(rows)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    scale = row_scale(row_ptr, col_ind, val, x)
    rows_of = np.repeat(np.arange(r0, r1), np.diff(row_ptr))
    far_true = float((np.abs(col_ind.astype(np.int64) - rows_of) > 4096).mean())
    dx = dev(torch, x)
    A = sm.CsrMatrix(n_loc, rows, row_ptr, col_ind, val, first_row=r0)
    assert abs(A.far_share() - far_true) < 1e-6 and 0.3 < far_true < 0.5
    assert A.get_kernel() == (sm.CSR_KERNEL_BINNED, 4096) and A.describe()[0].startswith("csr_binned: csr_near_window + ")
    B = sm.CsrMatrix(n_loc, rows, row_ptr, col_ind, val)          # no offset: the band around the (true) diagonal counts as far
    assert B.far_share() > 0.95
    B.set_kernel(sm.CSR_KERNEL_BINNED, 0)
    for M in (A, B):
        dy = torch.full((n_loc,), float("nan"), dtype=torch.float64, device="cuda")
        M.spmv(dx, dy)
        torch.cuda.synchronize()
        assert_close(dy.cpu().numpy(), ref, scale)
    # the split is the whole matrix's: 20 B per far entry + 10 B per near slot -- the all-far plan is much larger
    assert A.plan_info()["plan_bytes"] < 0.8 * B.plan_info()["plan_bytes"]
    # the tile kernel's near engine and another band on the offset handle
    A.set_kernel(sm.CSR_KERNEL_BINNED, 512)
    dy = torch.full((n_loc,), float("nan"), dtype=torch.float64, device="cuda")
    A.spmv(dx, dy)
    torch.cuda.synchronize()
    assert_close(dy.cpu().numpy(), ref, scale)
    A.close()
    B.close()
    # the sharded layer passes every chunk's first row on: 2 and 3 virtual ranks x 2 chunks on the whole model, binned by choice
    full = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, rows, rows)
    ref_full = ob.csr_spmv(full[0], full[1], full[2], x)
    scale_full = row_scale(full[0], full[1], full[2], x)
    for ranks in (2, 3):
        S = sm.ShardedMatrix("csr", ranks, rows, rows, csr=full, devices=[0] * ranks, chunks=2, exchange=sm.EXCHANGE_DIRECT)
        S.set_csr_kernel(sm.CSR_KERNEL_BINNED, 0)
        S.set_x(x)
        S.spmv(allgather=sm.GATHER_OVERLAPPED)
        S.synchronize()
        assert_close(S.get_y(ranks - 1, gathered=True), ref_full, scale_full)
        S.close()


@pytest.mark.parametrize("near", ["window", "tile"])
def test_binned_plan_on_the_random_model(torch, near):
    """SURVEY 8(d)'s memplus-shaped random model (2^22 rows here): AUTO picks the binned plan -- near part with a row
    block's window of x in LDS (or, plan option "binned_near" = 1, on the tile kernel), far part through the
    LDS-binned passes; the result is within the normwise bound of the serial loop, the same bits from run to run, equal
    to the serial loop's bits on every short row without far entries; the plan's size is what smvp_csr_plan_info says."""
    sm.set_option("binned_near", 1 if near == "tile" else 0)        # (conftest.py resets the plan options after every test)
    rows = 1 << 22
    row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, rows, rows)
    x = sm.vector_random(rows)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    scale = row_scale(row_ptr, col_ind, val, x)
    A = sm.CsrMatrix(rows, rows, row_ptr, col_ind, val)
    assert A.get_kernel() == (sm.CSR_KERNEL_BINNED, 4096) and A.launches() == 3
    name, alg = A.describe()
    assert name.startswith("csr_binned: csr_near_window + " if near == "window" else "csr_binned: csr_stream_owner<")
    assert "csr_binned_far_products" in name and "csr_binned_far_sums" in name
    assert alg == 12.0 * len(val) + 4.0 * (rows + 1) + 16.0 * rows
    dx = dev(torch, x)
    ys = []
    for _ in range(3):
        dy = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
        A.spmv(dx, dy)
        torch.cuda.synchronize()
        ys.append(dy.cpu().numpy())
    assert_close(ys[0], ref, scale)
    assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])
    rows_of = np.repeat(np.arange(rows), np.diff(row_ptr))
    far = np.abs(col_ind.astype(np.int64) - rows_of) > 4096
    has_far = np.bincount(rows_of[far], minlength=rows) > 0
    short = np.diff(row_ptr) <= 16               # (the window plan sums a longer row across a wavefront)
    assert 0.3 < far.mean() < 0.5 and (short & ~has_far).sum() > 1000
    assert np.array_equal(ys[0][short & ~has_far], ref[short & ~has_far])
    info = A.plan_info()
    n, nf = len(val), int(far.sum())
    assert info["matrix_bytes"] == 12.0 * n + 4.0 * (rows + 1)
    # near copy 12 B (+ 2 B of column offsets) per near entry on the tile kernel, 10 B per slot (entries + 11 % padding) and
    # 2 B per row in the window plan; 20 B per far entry (two streams + the bins), 16 B per far row
    lo = 12.0 if near == "tile" else 10.0
    assert lo * (n - nf) + 20.0 * nf < info["plan_bytes"] < 14.5 * (n - nf) + 21.0 * nf + 24.0 * rows and info["build_ms"] > 0
    # the tile kernel on the same handle: the same product
    A.set_kernel(sm.CSR_KERNEL_STREAM, 0)
    dy = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    A.spmv(dx, dy)
    torch.cuda.synchronize()
    assert_close(dy.cpu().numpy(), ref, scale)
    assert A.plan_info()["plan_bytes"] < 3.0 * n
    # other bands, ones as the operand (the reference's), and back to AUTO
    for band in (64, 1 << 20):
        A.set_kernel(sm.CSR_KERNEL_BINNED, band)
        assert A.get_kernel() == (sm.CSR_KERNEL_BINNED, band)
        dy.fill_(float("nan"))
        A.spmv(dx, dy)
        torch.cuda.synchronize()
        assert_close(dy.cpu().numpy(), ref, scale)
    A.set_kernel(sm.CSR_KERNEL_AUTO, 0)
    assert A.get_kernel() == (sm.CSR_KERNEL_BINNED, 4096)
    ones = np.ones(rows)
    dy.fill_(float("nan"))
    A.spmv(dev(torch, ones), dy)
    torch.cuda.synchronize()
    assert_close(dy.cpu().numpy(), ob.csr_spmv(row_ptr, col_ind, val, ones), row_scale(row_ptr, col_ind, val, ones))
    A.close()


def test_binned_plan_corner_structures(torch):
    """The binned plan where its bookkeeping is stressed: far rows longer than a wavefront's 32 (summed by a whole
    wavefront) and longer than the cap of 1024 (kept near), rows with only far / only near entries, empty rows, more
    column blocks than a workgroup stages shifts for, duplicate and unsorted columns, a rectangular matrix."""
    rng = np.random.default_rng(2024)
    # wide: 3 M columns = 184 column blocks; rows of 0 ... 1500 entries, a few of them duplicates, columns NOT sorted
    rows, cols = 6000, 3_000_000
    lens = np.where(rng.random(rows) < 0.05, rng.integers(500, 1500, rows), rng.integers(0, 40, rows))
    lens[:3] = (0, 1, 33)
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col_ind = rng.integers(0, cols, int(row_ptr[-1])).astype(np.int32)
    idx = np.arange(0, len(col_ind) - 1, 97)
    col_ind[idx] = col_ind[idx + 1]                               # some repeats (mostly inside rows)
    val = rng.uniform(-1, 1, len(col_ind))
    x = rng.random(cols)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    scale = row_scale(row_ptr, col_ind, val, x)
    for band in (0, 5, 2_000_000):
        y = gpu_csr(torch, rows, cols, row_ptr, col_ind, val, x, sm.CSR_KERNEL_BINNED, band)
        assert_close(y, ref, scale)
    # tall and narrow: every column within a few blocks, many far rows per row block, rows == 0 far entries in stretches
    rows, cols = 200_000, 40_000
    lens = rng.integers(0, 9, rows)
    lens[50_000:90_000] = 0
    row_ptr, col_ind, val = csr_from_lengths(rng, lens.tolist(), cols)
    x = rng.random(cols)
    y = gpu_csr(torch, rows, cols, row_ptr, col_ind, val, x, sm.CSR_KERNEL_BINNED, 16)
    assert_close(y, ob.csr_spmv(row_ptr, col_ind, val, x), row_scale(row_ptr, col_ind, val, x))


def test_binned_near_engines(torch):
    """The near part's two engines -- the window plan (a row block's window of x in LDS, rows sorted by length, long rows
    summed by a wavefront) and the tile kernel -- against the oracle on structures that stress the window plan: several
    row blocks with a ragged last one, rows of 0 ... 16 (lanes of a slice), 17 ... 300 (slices of their own) entries, empty
    stretches, columns at the window's edges, a rectangular matrix with fewer columns than rows, rows that keep their far
    entries near (outside the window), unaligned operands, more long rows in a block than the window plan takes (it then
    stays on the tile kernel); the window plan's bits do not change from run to run, nor with pass A running beside the near
    part (a stream of its own, the default) or behind it."""
    rng = np.random.default_rng(77)
    cases = []
    for rows, cols, band in ((20_000, 20_000, 0), (8192 * 2 + 77, 30_000, 4096), (9_000, 2_500, 100), (70_000, 70_000, 1)):
        reach = band if band else 4096
        lens = rng.choice([0, 1, 2, 3, 5, 8, 16, 17, 40, 300], rows, p=[.15, .2, .2, .15, .1, .08, .05, .03, .03, .01])
        lens[rows // 3: rows // 3 + 700] = 0
        row_of = np.repeat(np.arange(rows), lens)
        off = rng.integers(-reach, reach + 1, row_of.size)
        edge = rng.random(row_of.size) < 0.1
        off[edge] = rng.choice([-reach, reach], int(edge.sum()))            # the window's last columns
        col = np.clip(row_of + off, 0, cols - 1)
        far = rng.random(row_of.size) < 0.15
        col[far] = rng.integers(0, cols, int(far.sum()))                    # some entries anywhere: the far passes
        heavy = np.flatnonzero(lens == 300)[:3]                              # rows with more far entries than the cap: kept near
        cases.append((rows, cols, band, lens, row_of, col, heavy))
    for rows, cols, band, lens, row_of, col, heavy in cases:
        extra_rows, extra_cols = [], []
        for r in heavy:
            extra_rows.append(np.full(1500, r))
            extra_cols.append(rng.integers(0, cols, 1500))
        all_rows = np.concatenate([row_of] + extra_rows)
        all_cols = np.concatenate([col] + extra_cols)
        order = np.lexsort((all_cols, all_rows))
        all_rows, all_cols = all_rows[order], all_cols[order]
        row_ptr = np.concatenate([[0], np.cumsum(np.bincount(all_rows, minlength=rows))]).astype(np.int32)
        col_ind = all_cols.astype(np.int32)
        val = rng.uniform(-1, 1, col_ind.size)
        x = rng.random(cols)
        ref = ob.csr_spmv(row_ptr, col_ind, val, x)
        scale = row_scale(row_ptr, col_ind, val, x)
        got = {}
        for near in ("window", "window, one stream", "tile"):
            sm.set_option("binned_near", 1 if near.startswith("tile") else 0)
            sm.set_option("binned_overlap", 0 if "one stream" in near else 1)   # pass A beside the near part, or behind it
            A = sm.CsrMatrix(rows, cols, row_ptr, col_ind, val)
            A.set_kernel(sm.CSR_KERNEL_BINNED, band)
            name = A.describe()[0]
            assert ("csr_near_window" in name) == near.startswith("window"), name
            buf_x = torch.zeros(cols + 1, dtype=torch.float64, device="cuda")
            buf_y = torch.full((rows + 1,), float("nan"), dtype=torch.float64, device="cuda")
            for shift in (0, 1):                                             # 16-byte aligned and not
                dx, dy = buf_x[shift:shift + cols], buf_y[shift:shift + rows]
                dx.copy_(dev(torch, x))
                dy.fill_(float("nan"))
                A.spmv(dx, dy)
                torch.cuda.synchronize()
                y = dy.cpu().numpy()
                assert_close(y, ref, scale)
                assert np.array_equal(y, got.setdefault(near, y))            # the same bits, aligned or not, run to run
            A.close()
        short = (np.diff(row_ptr) <= 16)
        assert np.array_equal(got["window"][short], got["tile"][short])     # both sum a short row left to right
        assert np.array_equal(got["window"], got["window, one stream"])     # the same kernels, side by side or one after the other
    # more long rows in a block than the window plan takes: it stays on the tile kernel
    sm.set_option("binned_near", 0)
    rows = 8192
    lens = np.full(rows, 40)
    row_ptr, col_ind, val = csr_from_lengths(rng, lens.tolist(), rows)
    A = sm.CsrMatrix(rows, rows, row_ptr, col_ind, val)
    A.set_kernel(sm.CSR_KERNEL_BINNED, 4096)
    assert "csr_stream_owner" in A.describe()[0]
    x = rng.random(rows)
    dy = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    A.spmv(dev(torch, x), dy)
    torch.cuda.synchronize()
    assert_close(dy.cpu().numpy(), ob.csr_spmv(row_ptr, col_ind, val, x), row_scale(row_ptr, col_ind, val, x))
    A.close()


def test_binned_products_in_stream_order(torch):
    """The binned plan runs pass A on a stream of its own beside the near part.  Towards the caller it must still behave like
    one stream: x written just before a product (by a copy on the caller's stream) is the x every kernel of that product reads,
    and y is complete for whatever the caller enqueues next -- a power iteration on a side stream of the caller's, five
    products with the operand rewritten from y in between, against the same iteration on the host."""
    rows = 1 << 20
    row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 4242, rows, rows)
    A = sm.CsrMatrix(rows, rows, row_ptr, col_ind, val)
    A.set_kernel(sm.CSR_KERNEL_BINNED, 0)
    assert "csr_near_window" in A.describe()[0]
    x = sm.vector_random(rows)
    want = x.copy()
    for _ in range(5):
        want = ob.csr_spmv(row_ptr, col_ind, val, want)
        want /= np.abs(want).max()
    s = torch.cuda.Stream()
    dx, dy = dev(torch, x), torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for _ in range(5):
            A.spmv(dx, dy, stream=s)
            dx.copy_(dy / dy.abs().max())          # enqueued on s right behind the product; the next product reads it
            dy.fill_(float("nan"))                 # ... and nothing of the product may still be writing y
    s.synchronize()
    got = dx.cpu().numpy()
    assert np.all(np.isfinite(got)) and np.max(np.abs(got - want)) < 1e-9
    A.close()


@pytest.mark.parametrize("kernel,param", CSR_VARIANTS)
def test_csr_matrix_without_rows(torch, kernel, param):
    """rows == 0 (the sharded layer makes such handles for empty chunks): every family plans and launches nothing."""
    A = sm.CsrMatrix(0, 7, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0))
    A.set_kernel(kernel, param)
    assert A.get_kernel()[0] == kernel
    A.spmv(torch.ones(7, dtype=torch.float64, device="cuda"), torch.zeros(1, dtype=torch.float64, device="cuda"))
    torch.cuda.synchronize()
    A.close()


def test_plan_info_bounds(torch):
    """smvp_csr_plan_info / smvp_tjds_plan_info: what each launch plan keeps beside the format's own arrays.  The tile plan
    of CSR is a few words per tile (+ 2 B per entry where the 16-bit column offsets engage): never more than half the
    matrix again; the column sweep and the binned plan keep the entries a second time (about 14 / 12.6 = 1.1 and up to
    20 / 12 = 1.7 times the matrix); the one-kernel TJDS plan 6.2 B of index per entry plus the cached values."""
    m, n, coo = load("memplus.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    A = sm.CsrMatrix(m, n, row_ptr, col_ind, val)
    i = A.plan_info()
    assert i["matrix_bytes"] == 12.0 * len(val) + 4.0 * (m + 1) and 0 < i["plan_bytes"] <= 0.5 * i["matrix_bytes"] and i["build_ms"] > 0
    A.set_kernel(sm.CSR_KERNEL_VECTOR, 8)
    assert A.plan_info()["plan_bytes"] == 0
    A.set_kernel(sm.CSR_KERNEL_COLSWEEP, 0)
    assert 1.0 * i["matrix_bytes"] < A.plan_info()["plan_bytes"] < 1.25 * i["matrix_bytes"]
    A.set_kernel(sm.CSR_KERNEL_BINNED, 64)
    assert 0.9 * i["matrix_bytes"] < A.plan_info()["plan_bytes"] < 2.2 * i["matrix_bytes"]
    A.close()
    T = sm.TjdsMatrix(sm.tjds_from_coo(coo, m, n))
    t = T.plan_info()
    assert t["matrix_bytes"] == 12.0 * len(val) + 4.0 * (T.t.num_diag + 1) + 4.0 * n if hasattr(T, "t") else t["matrix_bytes"] > 12.0 * len(val)
    assert 0.5 * t["matrix_bytes"] < t["plan_bytes"] < 1.7 * t["matrix_bytes"] and t["build_ms"] > 0
    T.set_mode(sm.TJDS_MODE_TWO_PHASE)
    assert T.plan_info()["plan_bytes"] > t["plan_bytes"] + 12.0 * len(val)       # + the products and the row-inverted index
    T.close()
    # a 16-bit-offset plan on a large banded matrix: tile words + 2 B per entry + 2 B per row (the 16-bit row offsets)
    nb = 2_000_000
    band = ((np.arange(nb, dtype=np.int64)[:, None] + np.arange(-4, 4)) % nb).astype(np.int32)
    band.sort(axis=1)
    B = sm.CsrMatrix(nb, nb, (np.arange(nb + 1, dtype=np.int64) * 8).astype(np.int32), band.ravel(), np.ones(8 * nb))
    b = B.plan_info()
    assert B.describe()[0] == "csr_stream_owner<8, 5, false>" and 2.0 * 8 * nb + 2.0 * nb < b["plan_bytes"] < 2.1 * 8 * nb + 2.0 * nb
    assert b["plan_bytes"] <= 1.5 * b["matrix_bytes"]
    B.close()


def test_auto_plan_choice(torch):
    """AUTO: owner-completes tiles by default, the carry form when some row is extremely long; never the vector kernel;
    the column sweep only for large matrices whose gathers scatter over an operand much larger than the L2."""
    # a band of 8 entries per row, 32 M entries over a 32 MB operand: neighbouring gathers share lines -> tile kernel;
    # the SURVEY 8(d) random model (39 % of its entries uniform over the operand: spread about 0.4) -> the binned plan
    n = 4_000_000
    band = ((np.arange(n, dtype=np.int64)[:, None] + np.arange(-4, 4)) % n).astype(np.int32)
    band.sort(axis=1)
    A = sm.CsrMatrix(n, n, (np.arange(n + 1, dtype=np.int64) * 8).astype(np.int32), band.ravel(), np.ones(8 * n))
    assert A.get_kernel()[0] == sm.CSR_KERNEL_STREAM
    A.close()
    row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 5, 1 << 22, 1 << 22)
    A = sm.CsrMatrix(1 << 22, 1 << 22, row_ptr, col_ind, val)
    assert A.get_kernel() == (sm.CSR_KERNEL_BINNED, 4096) and 0.2 <= A.gather_spread() < 0.6
    A.close()
    rng = np.random.default_rng(11)
    for lens, want in (([5] * 300000, sm.CSR_KERNEL_STREAM), ([3] * 500 + [40000] + [2] * 500, sm.CSR_KERNEL_STREAM_CARRY),
                       ([128] * 300, sm.CSR_KERNEL_STREAM)):
        row_ptr, col_ind, val = csr_from_lengths(rng, lens, 50000)
        A = sm.CsrMatrix(len(lens), 50000, row_ptr, col_ind, val)
        assert A.get_kernel()[0] == want
        x = rng.random(50000)
        dy = torch.empty(len(lens), dtype=torch.float64, device="cuda")
        A.spmv(dev(torch, x), dy)
        torch.cuda.synchronize()
        assert_close(dy.cpu().numpy(), ob.csr_spmv(row_ptr, col_ind, val, x), row_scale(row_ptr, col_ind, val, x))
        A.close()


TJDS_MODES = [sm.TJDS_MODE_ROW_GATHER, sm.TJDS_MODE_TWO_PHASE, sm.TJDS_MODE_ATOMIC]
# the one-kernel product: (index form of its stream, entries per tile); "half" = every tile in TJDS order, position + a
# 16-bit slot | run-hint word per entry (default); "sorted" = the same order with a 32-bit slot | diagonal word; "k32" = row order
TJDS_GATHER_VARIANTS = [("half", 0), ("half", 256), ("half", 1024), ("half", 2048), ("sorted", 0), ("sorted", 256),
                        ("sorted", 1024), ("sorted", 2048), ("k32", 256), ("k32", 1024), ("k32", 2048)]
TJDS_FLAVORS = {"half": (4,), "sorted": (3,), "k32": (2,)}


def tjds_gather_matrix(t, index, tile):
    """TjdsMatrix on the one-kernel product with the given stream form and tile size (0 = the plan's own choice)."""
    with sm.option("tjds_index", {"half": 0, "sorted": 1, "k32": 2}[index]):
        T = sm.TjdsMatrix(t)
    if tile:
        T.set_tile(tile)
    name = T.describe()[0]
    assert any(name.endswith(", %d, false>" % f) for f in TJDS_FLAVORS[index]), name
    if tile:
        assert any(name == "csr_stream_owner<%d, %d, false>" % (tile // 256, f) for f in TJDS_FLAVORS[index]), name
    return T


def test_tjds_value_cache_and_run_words_do_not_change_a_bit(torch):
    """The one-kernel product sums every row in ascending TJDS position whatever the plan keeps: values read from val or
    from the tiles' cache (lines shared by >= 1, 2, 8, 16 tiles, or none), the second index word 32 or 16 bits wide, any tile
    size -- the same bits every time."""
    for name in ("memplus.mtx", "pwt.mtx"):
        m, n, coo = load(name)
        t = sm.tjds_from_coo(coo, m, n)
        x = dev(torch, np.random.default_rng(3).random(n))
        want = None
        for index in ("sorted", "half"):
            for tile in (256, 2048):
                T = tjds_gather_matrix(t, index, tile)
                T.set_x(x)
                for cache in (0, 1, 2, 8, 16):
                    T.set_value_cache(cache)
                    got_min, cached = T.get_value_cache()
                    assert got_min == cache and (cached == 0) == (cache == 0) or cache > 8
                    if cache == 1:
                        assert cached == len(coo)
                    dy = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
                    T.spmv(dy)
                    torch.cuda.synchronize()
                    want = dy.clone() if want is None else want
                    assert torch.equal(dy, want), (name, index, tile, cache)
                T.close()


def test_tjds_half_words_with_a_run_per_entry(torch):
    """A row whose 1000 entries each lie in another jagged diagonal: its tile's sorted entries form 1000 runs of one entry,
    32 runs inside every group of 32 -- the most the 5-bit run hint has to tell apart."""
    m, ncol = 1100, 1000
    rows = np.concatenate([np.arange(j) for j in range(ncol)] + [np.full(ncol, m - 1)])       # column j: rows 0..j-1, then m-1
    cols = np.concatenate([np.full(j, j) for j in range(ncol)] + [np.arange(ncol)])
    rng = np.random.default_rng(12)
    coo = sm.make_coo(rows, cols, rng.uniform(-1, 1, len(rows)))
    t = sm.tjds_from_coo(coo, m, ncol)
    assert t.num_diag == ncol
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    x = rng.random(ncol)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    T = tjds_gather_matrix(t, "half", 0)
    for tile in (256, 2048, 1024):
        for cache in (4, 0, 1):
            T.set_tile(tile)
            T.set_value_cache(cache)
            assert T.describe()[0] == "csr_stream_owner<%d, 4, false>" % (tile // 256)
            dy = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
            T.set_x(dev(torch, x))
            T.spmv(dy)
            torch.cuda.synchronize()
            assert_close(dy.cpu().numpy(), ref, row_scale(row_ptr, col_ind, val, x))
    T.close()


def test_tjds_half_words_across_position_blocks(torch):
    """The 16-bit position words count from the base of an aligned block of 2^16 positions (cached entries: columns): a matrix
    whose tiles reach over many such blocks -- 300 000 columns, uniformly scattered -- gives the same bits as the plan that
    keeps whole 32-bit words, with and without the value cache, and agrees with the serial loop."""
    m = n = 300_000
    rng = np.random.default_rng(77)
    per_row = rng.integers(1, 12, m)
    rows = np.repeat(np.arange(m), per_row)
    cols = rng.integers(0, n, len(rows))
    key = np.unique(rows.astype(np.int64) * n + cols)
    rows, cols = (key // n).astype(np.int64), (key % n).astype(np.int64)
    coo = sm.make_coo(rows, cols, rng.uniform(-1, 1, len(rows)))
    t = sm.tjds_from_coo(coo, m, n)
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    x = rng.random(n)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    scale = row_scale(row_ptr, col_ind, val, x)
    want = None
    for index in ("sorted", "half"):
        for tile in (2048, 256, 1024):
            T = tjds_gather_matrix(t, index, tile)
            T.set_x(dev(torch, x))
            for cache in (2, 0, 1):
                T.set_value_cache(cache)
                dy = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
                T.spmv(dy)
                torch.cuda.synchronize()
                if want is None:
                    want = dy.clone()
                    assert_close(dy.cpu().numpy(), ref, scale)
                assert torch.equal(dy, want), (index, tile, cache)
            T.close()


@pytest.mark.parametrize("name", SAMPLES)
@pytest.mark.parametrize("mode", TJDS_MODES)
def test_tjds_sample_matrices(torch, name, mode):
    m, n, coo = load(name)
    t = sm.tjds_from_coo(coo, m, n)
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    T = sm.TjdsMatrix(t)
    assert T.describe()[0].startswith("csr_stream_owner<")     # the default is the one-kernel product
    T.set_mode(mode)
    assert ("products" in T.describe()[0]) == (mode == sm.TJDS_MODE_TWO_PHASE)
    assert ("scatter" in T.describe()[0]) == (mode == sm.TJDS_MODE_ATOMIC)
    for x in (np.ones(n), np.random.default_rng(67890).random(n)):
        ref = ob.csr_spmv(row_ptr, col_ind, val, x)          # a correct TJDS computes A x
        dy = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
        T.set_x(dev(torch, x))
        T.zero_y(dy)
        T.spmv(dy)
        torch.cuda.synchronize()
        y = dy.cpu().numpy()
        assert_close(y, ref, row_scale(row_ptr, col_ind, val, x), exact=(name in EXACT and x[0] == 1.0))
        assert_close(y, ob.tjds_spmv(ob.tjds_build(coo, m, n), x), row_scale(row_ptr, col_ind, val, x))
    T.close()


@pytest.mark.parametrize("name", [s for s in SAMPLES if REPORTS[s][1]])
def test_tjds_ref_quirks_reproduce_committed_reports(torch, name):
    """--ref-quirks: same kernel, host-edited plan, reproduces the reference's (defective) TJDS reports."""
    m, n, coo = load(name)
    t = sm.tjds_from_coo(coo, m, n)
    T = sm.TjdsMatrix(t)
    T.set_ref_quirks(True)
    dy = torch.zeros(m, dtype=torch.float64, device="cuda")
    T.set_x(dev(torch, np.ones(n)))
    T.zero_y(dy)
    T.spmv(dy)
    torch.cuda.synchronize()
    y = dy.cpu().numpy()
    want = ob.report_y_lines(ob.read_report("smvp-toolbox_report_TJDS_%s.txt" % REPORTS[name][1]))
    oracle_y = ob.tjds_spmv(ob.tjds_build(coo, m, n), np.ones(n), refquirks=True)
    if name in EXACT:
        assert ob.fmt_g(y) == want
        assert np.array_equal(y, oracle_y)
    else:
        # memplus: atomics reorder the sums; compare normwise and the %g text on well-conditioned rows
        row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
        scale = row_scale(row_ptr, col_ind, val, np.ones(n))
        assert_close(y, oracle_y, scale)
        got = ob.fmt_g(y)
        well = np.abs(oracle_y) > 1e-6 * scale
        diff = [i for i in np.flatnonzero(well) if got[i] != want[i]]
        assert len(diff) <= 0.001 * m
    # back to the corrected product
    T.set_ref_quirks(False)
    T.zero_y(dy)
    T.spmv(dy)
    torch.cuda.synchronize()
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    assert_close(dy.cpu().numpy(), ob.csr_spmv(row_ptr, col_ind, val, np.ones(n)),
                 row_scale(row_ptr, col_ind, val, np.ones(n)))
    T.close()


# ------------------------------------------------------------------- edge cases
def csr_from_lengths(rng, lens, cols):
    rows = len(lens)
    row_ptr = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(lens, out=row_ptr[1:])
    col_ind = np.concatenate([np.sort(rng.choice(cols, size=l, replace=False)) for l in lens] + [np.zeros(0, int)])
    val = rng.uniform(-1, 1, int(row_ptr[-1]))
    return row_ptr, col_ind.astype(np.int32), val


EDGE_CASES = {
    "empty_matrix": lambda rng: ([0, 0, 0, 0], 5),
    "single_entry": lambda rng: ([1], 1),
    "leading_and_trailing_empty_rows": lambda rng: ([0, 0, 0, 3, 0, 2, 0, 0], 7),
    "all_rows_empty_but_one": lambda rng: ([0] * 500 + [40] + [0] * 500, 64),
    "row_spanning_many_tiles": lambda rng: ([3, 5000, 2, 0, 7000, 1], 8192),
    "row_ending_exactly_on_tile_edges": lambda rng: ([1024, 1024, 2048, 1, 1023], 4096),
    "rows_just_past_a_tile_edge": lambda rng: ([1020, 5, 1, 1000, 30, 260, 763, 1, 300, 700, 280], 4096),
    "one_huge_row_between_short_ones": lambda rng: ([2] * 100 + [40000] + [3] * 100, 65536),
    "many_short_rows": lambda rng: (rng.integers(0, 4, 20000).tolist(), 300),
    "mixed_skew": lambda rng: (np.where(rng.random(3000) < 0.01, rng.integers(200, 3000, 3000),
                                        rng.integers(0, 12, 3000)).tolist(), 5000),
    "wide_rectangular": lambda rng: (rng.integers(0, 50, 100).tolist(), 100000),
    "tall_rectangular": lambda rng: (rng.integers(0, 3, 50000).tolist(), 3),
    "exactly_33_per_row": lambda rng: ([33] * 777, 100),
}


@pytest.mark.parametrize("case", sorted(EDGE_CASES))
@pytest.mark.parametrize("kernel,param", CSR_VARIANTS)
def test_csr_edge_cases(torch, case, kernel, param):
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    lens, cols = EDGE_CASES[case](rng)
    row_ptr, col_ind, val = csr_from_lengths(rng, lens, cols)
    rows = len(lens)
    x = rng.random(cols)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    y = gpu_csr(torch, rows, cols, row_ptr, col_ind, val, x, kernel, param)
    assert_close(y, ref, row_scale(row_ptr, col_ind, val, x), exact=kernel == sm.CSR_KERNEL_COLSWEEP)


@pytest.mark.parametrize("name", SAMPLES)
@pytest.mark.parametrize("index,tile", TJDS_GATHER_VARIANTS)
def test_tjds_row_gather_variants_on_samples(torch, name, index, tile):
    """Every stream form / tile size of the one-kernel TJDS product against the oracle's CSR and TJDS results."""
    m, n, coo = load(name)
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    T = tjds_gather_matrix(sm.tjds_from_coo(coo, m, n), index, tile)
    for x in (np.ones(n), np.random.default_rng(67890).random(n)):
        dy = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
        T.set_x(dev(torch, x))
        T.spmv(dy)
        torch.cuda.synchronize()
        y = dy.cpu().numpy()
        scale = row_scale(row_ptr, col_ind, val, x)
        assert_close(y, ob.csr_spmv(row_ptr, col_ind, val, x), scale, exact=(name in EXACT and x[0] == 1.0))
        assert_close(y, ob.tjds_spmv(ob.tjds_build(coo, m, n), x), scale)
    T.close()


def test_tjds_products_are_bit_reproducible_and_need_no_zeroing(torch):
    """The one-kernel and the two-phase product sum each row's products in one fixed order (ascending TJDS position):
    identical bits run to run, between the two forms and between tile sizes; y may hold garbage on entry."""
    m, n, coo = load("memplus.mtx")
    T = sm.TjdsMatrix(sm.tjds_from_coo(coo, m, n))
    T.set_x(dev(torch, np.random.default_rng(2).random(n)))
    ys = []
    for fill in (float("nan"), 7.0, -1e300):
        dy = torch.full((m,), fill, dtype=torch.float64, device="cuda")
        T.zero_y(dy)          # a no-op in this mode
        T.spmv(dy)
        torch.cuda.synchronize()
        ys.append(dy.clone())
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    for mode, tile in ((sm.TJDS_MODE_TWO_PHASE, 0), (sm.TJDS_MODE_ROW_GATHER, 1024), (sm.TJDS_MODE_ROW_GATHER, 2048)):
        T.set_mode(mode)
        if tile:
            T.set_tile(tile)
        dy = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
        T.spmv(dy)
        torch.cuda.synchronize()
        assert torch.equal(dy, ys[0]), (mode, tile)
    T.set_mode(sm.TJDS_MODE_ROW_GATHER)
    # the same through device-built arrays
    d_coo = torch.from_numpy(np.ascontiguousarray(coo).view(np.uint8).copy()).cuda()
    T2 = sm.TjdsMatrix(sm.tjds_from_coo_device(d_coo, m, n, len(coo)))
    T2.set_x(dev(torch, np.random.default_rng(2).random(n)))
    dy = torch.empty(m, dtype=torch.float64, device="cuda")
    T2.spmv(dy)
    torch.cuda.synchronize()
    assert torch.equal(dy, ys[0])


@pytest.mark.parametrize("index,tile", TJDS_GATHER_VARIANTS)
@pytest.mark.parametrize("case", sorted(EDGE_CASES))
def test_tjds_row_gather_edge_cases(torch, case, index, tile):
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    lens, cols = EDGE_CASES[case](rng)
    row_ptr, col_ind, val = csr_from_lengths(rng, lens, cols)
    rows = len(lens)
    coo = sm.make_coo(np.repeat(np.arange(rows), lens), col_ind, val)
    x = rng.random(cols)
    T = tjds_gather_matrix(sm.tjds_from_coo(coo, rows, cols), index, tile)
    T.set_x(dev(torch, x))
    dy = torch.full((max(rows, 1),), float("nan"), dtype=torch.float64, device="cuda")
    T.spmv(dy)
    torch.cuda.synchronize()
    assert_close(dy.cpu().numpy()[:rows], ob.csr_spmv(row_ptr, col_ind, val, x), row_scale(row_ptr, col_ind, val, x))
    T.close()


@pytest.mark.parametrize("mode", TJDS_MODES)
@pytest.mark.parametrize("case", sorted(EDGE_CASES))
def test_tjds_edge_cases_handles(torch, case, mode):
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    lens, cols = EDGE_CASES[case](rng)
    row_ptr, col_ind, val = csr_from_lengths(rng, lens, cols)
    rows = len(lens)
    coo = sm.make_coo(np.repeat(np.arange(rows), lens), col_ind, val)
    x = rng.random(cols)
    T = sm.TjdsMatrix(sm.tjds_from_coo(coo, rows, cols))
    T.set_mode(mode)
    T.set_x(dev(torch, x))
    dy = torch.full((max(rows, 1),), float("nan"), dtype=torch.float64, device="cuda")
    T.zero_y(dy)
    T.spmv(dy)
    torch.cuda.synchronize()
    assert_close(dy.cpu().numpy()[:rows], ob.csr_spmv(row_ptr, col_ind, val, x), row_scale(row_ptr, col_ind, val, x))


@pytest.mark.parametrize("case", sorted(EDGE_CASES))
def test_tjds_edge_cases(torch, case):
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    lens, cols = EDGE_CASES[case](rng)
    row_ptr, col_ind, val = csr_from_lengths(rng, lens, cols)
    rows = len(lens)
    coo = sm.make_coo(np.repeat(np.arange(rows), lens), col_ind, val)
    coo = coo[rng.permutation(len(coo))]
    x = rng.random(cols)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    y, ms, st = sm.tjds_compute(coo, rows, cols, iters=2, x=x)
    assert_close(y, ref, row_scale(row_ptr, col_ind, val, x))
    y, ms, st = sm.csr_compute(coo, rows, cols, iters=2, x=x)
    assert_close(y, ref, row_scale(row_ptr, col_ind, val, x))


def test_create_rejects_malformed_arrays(torch):
    good = (np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32), np.ones(2))
    sm.CsrMatrix(2, 2, *good).close()
    with pytest.raises(sm.SmvpError):
        sm.CsrMatrix(2, 2, np.array([0, 2, 1], np.int32), good[1], good[2])     # decreasing row_ptr
    with pytest.raises(sm.SmvpError):
        sm.CsrMatrix(2, 2, good[0], np.array([0, 7], np.int32), good[2])        # column out of range
    t = sm.tjds_from_coo(sm.make_coo([0, 1], [0, 1], [1.0, 2.0]), 2, 2)
    t.row_ind = np.array([0, 9], np.int32)
    with pytest.raises(sm.SmvpError):
        sm.TjdsMatrix(t)


def test_adopted_device_arrays_are_range_checked(torch):
    """A column / row index outside the matrix in an adopted device array must be refused, not dereferenced."""
    rp = dev(torch, np.array([0, 2, 3], np.int32))
    v = dev(torch, np.ones(4))[:3]
    with pytest.raises(sm.SmvpError) as e:
        sm.CsrMatrix(2, 5, rp, dev(torch, np.array([0, 9, 1, 0], np.int32))[:3], v)
    assert e.value.code == sm.ERR_INVALID and "col_ind[1]" in str(e.value)
    with pytest.raises(sm.SmvpError):
        sm.CsrMatrix(2, 5, rp, dev(torch, np.array([0, -1, 1, 0], np.int32))[:3], v)
    t = sm.tjds_from_coo_device(_coo_to_device(torch, sm.make_coo([0, 1], [0, 1], [1.0, 2.0])), 2, 2, 2)
    t.row_ind = dev(torch, np.array([0, 7], np.int32))
    with pytest.raises(sm.SmvpError):
        sm.TjdsMatrix(t)


def test_adopted_device_arrays(torch):
    """SMVP_MEM_DEVICE: arrays that already live in HBM (torch tensors) are used in place."""
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 20, 5000).tolist()
    row_ptr, col_ind, val = csr_from_lengths(rng, lens, 4000)
    x = rng.random(4000)
    A = sm.CsrMatrix(5000, 4000, dev(torch, row_ptr), dev(torch, col_ind), dev(torch, val))
    dy = torch.empty(5000, dtype=torch.float64, device="cuda")
    A.spmv(dev(torch, x), dy)
    torch.cuda.synchronize()
    assert_close(dy.cpu().numpy(), ob.csr_spmv(row_ptr, col_ind, val, x), row_scale(row_ptr, col_ind, val, x))


def test_runs_on_a_non_default_stream(torch):
    m, n, coo = load("memplus.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    A = sm.CsrMatrix(m, n, row_ptr, col_ind, val)
    s = torch.cuda.Stream()
    dx = dev(torch, np.ones(n))
    dy = torch.zeros(m, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(5):
        A.spmv(dx, dy, stream=s)
    s.synchronize()
    assert_close(dy.cpu().numpy(), ob.csr_spmv(row_ptr, col_ind, val, np.ones(n)),
                 row_scale(row_ptr, col_ind, val, np.ones(n)))


# ------------------------------------------- full-size synthetic: properties only
@pytest.fixture(scope="module")
def big(torch):
    """memplus-shaped synthetic at 2^22 rows (~30 M entries, ~0.44 GB of SpMV traffic)."""
    M = 1 << 22
    row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, M, M)
    return M, row_ptr, col_ind, val


def test_big_synthetic_properties(torch, big):
    M, row_ptr, col_ind, val = big
    A = sm.CsrMatrix(M, M, row_ptr, col_ind, val)
    ones = torch.ones(M, dtype=torch.float64, device="cuda")
    y1 = torch.empty(M, dtype=torch.float64, device="cuda")
    A.spmv(ones, y1)
    torch.cuda.synchronize()
    # (1) x = ones: y is the row sum -- an independent host computation
    host = np.add.reduceat(val, row_ptr[:-1])
    scale = np.add.reduceat(np.abs(val), row_ptr[:-1])
    assert np.all(np.abs(y1.cpu().numpy() - host) <= TOL * scale)
    # (2) kernel families agree with each other
    rng = np.random.default_rng(67890)
    xa, xb = dev(torch, rng.random(M)), dev(torch, rng.random(M))
    ya, yb, yab, yv = (torch.empty(M, dtype=torch.float64, device="cuda") for _ in range(4))
    A.spmv(xa, ya)
    A.spmv(xb, yb)
    A.spmv(xa + xb, yab)
    A.set_kernel(sm.CSR_KERNEL_VECTOR, 8)
    A.spmv(xa, yv)
    torch.cuda.synchronize()
    sc = torch.from_numpy(scale).cuda() * 2
    assert bool(((ya - yv).abs() <= TOL * sc).all())
    # (3) linearity: A(xa + xb) = A xa + A xb
    assert bool(((yab - (ya + yb)).abs() <= TOL * sc).all())
    # (4) a slice of rows against the oracle
    k = 100_000
    sub = ob.csr_spmv(row_ptr[:k + 1].copy(), col_ind[:row_ptr[k]], val[:row_ptr[k]], xa.cpu().numpy())
    assert np.all(np.abs(ya.cpu().numpy()[:k] - sub) <= TOL * 2 * scale[:k])
    # (5) idempotence: the same launch twice gives the same bits (no atomics on the CSR path) -- the plan AUTO picks for
    # this matrix (the binned one) and the tile kernel each repeat themselves; between them the normwise bound holds
    y2 = torch.empty_like(ya)
    for kernel in (sm.CSR_KERNEL_AUTO, sm.CSR_KERNEL_STREAM):
        A.set_kernel(kernel, 0)
        A.spmv(xa, y2)
        y3 = torch.empty_like(ya)
        A.spmv(xa, y3)
        torch.cuda.synchronize()
        assert torch.equal(y2, y3)
        assert torch.equal(ya, y2) if kernel == sm.CSR_KERNEL_AUTO else bool(((ya - y2).abs() <= TOL * sc).all())
    A.close()


def test_big_synthetic_tjds_agrees_with_csr(torch):
    M = 1 << 20
    row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, M, M)
    coo = sm.make_coo(np.repeat(np.arange(M), np.diff(row_ptr)), col_ind, val)
    x = np.random.default_rng(3).random(M)
    y_t, _, _ = sm.tjds_compute(coo, M, M, iters=1, x=x)
    y_c, _, _ = sm.csr_compute(coo, M, M, iters=1, x=x)
    scale = np.add.reduceat(np.abs(val), row_ptr[:-1])
    assert np.all(np.abs(y_t - y_c) <= TOL * scale)


# ------------------------------------------------------------ entry points + CLI
@pytest.mark.parametrize("name", SAMPLES)
def test_reference_shaped_entry_points(torch, name):
    m, n, coo = load(name)
    before = coo.tobytes()
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    ref = ob.csr_spmv(row_ptr, col_ind, val, np.ones(n))
    scale = row_scale(row_ptr, col_ind, val, np.ones(n))
    y, ms, st = sm.csr_compute(coo, m, n, iters=20)
    assert_close(y, ref, scale, exact=name in EXACT)
    assert len(ms) == 20 and np.all(ms > 0) and st.time_total == pytest.approx(ms.sum())
    assert st.time_min == ms.min() and st.time_max == ms.max() and st.time_stdev == pytest.approx(ms.std())
    y, ms, st = sm.tjds_compute(coo, m, n, iters=20)
    assert_close(y, ref, scale, exact=name in EXACT)
    assert coo.tobytes() == before


@pytest.mark.parametrize("name", ["ibm32.mtx", "curtis54.mtx", "pwt.mtx"])
def test_cli_all_algs_end_to_end(torch, name, tmp_path):
    """BASELINE config 1/5 plumbing: --all-algs -n 1000 writes both reports; y blocks equal the committed ones."""
    path = ob.fixture_path(name)
    p = subprocess.run([sm.CLI_PATH, "--all-algs", "-n", "1000", "-d", str(tmp_path), path],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    for tag in ("[START]", "Converting loaded content to CSR format.", "Calculating 1000 iterations of SMVP CSR.",
                "Converting loaded content to TJDS format.", "Calculating 1000 iterations of SMVP TJDS.",
                "Execution report file saved as:", "[STOP]\tExit smvp-toolbox v0.6.4"):
        assert tag in p.stdout
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 2 and files[0].startswith("smvp-toolbox_report_CSR_") and \
        files[1].startswith("smvp-toolbox_report_TJDS_")
    csr_ref = ob.read_report("smvp-toolbox_report_CSR_%s.txt" % REPORTS[name][0])
    got = open(tmp_path / files[0]).read()
    assert _mask(got).replace(path, "PATH") == _mask(csr_ref).replace(csr_ref.split("\n")[4], "PATH")
    # corrected TJDS: y block equals the CSR report's y block
    got_t = open(tmp_path / files[1]).read()
    assert ob.report_y_lines(got_t) == ob.report_y_lines(csr_ref)
    assert "TJDS algorithm" in got_t


def test_cli_memplus_csr_and_tjds_n1000(torch, tmp_path):
    """BASELINE configs 2 and 3 literally: `memplus.mtx -c -t -n 1000` through the command line.  The report's "%g" text
    equals the committed report's on every row summed in the serial order (up to 32 entries: one lane) and on every
    well-conditioned longer row; both reports carry 1000 products' statistics; stdout says how the window was taken."""
    path = ob.fixture_path("memplus.mtx")
    p = subprocess.run([sm.CLI_PATH, "-c", "-t", "-n", "1000", "-d", str(tmp_path), path], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    for tag in ("Calculating 1000 iterations of SMVP CSR.", "Calculating 1000 iterations of SMVP TJDS.",
                "CSR timing:", "TJDS timing:", "in the report file", "host wall per product"):
        assert tag in p.stdout, tag
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 2
    m, n, coo = load("memplus.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    ref = ob.csr_spmv(row_ptr, col_ind, val, np.ones(n))
    scale = row_scale(row_ptr, col_ind, val, np.ones(n))
    lens = np.diff(row_ptr)
    want = ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_%s.txt" % REPORTS["memplus.mtx"][0]))
    well = np.abs(ref) > 1e-6 * scale
    for f, alg in zip(files, ("CSR", "TJDS")):
        text = open(tmp_path / f).read()
        assert "%s algorithm" % alg in text and "Compute times for 1000 iterations:" in text
        got = ob.report_y_lines(text)
        assert len(got) == m
        if alg == "CSR":
            assert all(got[i] == want[i] for i in np.flatnonzero(lens <= 32))
        assert sum(got[i] != want[i] for i in np.flatnonzero(well)) <= 2       # a last-digit rounding tie at most
        y = np.array([float(v) for v in got])
        assert np.all(np.abs(y - ref) <= 1e-5 * scale + 1e-300)                # "%g" keeps six digits


def test_cli_dump_arrays_and_random_operand(torch, tmp_path):
    """--dump-arrays prints the reference's debug dumps (main-cli.c:374-394, 458-466, 870-892, 969-992, 1150-1158,
    1166-1191): on ibm32 the reordering table and start_pos are the ones recorded from the reference itself;
    --x random multiplies by the documented random operand."""
    from test_oracle_golden import IBM32_PERM, IBM32_START_POS

    path = ob.fixture_path("ibm32.mtx")
    m, n, coo = load("ibm32.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    p = subprocess.run([sm.CLI_PATH, "-c", "-t", "--dump-arrays", "-n", "3", "-d", str(tmp_path), path], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    out = p.stdout

    def listed(head):
        i = out.index(head) + len(head)
        return [t for t in out[i:out.index("]", i)].replace("\n", " ").split(", ") if t.strip()]

    assert [int(t) for t in listed("[DEBUG]\tCSR JIT row_ptr:\n\t[")] == row_ptr.tolist()
    assert [int(t) for t in listed("[DEBUG]\tCSR JIT col_ind:\n\t[")] == col_ind.tolist()
    assert listed("[DEBUG]\tCSR JIT val:\n\t[") == ["%g" % v for v in val]
    want = ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_%s.txt" % REPORTS["ibm32.mtx"][0]))
    for head in ("[DEBUG]\tCSR JIT Vector Out:\n\t[", "[DEBUG]\tCSR Output Vector:\n\t[", "[DEBUG]\tTJDS PHASE 8: Output Vector:\n\t["):
        assert listed(head) == want, head
    assert "[DEBUG]\tCSR Iterations: 3\n[DEBUG]\tCSR fInputRows: 32\n[DEBUG]\tCSR fInputNonZeros: 126\n" in out
    assert len(listed("[DEBUG]\tCSR Times:\n\t[")) == 3
    assert [int(t) for t in listed("origCol\t[")] == IBM32_PERM
    assert [int(t) for t in listed("\tstart_pos:\t[")] == IBM32_START_POS
    assert "num_tjdiag (count, not 0-index):\t7" in out
    t = sm.tjds_from_coo(coo, m, n)
    assert [int(x) for x in listed("\trow_ind:\t[")] == t.row_ind.tolist()
    assert listed("\tval:\t\t[") == ["%g" % v for v in t.val]
    assert [int(x) for x in listed("colLen\t[")] == np.bincount(coo["col"], minlength=n)[t.perm].tolist()
    # the reference's own count with --ref-quirks (length of original column 0): start_pos over 6 + 1 entries
    q = subprocess.run([sm.CLI_PATH, "-t", "--ref-quirks", "--dump-arrays", "-n", "1", "-d", str(tmp_path), path],
                       capture_output=True, text=True).stdout
    assert "num_tjdiag (count, not 0-index):\t6" in q
    # without the flag nothing of it is printed
    quiet = subprocess.run([sm.CLI_PATH, "-c", "-n", "1", "-d", str(tmp_path), path], capture_output=True, text=True).stdout
    assert "[DEBUG]" not in quiet
    # --x random
    sub = tmp_path / "random_x"              # its own folder: report names carry the second they were written in
    sub.mkdir()
    r = subprocess.run([sm.CLI_PATH, "-c", "-t", "--x", "random", "-n", "2", "-d", str(sub), path], capture_output=True, text=True)
    assert r.returncode == 0 and "Random vector" in r.stdout
    x = sm.vector_random(n, 67890)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    new = sorted(os.listdir(sub))
    assert len(new) == 2
    for f in new:
        got = np.array([float(v) for v in ob.report_y_lines(open(sub / f).read())])
        assert np.all(np.abs(got - ref) <= 1e-5 * row_scale(row_ptr, col_ind, val, x))


def test_cli_ref_quirks_reproduces_reference_tjds_report(torch, tmp_path):
    name = "curtis54.mtx"
    p = subprocess.run([sm.CLI_PATH, "-t", "--ref-quirks", "-n", "5", "-d", str(tmp_path), ob.fixture_path(name)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    got = open(tmp_path / os.listdir(tmp_path)[0]).read()
    want = ob.read_report("smvp-toolbox_report_TJDS_%s.txt" % REPORTS[name][1])
    assert ob.report_y_lines(got) == ob.report_y_lines(want)


# ------------------------------------------------------------- BASELINE config 4
def test_uniform32_config4_shape(torch):
    """10 M x 10 M, 32 uniform entries per row (BASELINE config 4) at 1/10 size: oracle on a slice + row sums."""
    M = 1_000_000
    row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, M, M, param=32)
    x = np.random.default_rng(4).random(M)
    A = sm.CsrMatrix(M, M, row_ptr, col_ind, val)
    dx = dev(torch, x)
    scale = np.add.reduceat(np.abs(val * x[col_ind]), row_ptr[:-1])
    host = np.add.reduceat(val * x[col_ind], row_ptr[:-1])
    for kernel, param in ((sm.CSR_KERNEL_STREAM, 0), (sm.CSR_KERNEL_VECTOR, 32), (sm.CSR_KERNEL_VECTOR, 64)):
        A.set_kernel(kernel, param)
        dy = torch.full((M,), float("nan"), dtype=torch.float64, device="cuda")
        A.spmv(dx, dy)
        torch.cuda.synchronize()
        y = dy.cpu().numpy()
        assert np.all(np.abs(y - host) <= TOL * scale)
        k = 50_000
        ref = ob.csr_spmv(row_ptr[:k + 1].copy(), col_ind[:row_ptr[k]], val[:row_ptr[k]], x)
        assert np.all(np.abs(y[:k] - ref) <= TOL * scale[:k])
    A.close()


def _bench(tmp_path, argv, env=None, timeout=900):
    """bench.py as the driver runs it -> (the parsed LAST stdout line, its text, bench_detail.json)."""
    import json
    import sys

    from conftest import ROOT

    detail = str(tmp_path / "bench_detail.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv + ["--detail", detail], capture_output=True, text=True, env=env,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    out = p.stdout.splitlines()
    assert len(out) == 1 and out[0].startswith("{")      # ONE line and nothing else on stdout (library banners go to stderr)
    j = json.loads(out[-1])
    assert len(out[-1]) < 8000, len(out[-1])
    for obj in (j["roofline"], j["config"], j["cpu_baseline"] or {}):         # flat: nothing nested for a parser to drop
        assert all(not isinstance(v, (dict, list)) for v in obj.values())
    assert not [k for k in j["roofline"] if k.endswith("_error")], {k: v for k, v in j["roofline"].items() if k.endswith("_error")}
    assert "watchdog" not in j["roofline"] and "dropped_for_length" not in j["roofline"]
    return j, out[-1], json.load(open(detail))


def test_bench_script_runs_small(torch, tmp_path):
    """bench.py end to end at toy size: ONE compact JSON line (< 8000 characters, the last line of stdout) with the contract's keys,
    `roofline` and `cpu_baseline` as flat scalars; everything else in bench_detail.json."""
    j, text, d = _bench(tmp_path, ["--copies", "8", "--rows-log2", "16", "--rows", "300000", "--steps", "5", "--warmup", "2", "--cpu-iters", "2",
                                   "--no-live-traffic"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["dtype"] == "f64" and j["value"] > 0 and j["detail"] == "bench_detail.json"
    r, cpu = j["roofline"], j["cpu_baseline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert cpu["cores"] == 1 and cpu["kind"] == "port" and cpu["agrees_with_gpu"] and cpu["value"] > 0 and cpu["host_cpu"]
    assert cpu["gpu_rows_bit_identical_to_serial"] > 0.98 and r["y_equals_tiled_reference_y"] is True
    assert "substitute" in j["config"]["workload"]
    # every figure a record needs is a flat scalar of `roofline`
    for key in ("frac", "traffic", "frac_tjds", "frac_tjds_colmajor", "frac_survey_random_model", "frac_config4", "frac_pwt_csr", "frac_pwt_tjds",
                "ms_tjds", "ms_survey_random_model", "ms_config4",
                "config4_t1_ms", "config4_tN_step_ms", "config4_tN_step_after_ms", "config4_tN_products_only_ms",
                "config4_speedup_overlapped", "config4_speedup_after", "config4_speedup_products_only", "config4_chunks_chosen",
                "config4_eighth_ms_1chunk", "config4_eighth_ms_2chunk", "config4_eighth_ms_4chunk", "config4_G_gathers_per_s_per_gpu",
                "config4_frac_of_l2_gather_ceiling",
                "config4_c_layer_products_only_ms_1chunk", "config4_c_layer_overlapped_ms_1chunk", "config4_c_layer_after_ms_4chunk",
                "exchange_rccl_ms", "exchange_copies_ms", "exchange_direct_ms", "c_layer_exchange_chosen",
                "memplus_csr_us", "memplus_tjds_us", "memplus_csr_loop_wall_us", "pwt_csr_us", "config5_csr_us", "config5_both_us",
                "exchange", "dist_backend", "rccl_ranks", "n_gpus", "self_launched", "prewarm_ms", "leg_seconds", "wall_s"):
        assert key in r, key
    assert r["rccl_ranks"] == 0 and r["config4_speedup_overlapped"] == 1.0 and r["self_launched"] is False
    # the detail file: the same figures with what the line leaves out
    o, extra = d["others"], d["extra"]
    assert d["roofline"]["frac"] == r["frac"] and d["cpu_baseline"]["value"] == cpu["value"] and not d["errors"]
    assert extra["full_size_parity"]["y_equals_tiled_reference_memplus_y"]
    assert "error" not in extra["tjds"] and "error" not in extra["survey_random_model"]
    c4, pw = extra["config4"], extra["pwt_tiled"]
    assert c4["n_gpus"] == 1 and c4["rows"] == 300000 and c4["nnz"] == 32 * 300000 and c4["spmv_only_ms"] > 0
    assert pw["y_equals_tiled_reference_pwt_y"] and pw["tjds"]["equals_csr_bit_for_bit"]
    c5 = extra["config5_pwt"]
    assert c5["y_equals_reference_report"] and 0 < c5["csr_ms_per_step"] < c5["csr_then_tjds_ms_per_step"]
    sm_ = extra["sample_matrices"]["memplus.mtx"]
    assert sm_["csr_avg_ms"] < sm_["csr_avg_ms_event_pairs"] and sm_["csr_agrees_with_cpu"] and sm_["tjds_agrees_with_cpu"]
    for key in ("tjds", "config4", "config4_c_layer", "pwt_tiled_csr", "pwt_tiled_tjds", "survey_random_model", "sample_matrices_us_per_product"):
        assert key in o and "error" not in o[key], key
    assert r["frac_tjds"] == o["tjds"]["frac"] and r["frac_config4"] == o["config4"]["frac"] and r["config4_t1_ms"] == o["config4"]["t1_ms"]
    assert 0 < o["tjds"]["frac"] < 1 and 0 < o["config4"]["frac"] < 1 and o["config4"]["bit_identical_run_to_run"]
    cl = o["config4_c_layer"]
    assert cl["n_gpus"] == 1 and all(cl["chunks_%d" % c][f]["event_ms"] > 0 for c in (1, 4)
                                     for f in ("products_only", "products_then_allgather", "overlapped"))
    assert d["roofline"]["plan"]["plan_bytes"] > 0 and d["roofline"]["setup"]["device_arrays_equal_input"]
    assert o["config4"]["chunks_chosen_for_n8"] in (1, 2, 4) and set(o["config4"]["eighth_of_n8"]["estimates_ms"]) == {"1", "2", "4"}
    rm = o["survey_random_model"]
    assert rm["bit_identical_run_to_run"] and rm["plan"]["plan_bytes"] >= 0 and rm["launches_per_product"] >= 1


def test_bench_budget_skips_legs_and_keeps_the_line(torch, tmp_path):
    """--budget too small for the secondary legs: they are not started, each leaves `<leg>_error` on the line, the headline stands."""
    import json
    import sys

    from conftest import ROOT

    detail = str(tmp_path / "d.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--copies", "8", "--rows", "300000", "--steps", "3", "--warmup", "1",
                        "--no-live-traffic", "--budget", "1", "--detail", detail], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads(p.stdout.splitlines()[-1])
    r = j["roofline"]
    assert j["value"] > 0 and 0 < r["frac"] < 1 and j["cpu_baseline"] is None
    for leg in ("tjds", "cpu_baseline", "config4", "survey_random_model", "pwt_tiled"):
        assert r[leg + "_error"].startswith("skipped"), leg
    assert "frac_config4" not in r and len(p.stdout.splitlines()[-1]) < 8000


def test_bench_starts_its_own_ranks(torch, tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (the driver's command shape): the script starts its two ranks
    itself -- here over gloo, sharing the one GPU -- relays rank 0's compact line and leaves no process behind."""
    import psutil

    env = dict(os.environ, SMVP_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    before = {q.pid for q in psutil.process_iter()}
    j, text, d = _bench(tmp_path, ["--gpus", "2", "--copies", "8", "--rows", "300000", "--steps", "3", "--warmup", "1"], env=env)
    r = j["roofline"]
    assert j["n_gpus"] == 2 and r["n_gpus"] == 2 and r["self_launched"] is True and r["dist_backend"] == "gloo" and r["rccl_ranks"] == 0
    assert d["extra"]["dist"]["ranks_in_group"] == 2 and j["cpu_baseline"] is None
    for key in ("config4_t1_ms", "config4_tN_step_ms", "config4_tN_products_only_ms", "config4_speedup_overlapped",
                "config4_speedup_after", "config4_chunks_chosen", "headline_products_only_ms", "exchange"):
        assert key in r, key
    assert r["config4_c_layer_overlapped_ms_1chunk"] > 0 and r["exchange_direct_ms"] > 0
    left = [q for q in psutil.process_iter(["cmdline", "name"]) if q.pid not in before and "python" in (q.info["name"] or "")
            and "bench.py" in " ".join(q.info["cmdline"] or [])]      # (python processes only: a shell's command line may name bench.py too)
    assert not left, left


def test_device_primitives(torch, tmp_path):
    """smvp_prim.h -- the hand-written stable radix sort of (key, value) pairs and the prefix sums that the device-side converters
    and the plan builders stand on (rocPRIM until round 5) -- against std::stable_sort and running sums on the host: 64- and 32-bit
    keys, bit windows, few distinct keys (stability), sizes around every tile edge up to 20 M, in-place scans, three scan levels.
    tests/prim_check.hip is built by `make all` (or, when that binary is missing or stale, compiled here with hipcc)."""
    import shutil

    from conftest import ROOT

    src = [os.path.join(ROOT, "tests", "prim_check.hip"), os.path.join(ROOT, "smvp-toolkit_amd", "csrc", "smvp_prim.h")]
    exe = os.path.join(ROOT, "smvp-toolkit_amd", "bin", "prim_check")        # built by `make all` (__graft_entry__.build())
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(f) for f in src):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        exe = str(tmp_path / "prim_check")
        b = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "smvp-toolkit_amd", "csrc"),
                            src[0], "-o", exe], capture_output=True, text=True, timeout=600)
        assert b.returncode == 0, b.stderr[-2000:]
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "prim ok" in p.stdout, p.stdout[-2000:] + p.stderr[-1000:]


# --------------------------------------------------- device-side format conversion
def _coo_to_device(torch, coo):
    raw = np.ascontiguousarray(coo, dtype=sm.COO_DTYPE).view(np.uint8)
    if raw.size == 0:
        return torch.zeros(16, dtype=torch.uint8, device="cuda")
    return torch.from_numpy(raw.copy()).cuda()


def _check_device_conversion(torch, coo, rows, cols):
    d_coo = _coo_to_device(torch, coo)
    before = d_coo.clone()
    rp, ci, v = sm.csr_from_coo_device(d_coo, rows, cols, len(coo))
    hrp, hci, hv = sm.csr_from_coo(coo, rows)
    assert np.array_equal(rp.cpu().numpy(), hrp)
    assert np.array_equal(ci.cpu().numpy(), hci)
    assert v.cpu().numpy().tobytes() == hv.tobytes()
    t = sm.tjds_from_coo_device(d_coo, rows, cols, len(coo))
    h = sm.tjds_from_coo(coo, rows, cols)
    assert (t.num_diag, t.ref_num_tjdiag, t.last_diag_single) == (h.num_diag, h.ref_num_tjdiag, h.last_diag_single)
    for f in ("perm", "start_pos", "row_ind"):
        assert np.array_equal(getattr(t, f).cpu().numpy(), getattr(h, f)), f
    assert t.val.cpu().numpy().tobytes() == h.val.tobytes()
    assert torch.equal(d_coo, before)          # the input is not modified
    return t


@pytest.mark.parametrize("name", SAMPLES)
def test_device_conversion_sample_matrices(torch, name):
    m, n, coo = load(name)
    _check_device_conversion(torch, coo, m, n)


@pytest.mark.parametrize("seed", range(8))
def test_device_conversion_random(torch, seed):
    rng = np.random.default_rng(100 + seed)
    rows, cols = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    nnz = int(rng.integers(0, min(rows * cols, 5000) + 1))
    flat = rng.choice(rows * cols, size=nnz, replace=False)
    r, c = flat // cols, flat % cols
    if seed % 2:                                  # leave some rows and columns empty
        keep = ((r % 3) != 1) & ((c % 5) != 2)
        r, c = r[keep], c[keep]
    coo = sm.make_coo(r, c, rng.uniform(-1, 1, len(r)))
    _check_device_conversion(torch, coo, rows, cols)


def test_device_conversion_edge_cases(torch):
    _check_device_conversion(torch, sm.make_coo([], [], []), 5, 7)                       # no entries
    _check_device_conversion(torch, sm.make_coo([0], [0], [2.5]), 1, 1)
    _check_device_conversion(torch, sm.make_coo([3, 3, 3, 0], [0, 0, 0, 0], [1.0, 2.0, 3.0, 4.0]), 4, 2)  # duplicates keep input order
    with pytest.raises(sm.SmvpError):
        sm.csr_from_coo_device(_coo_to_device(torch, sm.make_coo([9], [0], [1.0])), 3, 3, 1)
    with pytest.raises(sm.SmvpError):
        sm.tjds_from_coo_device(_coo_to_device(torch, sm.make_coo([0], [-1], [1.0])), 3, 3, 1)


def test_device_conversion_large_and_product(torch):
    """2^20-row memplus-shaped matrix, entries shuffled: device-built CSR and TJDS feed the kernels directly."""
    M = 1 << 20
    row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 777, M, M)
    coo = sm.make_coo(np.repeat(np.arange(M), np.diff(row_ptr)), col_ind, val)
    coo = coo[np.random.default_rng(5).permutation(len(coo))]
    d_coo = _coo_to_device(torch, coo)
    rp, ci, v = sm.csr_from_coo_device(d_coo, M, M, len(coo))
    assert np.array_equal(rp.cpu().numpy(), row_ptr) and np.array_equal(ci.cpu().numpy(), col_ind)
    assert v.cpu().numpy().tobytes() == val.tobytes()
    A = sm.CsrMatrix(M, M, rp, ci, v)
    x = np.random.default_rng(6).random(M)
    dx = dev(torch, x)
    dy = torch.empty(M, dtype=torch.float64, device="cuda")
    A.spmv(dx, dy)
    torch.cuda.synchronize()
    scale = np.add.reduceat(np.abs(val * x[col_ind]), row_ptr[:-1])
    host = np.add.reduceat(val * x[col_ind], row_ptr[:-1])
    assert np.all(np.abs(dy.cpu().numpy() - host) <= TOL * scale)
    t = sm.tjds_from_coo_device(d_coo, M, M, len(coo))
    h = sm.tjds_from_coo(coo, M, M)
    for f in ("perm", "start_pos", "row_ind"):
        assert np.array_equal(getattr(t, f).cpu().numpy(), getattr(h, f)), f
    assert t.val.cpu().numpy().tobytes() == h.val.tobytes()


@pytest.mark.parametrize("name", ["ibm32.mtx", "memplus.mtx", "pwt.mtx"])
def test_entry_points_with_device_conversion(torch, name):
    m, n, coo = load(name)
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    ref = ob.csr_spmv(row_ptr, col_ind, val, np.ones(n))
    scale = row_scale(row_ptr, col_ind, val, np.ones(n))
    y, ms, st = sm.csr_compute(coo, m, n, iters=3, device_convert=True)
    assert_close(y, ref, scale, exact=name in EXACT)
    y, ms, st = sm.tjds_compute(coo, m, n, iters=3, device_convert=True)
    assert_close(y, ref, scale, exact=name in EXACT)
    y, ms, st = sm.tjds_compute(coo, m, n, iters=1, device_convert=True, ref_quirks=True)
    assert_close(y, ob.tjds_spmv(ob.tjds_build(coo, m, n), np.ones(n), refquirks=True), scale, exact=name in EXACT)


def test_cli_device_convert_flag(torch, tmp_path):
    p = subprocess.run([sm.CLI_PATH, "--all-algs", "--device-convert", "-n", "3", "-d", str(tmp_path),
                        ob.fixture_path("curtis54.mtx")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    want = ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_1615284695.txt"))
    for f in os.listdir(tmp_path):
        assert ob.report_y_lines(open(tmp_path / f).read()) == want


# --------------------------------------------------- several GPUs from one process (RCCL)
def _gpu_counts():
    try:
        import torch as _t
        n = _t.cuda.device_count()
    except Exception:
        n = 1
    return [g for g in (1, 2, 4, 8) if g <= max(n, 1)]


@pytest.mark.parametrize("ngpus", _gpu_counts())
@pytest.mark.parametrize("name", ["ibm32.mtx", "memplus.mtx", "pwt.mtx"])
def test_single_process_sharded_products(torch, name, ngpus):
    """Row blocks + ncclAllGather through the C ABI (smvp_sharded_*); on a one-GPU box this runs with one block."""
    m, n, coo = load(name)
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    x = np.random.default_rng(8).random(n)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    scale = row_scale(row_ptr, col_ind, val, x)
    for fmt in ("csr", "tjds"):
        S = sm.ShardedMatrix(fmt, ngpus, m, n, coo=coo, csr=(row_ptr, col_ind, val))
        assert S.info()[0] == ngpus and S.info()[1] * ngpus >= m
        S.set_x(x)
        for _ in range(3):
            S.spmv(allgather=True, timed=True)
            ms = S.synchronize()
        assert ms > 0
        for slot in range(ngpus):                      # every GPU holds the whole gathered vector
            assert_close(S.get_y(slot, gathered=True), ref, scale)
        assert_close(S.get_y(0, gathered=False), ref, scale)
        S.set_x(None)                                  # the reference's operand: ones
        S.spmv()
        S.synchronize()
        assert_close(S.get_y(), ob.csr_spmv(row_ptr, col_ind, val, np.ones(n)),
                     row_scale(row_ptr, col_ind, val, np.ones(n)), exact=name in EXACT)
        S.close()


@pytest.mark.parametrize("chunks,balance", [(1, True), (3, True), (4, False), (7, True)])
@pytest.mark.parametrize("gather", [sm.GATHER_OVERLAPPED, sm.GATHER_AFTER])
def test_sharded_chunks_and_both_exchange_forms(torch, chunks, balance, gather):
    """Row chunks inside a block (each its own handle), entry-balanced bounds padded on the wire, the gathered pieces
    put back in row order; overlapped and plain exchange.  One GPU here: the chunk / padding / placement logic is
    what is exercised (the N > 1 exchange itself has only ever run in the gloo tests of the Python layer)."""
    m, n, coo = load("memplus.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    x = np.random.default_rng(3).random(n)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    scale = row_scale(row_ptr, col_ind, val, x)
    for fmt in ("csr", "tjds"):
        S = sm.ShardedMatrix(fmt, 1, m, n, coo=coo, csr=(row_ptr, col_ind, val), chunks=chunks, balance=balance)
        c, bounds, cb = S.layout()
        assert c == chunks and bounds.tolist() == [0, m] and cb[0, 0] == 0 and cb[0, -1] == m and np.all(np.diff(cb[0]) >= 0)
        if balance and chunks > 1:      # memplus: balanced by entries, so the chunks are of unequal height
            per = np.diff(row_ptr[cb[0]])
            assert per.max() <= 1.3 * per.mean() and len(set(np.diff(cb[0]).tolist())) > 1
        S.set_x(x)
        for _ in range(2):
            S.spmv(allgather=gather, timed=True)
            assert S.synchronize() > 0
        assert_close(S.get_y(0, gathered=True), ref, scale)
        assert_close(S.get_y(0, gathered=False), ref, scale)
        if fmt == "csr":                                   # another kernel family on every chunk
            S.set_csr_kernel(sm.CSR_KERNEL_COLSWEEP, 1024)
            S.spmv(allgather=gather)
            S.synchronize()
            assert_close(S.get_y(0, gathered=True), ref, scale)
        else:
            with pytest.raises(sm.SmvpError):
                S.set_csr_kernel(sm.CSR_KERNEL_STREAM, 1024)
        S.close()


def test_sharded_more_chunks_than_rows_and_tiny_matrices(torch):
    m, n, coo = load("pdp08-pg4.mtx")                      # 6 x 6
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    ref = ob.csr_spmv(row_ptr, col_ind, val, np.ones(n))
    for fmt in ("csr", "tjds"):
        S = sm.ShardedMatrix(fmt, 1, m, n, coo=coo, csr=(row_ptr, col_ind, val), chunks=16)
        S.set_x(None)
        S.spmv()
        S.synchronize()
        assert np.array_equal(S.get_y(), ref)
        S.close()


@pytest.mark.skipif(len(_gpu_counts()) < 2, reason="needs at least two GPUs")
@pytest.mark.parametrize("ngpus", [g for g in _gpu_counts() if g > 1])
def test_sharded_unequal_blocks_on_several_gpus(torch, ngpus):
    """rows % ngpus != 0, more GPUs than rows, every slot's gathered y against the host (runs only where N > 1 GPUs exist)."""
    rng = np.random.default_rng(ngpus)
    for rows, cols in ((1003, 997), (max(1, ngpus - 1), 5)):
        lens = rng.integers(0, 9, rows).tolist()
        lens = [min(l, cols) for l in lens]
        row_ptr, col_ind, val = csr_from_lengths(rng, lens, cols)
        coo = sm.make_coo(np.repeat(np.arange(rows), lens), col_ind, val)
        x = rng.random(cols)
        ref = ob.csr_spmv(row_ptr, col_ind, val, x)
        scale = row_scale(row_ptr, col_ind, val, x)
        for fmt in ("csr", "tjds"):
            got = {}
            for gather in (sm.GATHER_OVERLAPPED, sm.GATHER_AFTER):
                S = sm.ShardedMatrix(fmt, ngpus, rows, cols, coo=coo, csr=(row_ptr, col_ind, val), chunks=3)
                S.set_x(x)
                S.spmv(allgather=gather)
                S.synchronize()
                for slot in range(ngpus):
                    assert_close(S.get_y(slot, gathered=True), ref, scale)
                got[gather] = S.get_y(0, gathered=True)
                S.close()
            assert np.array_equal(got[sm.GATHER_OVERLAPPED], got[sm.GATHER_AFTER])    # the same bits either way


@pytest.mark.parametrize("push", ["copies", "direct"])
@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_sharded_virtual_ranks_on_one_gpu(torch, ranks, push):
    """The N-GPU code of the C layer with N ranks sharing this box's one GPU (SMVP_EXCHANGE_COPIES / _DIRECT: every rank
    pushes its chunks straight into every rank's full vector, by hipMemcpyAsync or by one kernel per chunk, ordered by events
    and a meeting point of the issuing threads): rows % N != 0, more ranks than rows, empty chunks, every rank's
    gathered y against the oracle, both gather modes bit-equal, CSR and TJDS, several products back to back (the full
    vectors are written again), power iteration, the column sweep on zero-row chunks, an early destroy."""
    rng = np.random.default_rng(100 + ranks)
    devices = [0] * ranks
    PUSH = sm.EXCHANGE_COPIES if push == "copies" else sm.EXCHANGE_DIRECT
    OTHER = sm.EXCHANGE_DIRECT if push == "copies" else sm.EXCHANGE_COPIES
    for rows, cols, chunks in ((1003, 997, 3), (max(1, ranks - 1), 5, 2), (257, 257, 1)):
        lens = [min(int(l), cols) for l in rng.integers(0, 9, rows)]
        row_ptr, col_ind, val = csr_from_lengths(rng, lens, cols)
        coo = sm.make_coo(np.repeat(np.arange(rows), lens), col_ind, val)
        x = rng.random(cols)
        ref = ob.csr_spmv(row_ptr, col_ind, val, x)
        scale = row_scale(row_ptr, col_ind, val, x)
        for fmt in ("csr", "tjds"):
            got = {}
            for gather in (sm.GATHER_OVERLAPPED, sm.GATHER_AFTER):
                S = sm.ShardedMatrix(fmt, ranks, rows, cols, coo=coo, csr=(row_ptr, col_ind, val), devices=devices, chunks=chunks,
                                     exchange=PUSH)
                n, tallest = S.info()
                c, bounds, cb = S.layout()
                assert n == ranks and c == chunks and bounds[0] == 0 and bounds[-1] == rows and np.all(np.diff(bounds) >= 0)
                S.set_x(x)
                for _ in range(3):                      # back to back: the wire buffers are written again
                    S.spmv(allgather=gather)
                S.synchronize()
                for slot in range(ranks):
                    assert_close(S.get_y(slot, gathered=True), ref, scale)
                assert_close(S.get_y(0, gathered=False), ref, scale)
                got[gather] = S.get_y(ranks - 1, gathered=True)
                info = S.exchange_info()
                assert info["active"] == PUSH and set(info["available"]) == {sm.EXCHANGE_COPIES, sm.EXCHANGE_DIRECT} and info["rccl_ranks"] == 0
                if gather == sm.GATHER_OVERLAPPED:       # the other push form on the same handle: the same bits
                    S.set_exchange(OTHER)
                    S.spmv(allgather=gather)
                    S.synchronize()
                    assert np.array_equal(S.get_y(0, gathered=True), got[gather])
                    with pytest.raises(sm.SmvpError):
                        S.set_exchange(sm.EXCHANGE_RCCL)     # ranks share a device: no communicator
                    S.set_exchange(PUSH)
                    probe = S.probe_exchange(2)              # both forms timed; an explicitly chosen form stays
                    assert probe["active"] == PUSH and probe["ms"]["copies"] > 0 and probe["ms"]["direct"] > 0 and "rccl" not in probe["ms"]
                if fmt == "csr" and gather == sm.GATHER_AFTER:
                    S.set_csr_kernel(sm.CSR_KERNEL_COLSWEEP, 1024)      # zero-row chunks included
                    S.spmv(allgather=gather)
                    S.synchronize()
                    assert_close(S.get_y(0, gathered=True), ref, scale)
                S.close()
            assert np.array_equal(got[sm.GATHER_OVERLAPPED], got[sm.GATHER_AFTER])    # the same bits either way
    # power iteration: the gathered y is the next operand on every rank
    m, n, coo = load("ibm32.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    S = sm.ShardedMatrix("csr", ranks, m, n, csr=(row_ptr, col_ind, val), devices=devices, chunks=2, exchange=PUSH)
    S.set_x(None)
    v = np.ones(n)
    for _ in range(4):
        S.spmv(allgather=sm.GATHER_OVERLAPPED)
        S.feed_back(normalize=False)
        v = ob.csr_spmv(row_ptr, col_ind, val, v)
    S.synchronize()
    assert np.array_equal(S.get_y(ranks - 1, gathered=True), v)        # pattern matrix: exact
    S.close()
    # early destroy with work in flight; and through the reference-shaped entry points
    S = sm.ShardedMatrix("tjds", ranks, m, n, coo=coo, devices=devices, exchange=PUSH)
    S.set_x(None)
    S.spmv(allgather=sm.GATHER_OVERLAPPED)
    S.close()
    y1, ms, st = sm.csr_compute(coo, m, n, iters=5, ngpus=ranks, exchange=PUSH)
    y2, _, _ = sm.tjds_compute(coo, m, n, iters=5, ngpus=ranks, exchange=PUSH)
    want = ob.csr_spmv(row_ptr, col_ind, val, np.ones(n))
    assert np.array_equal(y1, want) and np.array_equal(y2, want) and len(ms) == 5 and st.time_min > 0
    # RCCL cannot put two ranks on one device, and AUTO does not guess that virtual ranks are wanted: refused, not hung
    for ex in (sm.EXCHANGE_RCCL, sm.EXCHANGE_AUTO):
        with pytest.raises(sm.SmvpError):
            sm.ShardedMatrix("csr", 2, m, n, csr=(row_ptr, col_ind, val), devices=[0, 0], exchange=ex)


def test_sharded_exchange_auto_is_a_measured_choice(torch):
    """SMVP_EXCHANGE_AUTO (the default): every available form -- RCCL, peer copies, the push kernel -- moves one product's
    y when the handle is created, the fastest is kept, the times are reported, and every form gives the same bits
    (one GPU here: the choice itself only means something on eight)."""
    m, n, coo = load("memplus.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    ref = ob.csr_spmv(row_ptr, col_ind, val, np.ones(n))
    S = sm.ShardedMatrix("csr", 1, m, n, csr=(row_ptr, col_ind, val), chunks=3)      # exchange: AUTO
    info = S.exchange_info()
    assert set(info["available"]) == {sm.EXCHANGE_RCCL, sm.EXCHANGE_COPIES, sm.EXCHANGE_DIRECT} and info["rccl_ranks"] == 1
    assert set(info["ms"]) == {"rccl", "copies", "direct"} and all(v > 0 for v in info["ms"].values())
    # the fastest is kept -- unless it beats the starting form (RCCL) by less than 5 %: then AUTO stays where it started
    fastest = min(info["ms"].values())
    assert info["ms"][info["active_name"]] == fastest or (info["active"] == sm.EXCHANGE_RCCL and fastest >= 0.95 * info["ms"]["rccl"])
    S.set_x(None)
    got = {}
    for ex in (sm.EXCHANGE_RCCL, sm.EXCHANGE_COPIES, sm.EXCHANGE_DIRECT):
        S.set_exchange(ex)
        for gather in (sm.GATHER_OVERLAPPED, sm.GATHER_AFTER):
            S.spmv(allgather=gather)
            S.synchronize()
            got[ex, gather] = S.get_y(0, gathered=True)
    first = got[sm.EXCHANGE_RCCL, sm.GATHER_OVERLAPPED]
    assert np.allclose(first, ref, rtol=1e-12, atol=1e-12) and all(np.array_equal(first, g) for g in got.values())
    with pytest.raises(sm.SmvpError):
        S.set_exchange(sm.EXCHANGE_AUTO)
    S.close()
    # a caller built against another layout of the options struct is refused, not misread
    import ctypes as C
    o = sm.ShardOpts()
    sm.lib().smvp_shard_opts_default(C.byref(o))
    assert o.struct_size == C.sizeof(sm.ShardOpts) and o.exchange == sm.EXCHANGE_AUTO
    o.struct_size -= 4
    h = C.c_void_p()
    rc = sm.lib().smvp_csr_sharded_create_ex(C.byref(h), 1, None, m, n, len(coo), row_ptr.ctypes.data_as(C.c_void_p),
                                             col_ind.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p), C.byref(o))
    assert rc != 0 and b"smvp_shard_opts_default" in sm.lib().smvp_last_error()
    ro = sm.RunOpts()                      # never passed through smvp_run_opts_default
    y = np.zeros(m)
    rc = sm.lib().smvp_csr_compute(coo.ctypes.data_as(C.c_void_p), m, n, len(coo), 1, C.byref(ro), y.ctypes.data_as(C.c_void_p), None, None)
    assert rc != 0 and b"smvp_run_opts_default" in sm.lib().smvp_last_error()


@pytest.mark.parametrize("chunks", [1, 4])
def test_sharded_eight_virtual_ranks_at_config4_scale(torch, chunks):
    """BASELINE config 4's shape at half its size -- 5 M x 5 M, 32 uniform entries per row, seed 2024 -- over EIGHT virtual
    ranks (the N = 8 code of the C layer on this box's one GPU), 1 and 4 chunks per rank: AUTO gives every chunk the
    column sweep (its own plan over all of x, zero-padded `woff` / chunk bounds at 10^5-row scale), both gather modes and
    both push forms give the same bits on every rank, y equals the row sums of val on ALL rows (x = ones) and the oracle
    on a row sample (x random)."""
    rows, ranks = 5_000_000, 8
    rp, ci, v = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, rows, rows, 32, 0, rows, threads=16)
    assert int(rp[-1]) == 32 * rows
    S = sm.ShardedMatrix("csr", ranks, rows, rows, csr=(rp, ci, v), devices=[0] * ranks, chunks=chunks, exchange=sm.EXCHANGE_DIRECT)
    c, bounds, cb = S.layout()
    assert c == chunks and np.all(np.abs(np.diff(bounds) - rows // ranks) <= 1) and cb[-1][-1] == rows
    # x = ones (the reference's operand): every row's sum of val, all rows, every rank's copy
    host = np.add.reduceat(v, rp[:-1])
    scale = np.add.reduceat(np.abs(v), rp[:-1])
    S.set_x(None)
    got = {}
    for ex in (sm.EXCHANGE_DIRECT, sm.EXCHANGE_COPIES):
        S.set_exchange(ex)
        for gather in (sm.GATHER_OVERLAPPED, sm.GATHER_AFTER):
            S.spmv(allgather=gather)
            S.synchronize()
            got[ex, gather] = S.get_y(ranks - 1, gathered=True)
    first = got[sm.EXCHANGE_DIRECT, sm.GATHER_OVERLAPPED]
    assert np.all(np.abs(first - host) <= TOL * scale)
    assert all(np.array_equal(first, g) for g in got.values())
    for slot in (0, 3):
        assert np.array_equal(S.get_y(slot, gathered=True), first)
    assert np.array_equal(S.get_y(0, gathered=False), first)              # the slices as the ranks hold them
    # a general operand: the oracle on samples of rows at the seams between ranks and chunks
    x = sm.vector_random(rows)
    S.set_x(x)
    S.spmv(allgather=sm.GATHER_OVERLAPPED)
    S.synchronize()
    y = S.get_y(5, gathered=True)
    for a in (0, int(cb[0][1]) - 2000 if chunks > 1 else 300_000, int(bounds[1]) - 2000, int(bounds[7]) - 2000, rows - 4000):
        a = max(0, a)
        b = min(rows, a + 4000)
        ref = ob.csr_spmv((rp[a:b + 1] - rp[a]).astype(np.int32), ci[rp[a]:rp[b]], v[rp[a]:rp[b]], x)
        sc = ob.csr_spmv((rp[a:b + 1] - rp[a]).astype(np.int32), ci[rp[a]:rp[b]], np.abs(v[rp[a]:rp[b]]), np.abs(x))
        assert np.array_equal(y[a:b], ref) or np.all(np.abs(y[a:b] - ref) <= TOL * sc)
        # 32 entries per row, ascending columns: the sweep and the tile kernel both sum in serial order -- the oracle's bits
        assert np.array_equal(y[a:b], ref)
    S.close()
    # what AUTO gave a chunk of this size: the column sweep
    a, b = int(cb[2][0]), int(cb[2][1])
    A = sm.CsrMatrix(b - a, rows, (rp[a:b + 1] - rp[a]).astype(np.int32), ci[rp[a]:rp[b]], v[rp[a]:rp[b]], first_row=a)
    assert A.get_kernel()[0] == sm.CSR_KERNEL_COLSWEEP
    A.close()


def test_sharded_eight_virtual_ranks_config5_pwt(torch):
    """BASELINE config 5 over eight virtual ranks: pwt.mtx as stored, CSR then TJDS back to back, row blocks balanced by
    entries; every rank's gathered y equals the reference's committed report, both formats, both push forms."""
    m, n, coo = load("pwt.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    want = np.array([float(t) for t in ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_%s.txt" % REPORTS["pwt.mtx"][0]))])
    ranks = 8
    for ex in (sm.EXCHANGE_DIRECT, sm.EXCHANGE_COPIES):
        C_ = sm.ShardedMatrix("csr", ranks, m, n, csr=(row_ptr, col_ind, val), devices=[0] * ranks, chunks=2, exchange=ex)
        T_ = sm.ShardedMatrix("tjds", ranks, m, n, coo=coo, devices=[0] * ranks, chunks=2, exchange=ex)
        C_.set_x(None)
        T_.set_x(None)
        for _ in range(3):                  # CSR then TJDS, back to back
            C_.spmv(allgather=sm.GATHER_OVERLAPPED)
            T_.spmv(allgather=sm.GATHER_OVERLAPPED)
        C_.synchronize()
        T_.synchronize()
        for slot in (0, 4, 7):
            assert np.array_equal(C_.get_y(slot, gathered=True), want)
            assert np.array_equal(T_.get_y(slot, gathered=True), want)
        _, bounds, _ = C_.layout()
        cost = 12.0 * np.diff(row_ptr[bounds]) + 20.0 * np.diff(bounds)   # smvp_partition_rows: algorithmic bytes, 12 per entry + 20 per row
        assert cost.max() - cost.min() <= 2 * (12.0 * np.diff(row_ptr).max() + 20.0)
        C_.close()
        T_.close()


@pytest.mark.parametrize("exchange", [None, "copies", "direct"])
def test_cli_virtual_gpus(torch, tmp_path, exchange):
    """--gpus 4 --virtual-gpus [--exchange copies|direct] on the one-GPU box: the reports equal the committed ones."""
    out = tmp_path / "virt"
    out.mkdir()
    cmd = [sm.CLI_PATH, "-c", "-t", "-n", "20", "--gpus", "4", "--virtual-gpus"] + (["--exchange", exchange] if exchange else [])
    p = subprocess.run(cmd + ["-d", str(out), ob.fixture_path("ibm32.mtx")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "4 virtual GPUs" in p.stdout and ("peer copies" if exchange == "copies" else "peer pushes") in p.stdout
    for alg, stamp in (("CSR", REPORTS["ibm32.mtx"][0]), ("TJDS", REPORTS["ibm32.mtx"][1])):
        files = [f for f in os.listdir(out) if "_%s_" % alg in f]
        assert len(files) == 1
        got = ob.report_y_lines(open(os.path.join(out, files[0])).read())
        assert got == ob.report_y_lines(ob.read_report("smvp-toolbox_report_%s_%s.txt" % (alg, stamp)))


def test_cli_exchange_and_timing_flags(torch, tmp_path):
    """--exchange takes auto | rccl | copies | direct (one GPU: every form is available) and refuses anything else; --timing
    device-graph asks for one launch per product; the y blocks of the reports are the committed ones every way."""
    want = ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_%s.txt" % REPORTS["ibm32.mtx"][0]))
    for k, flags in enumerate((["--gpus", "1", "--exchange", "rccl"], ["--exchange", "auto"], ["--timing", "device-graph"], ["--timing", "device"])):
        out = tmp_path / ("o%d" % k)
        out.mkdir()
        p = subprocess.run([sm.CLI_PATH, "-c", "-n", "30"] + flags + ["-d", str(out), ob.fixture_path("ibm32.mtx")],
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        f = [x for x in os.listdir(out) if "_CSR_" in x]
        assert len(f) == 1 and ob.report_y_lines(open(os.path.join(out, f[0])).read()) == want
        if flags == ["--timing", "device-graph"]:
            assert "replayed from a hipGraph" in p.stdout
        if flags == ["--timing", "device"]:
            assert "repeating kernel" in p.stdout
    for bad in (["--exchange", "ring"], ["--timing", "gpu"]):
        p = subprocess.run([sm.CLI_PATH, "-c", "-n", "3"] + bad + ["-d", str(tmp_path), ob.fixture_path("ibm32.mtx")],
                           capture_output=True, text=True, timeout=120)
        assert p.returncode == 1 and "[ERROR]" in p.stdout + p.stderr


def test_sharded_issuing_thread_machinery_on_one_gpu(torch, tmp_path):
    """With several GPUs every GPU's launches and collectives are issued by its own thread (woken per product, joined at
    destroy).  The plan option "sharded_threads" = 1 runs that machinery with the one GPU of this box: many products, both exchange
    forms, power iteration, an early destroy -- same results as the caller's-thread path."""
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import smvp_toolkit_amd as sm, oracle_binding as ob
tc, m, n, coo = sm.mm_read_coo(ob.fixture_path("memplus.mtx"))
rp, ci, v = sm.csr_from_coo(coo, m)
ref = ob.csr_spmv(rp, ci, v, np.ones(n))
sm.set_option("sharded_threads", 1)
for fmt in ("csr", "tjds"):
    S = sm.ShardedMatrix(fmt, 1, m, n, coo=coo, csr=(rp, ci, v), chunks=4)
    S.set_x(None)
    for it in range(200):
        S.spmv(allgather=(sm.GATHER_OVERLAPPED, sm.GATHER_AFTER, sm.GATHER_NONE)[it %% 3])
        if it %% 50 == 0:
            S.synchronize()
    S.spmv(allgather=sm.GATHER_OVERLAPPED)
    ms = S.synchronize()
    y = S.get_y(0, gathered=True)
    assert ms > 0 and np.allclose(y, ref, rtol=1e-12, atol=1e-12), fmt
    S.feed_back(normalize=True)
    S.spmv()
    S.close()
S = sm.ShardedMatrix("csr", 1, m, n, csr=(rp, ci, v), chunks=2)       # destroyed without a product: no thread was started
S.close()
y, ms, st = sm.csr_compute(coo, m, n, iters=20)
assert np.allclose(y, ref, rtol=1e-12, atol=1e-12)
print("threads ok")
""" % (os.path.join(os.path.dirname(sm.LIB_PATH), "..", "python"), os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "threads ok" in p.stdout, p.stdout + p.stderr


def test_device_timing_is_refused_for_a_sharded_run(torch):
    """The in-kernel stamps time one launch on one GPU; a sharded product is several launches on several GPUs: asking for
    them is an error, not silently something else (the check comes before any GPU is opened)."""
    m, n, coo = load("ibm32.mtx")
    for fn in (sm.csr_compute, sm.tjds_compute):
        with pytest.raises(sm.SmvpError) as e:
            fn(coo, m, n, iters=2, ngpus=2, timing=sm.TIMING_DEVICE)
        assert e.value.code == sm.ERR_UNSUPPORTED


def test_calls_leave_the_callers_device_alone(torch):
    """Every entry point that touches a device runs on the handle's device and restores the caller's (hipGetDevice
    unchanged); with one GPU this can only check that nothing breaks, with several it is the real test."""
    m, n, coo = load("ibm32.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    last = torch.cuda.device_count() - 1
    before = torch.cuda.current_device()
    A = sm.CsrMatrix(m, n, row_ptr, col_ind, val, device=last)
    A.set_kernel(sm.CSR_KERNEL_STREAM, 256)
    T = sm.TjdsMatrix(sm.tjds_from_coo(coo, m, n), device=last)
    T.set_mode(sm.TJDS_MODE_TWO_PHASE)
    T.set_ref_quirks(True)
    S = sm.ShardedMatrix("csr", torch.cuda.device_count(), m, n, csr=(row_ptr, col_ind, val))
    S.set_x(None)
    S.spmv()
    S.synchronize()
    S.close()
    sm.csr_compute(coo, m, n, iters=2, device=last)
    sm.tjds_compute(coo, m, n, iters=2, device=last)
    A.close()
    T.close()
    assert torch.cuda.current_device() == before
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    dev = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(dev)) == 0 and dev.value == before


def test_set_kernel_rejects_tile_sizes_the_resolved_kernel_lacks(torch):
    m, n, coo = load("ibm32.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    A = sm.CsrMatrix(m, n, row_ptr, col_ind, val)
    for bad in (512, 100, 4096):
        with pytest.raises(sm.SmvpError):
            A.set_kernel(sm.CSR_KERNEL_AUTO, bad)          # AUTO resolves to the stream kernel: 256 / 1024 / 2048 only
    with pytest.raises(sm.SmvpError):
        A.set_kernel(sm.CSR_KERNEL_STREAM_CARRY, 256)
    A.set_kernel(sm.CSR_KERNEL_AUTO, 2048)
    assert A.get_kernel() == (sm.CSR_KERNEL_STREAM, 2048)
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    y = torch.empty(m, dtype=torch.float64, device="cuda")
    A.spmv(x, y)                                            # still launchable
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), ob.csr_spmv(row_ptr, col_ind, val, np.ones(n)))


def test_expanded_symmetric_product_on_the_gpu(torch, tmp_path):
    """--expand-symmetric (not the reference's behaviour): A_full . 1 for pwt, CSR and TJDS, against numpy on the
    fixture; and through the CLI, whose default run must still print the reference's stored-triangle report."""
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path("pwt.mtx"))
    full = sm.mm_expand_symmetric(tc, coo, m, n)
    want = np.zeros(m)
    np.add.at(want, coo["row"], coo["val"])
    off = coo["row"] != coo["col"]
    np.add.at(want, coo["col"][off], coo["val"][off])
    y, _, _ = sm.csr_compute(full, m, n, iters=2)
    assert np.array_equal(y, want)                     # 0/1 matrix: small integers, exact in any order
    y, _, _ = sm.tjds_compute(full, m, n, iters=2)
    assert np.array_equal(y, want)
    p = subprocess.run([sm.CLI_PATH, "-c", "-n", "3", "-d", str(tmp_path), "--expand-symmetric", "--cache",
                        ob.fixture_path("pwt.mtx")], capture_output=True, text=True, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-1500:]
    rep = [f for f in os.listdir(tmp_path) if f.startswith("smvp-toolbox_report_CSR_")]
    lines = open(os.path.join(tmp_path, rep[0])).read().split("\n")
    assert lines[lines.index("[") + 1:lines.index("]")] == ob.fmt_g(want)
    assert "Non-zero numbers contained in matrix: 326107" in lines
    try:
        os.remove(ob.fixture_path("pwt.mtx") + ".smvpbin")
    except OSError:
        pass


def test_sharded_rejects_bad_gpu_counts(torch):
    m, n, coo = load("ibm32.mtx")
    csr = sm.csr_from_coo(coo, m)
    with pytest.raises(sm.SmvpError):
        sm.ShardedMatrix("csr", torch.cuda.device_count() + 1, m, n, csr=csr)
    with pytest.raises(sm.SmvpError):
        sm.ShardedMatrix("csr", 0, m, n, csr=csr)
    if torch.cuda.device_count() >= 2:
        y, ms, st = sm.csr_compute(coo, m, n, iters=5, ngpus=2)
        assert ob.fmt_g(y) == ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_1615284655.txt"))
        y, ms, st = sm.tjds_compute(coo, m, n, iters=5, ngpus=2)
        assert ob.fmt_g(y) == ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_1615284655.txt"))


def test_bench_script_tjds_format(torch, tmp_path):
    j, text, d = _bench(tmp_path, ["--format", "tjds", "--copies", "8", "--steps", "3", "--warmup", "1", "--no-random-model", "--no-samples",
                                   "--no-cpu-baseline", "--no-config4", "--no-pwt-tiled"])
    assert j["config"]["format"] == "tjds" and "TJDS" in j["metric"] and j["roofline"]["kernel"] == "csr_stream_owner<4, 4, false>"
    assert d["extra"]["full_size_parity"]["y_equals_tiled_reference_memplus_y"] and j["value"] > 0
    # roofline.traffic of this line was measured in the run itself (rocprofv3 --pmc child passes) unless rocprofv3 is
    # missing; either way it can only lie between the algorithmic bytes and a few times them
    t = j["roofline"]["traffic"]
    if t is not None:
        assert j["roofline"]["alg_bytes_per_launch"] * 0.9 <= t <= j["roofline"]["alg_bytes_per_launch"] * 6
        assert "measured in this run" in j["roofline"]["traffic_source"] or "profiles/" in j["roofline"]["traffic_source"]


# ------------------------------------------------------------- power iteration (parity unpinned: no reference output)
@pytest.mark.parametrize("name", ["ibm32.mtx", "curtis54.mtx", "pwt.mtx"])
def test_power_iteration_exact_on_pattern_matrices(torch, name):
    """x <- A x three times on 0/1 matrices: every iterate is a vector of small integers, so any order is exact."""
    m, n, coo = load(name)
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    want = ob.csr_iterate(row_ptr, col_ind, val, np.ones(n), 3)
    y, ms, st = sm.csr_compute(coo, m, n, iters=3, iterate=True)
    assert np.array_equal(y, want) and len(ms) == 3
    y, ms, st = sm.tjds_compute(coo, m, n, iters=3, iterate=True)
    assert np.array_equal(y, want)
    # one iteration is the plain product
    y, _, _ = sm.csr_compute(coo, m, n, iters=1, iterate=True)
    assert np.array_equal(y, ob.csr_spmv(row_ptr, col_ind, val, np.ones(n)))


def test_power_iteration_normalised_memplus(torch):
    m, n, coo = load("memplus.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    for iters in (1, 2, 5, 40):
        want = ob.csr_iterate(row_ptr, col_ind, val, np.ones(n), iters, normalize=True)
        for fn in (sm.csr_compute, sm.tjds_compute):
            y, ms, st = fn(coo, m, n, iters=iters, iterate=True, normalize=True)
            assert np.abs(y).max() == pytest.approx(1.0, abs=1e-12)
            assert np.abs(y - want).max() <= 1e-9 * iters       # max-norm of both is 1
    with pytest.raises(sm.SmvpError):                            # rectangular: no power iteration
        sm.csr_compute(sm.make_coo([0], [1], [1.0]), 1, 2, iters=2, iterate=True)


def test_sharded_power_iteration(torch):
    m, n, coo = load("memplus.mtx")
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    want = ob.csr_iterate(row_ptr, col_ind, val, np.ones(n), 4, normalize=True)
    for fmt in ("csr", "tjds"):
        S = sm.ShardedMatrix(fmt, 1, m, n, coo=coo, csr=(row_ptr, col_ind, val))
        S.set_x(None)
        for _ in range(4):
            S.spmv(allgather=True, timed=True)
            S.synchronize()
            S.feed_back(normalize=True)
        S.synchronize()
        assert np.abs(S.get_y() - want).max() <= 4e-9
        S.close()


def test_cli_iterate_flag(torch, tmp_path):
    p = subprocess.run([sm.CLI_PATH, "-c", "-t", "--normalize", "-n", "5", "-d", str(tmp_path), ob.fixture_path("ibm32.mtx")],
                       capture_output=True, text=True)
    assert p.returncode == 0 and "Power iteration" in p.stdout
    m, n, coo = load("ibm32.mtx")
    want = ob.fmt_g(ob.csr_iterate(*sm.csr_from_coo(coo, m), np.ones(n), 5, normalize=True))
    for f in os.listdir(tmp_path):
        got = ob.report_y_lines(open(tmp_path / f).read())
        assert all(abs(float(a) - float(b)) <= 2e-6 for a, b in zip(got, want))


# ------------------------------------------------------------- randomised structures (fuzz)
def _fuzz_matrix(seed):
    """Random shape and skew: power-law row lengths, empty rows, occasional huge rows, narrow or wide column range."""
    rng = np.random.default_rng(9000 + seed)
    rows = int(rng.integers(1, 4000))
    cols = int(rng.integers(1, 6000))
    style = seed % 5
    if style == 0:
        lens = rng.integers(0, 6, rows)
    elif style == 1:
        lens = np.minimum((rng.pareto(1.2, rows) * 3).astype(int), cols)
    elif style == 2:
        lens = np.where(rng.random(rows) < 0.7, 0, rng.integers(1, 40, rows))
    elif style == 3:
        lens = rng.integers(0, 3, rows)
        lens[rng.integers(0, rows, 3)] = min(cols, int(rng.integers(1500, 5000)))
    else:
        lens = np.full(rows, min(cols, int(rng.integers(1, 70))))
    lens = np.minimum(lens, cols).astype(np.int64)
    row_ptr, col_ind, val = csr_from_lengths(rng, lens.tolist(), cols)
    val *= 10.0 ** rng.integers(-8, 8, len(val))
    x = rng.standard_normal(cols) * 10.0 ** rng.integers(-3, 3, cols)
    return rows, cols, row_ptr, col_ind, val, x


@pytest.mark.parametrize("seed", range(25))
def test_fuzz_all_kernels_against_oracle(torch, seed):
    rows, cols, row_ptr, col_ind, val, x = _fuzz_matrix(seed)
    ref = ob.csr_spmv(row_ptr, col_ind, val, x)
    scale = row_scale(row_ptr, col_ind, val, x)
    for kernel, param in CSR_VARIANTS:
        y = gpu_csr(torch, rows, cols, row_ptr, col_ind, val, x, kernel, param)
        assert_close(y, ref, scale)
    lens = np.diff(row_ptr)
    y = gpu_csr(torch, rows, cols, row_ptr, col_ind, val, x, sm.CSR_KERNEL_STREAM, 0)
    assert np.array_equal(y[lens <= 32], ref[lens <= 32])          # short rows: the serial loop's bits
    coo = sm.make_coo(np.repeat(np.arange(rows), lens), col_ind, val)
    coo = coo[np.random.default_rng(seed).permutation(len(coo))]
    for mode in TJDS_MODES:
        T = sm.TjdsMatrix(sm.tjds_from_coo(coo, rows, cols))
        T.set_mode(mode)
        if mode == sm.TJDS_MODE_ROW_GATHER:
            T.set_tile((256, 1024, 2048)[seed % 3])
        T.set_x(dev(torch, x))
        dy = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
        T.zero_y(dy)
        T.spmv(dy)
        torch.cuda.synchronize()
        assert_close(dy.cpu().numpy(), ref, scale)
        T.close()
    # device-built formats feed the same kernels
    d_coo = _coo_to_device(torch, coo)
    rp, ci, v = sm.csr_from_coo_device(d_coo, rows, cols, len(coo))
    assert np.array_equal(rp.cpu().numpy(), row_ptr) and np.array_equal(ci.cpu().numpy(), col_ind)


@pytest.mark.parametrize("timing", [sm.TIMING_EVENTS, sm.TIMING_DEVICE])
def test_many_iterations_use_the_timing_rings(torch, timing):
    """-n beyond the ring of 1024 event pairs / 256 stamped products per graph replay: every product still gets its own time."""
    m, n, coo = load("ibm32.mtx")
    y, ms, st = sm.csr_compute(coo, m, n, iters=2500, timing=timing)
    assert len(ms) == 2500 and np.all(ms > 0) and st.time_total == pytest.approx(ms.sum())
    assert sm.last_run_info().timing == timing
    assert ob.fmt_g(y) == ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_1615284655.txt"))
    y, ms, st = sm.tjds_compute(coo, m, n, iters=1025, timing=timing)
    assert len(ms) == 1025 and np.all(ms > 0)
    assert ob.fmt_g(y) == ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_1615284655.txt"))


def test_repeating_kernel_gives_up_quickly_and_the_run_falls_back(torch):
    """The repeating kernel's barrier is bounded: with no patience at all (repeat_patience_us < 0, the test hook) every workgroup
    that has to wait gives up, the launch says so on its top word and in the run's sticky word, the host -- which waits for the
    FIRST launch of a run before queueing the others -- does the run over from a hipGraph, one launch per product: same y, every
    product timed, and the whole thing takes milliseconds, not the patience of a thousand queued launches (-n 100000
    on a shared GPU held the device for minutes).  With the default patience the same run uses the repeating kernel."""
    import time

    m, n, coo = load("memplus.mtx")
    want = None
    for fn in (sm.csr_compute, sm.tjds_compute):
        y_ok, ms_ok, _ = fn(coo, m, n, iters=3000)
        info = sm.last_run_info()
        assert info.repeat_launches == 3 and info.repeat_gave_up == 0 and info.graph_replays == 0
        t0 = time.perf_counter()
        y, ms, st = fn(coo, m, n, iters=3000, repeat_patience_us=-1)
        wall = time.perf_counter() - t0
        info = sm.last_run_info()
        assert info.repeat_gave_up == 1 and info.repeat_launches == 0 and info.graph_replays > 0 and info.timing == sm.TIMING_DEVICE
        assert np.array_equal(y, y_ok) and len(ms) == 3000 and np.all(ms > 0) and 0.0005 < st.time_avg < 0.05
        assert wall < 5.0, wall          # conversion + one abandoned launch + 3000 replayed products
        # a single launch (no second one to protect): it gives up too, and is done over the same way
        y1, ms1, _ = fn(coo, m, n, iters=100, repeat_patience_us=-1)
        info = sm.last_run_info()
        assert info.repeat_gave_up == 1 and info.graph_replays > 0 and np.array_equal(y1, y_ok) and np.all(ms1 > 0)
        want = y_ok if want is None else want
    # a generous patience is as good as the default
    y, _, _ = sm.csr_compute(coo, m, n, iters=200, repeat_patience_us=10_000_000)
    assert sm.last_run_info().repeat_launches == 1 and np.array_equal(y, want)


def test_device_timing_agrees_with_events_and_is_the_default_for_small_launches(torch):
    """The reference's window holds the product only (main-cli.c:408-419).  On its own sample matrices the kernel times
    itself (wall-clock stamps per wave; up to 1024 products per launch of the repeating kernel, or -- TIMING_DEVICE_GRAPH -- one
    launch per product replayed from a hipGraph); a hipEvent pair around the same launch can only read longer (it includes
    the launch and the events), and never by more than a few launch overheads.  Every form gives the same y."""
    m, n, coo = load("memplus.mtx")
    for fn in (sm.csr_compute, sm.tjds_compute):
        y_d, ms_d, st_d = fn(coo, m, n, iters=300)                       # AUTO -> device stamps for 493 workgroups
        info = sm.last_run_info()
        assert info.timing == sm.TIMING_DEVICE and info.repeat_launches == 1 and info.graph_replays == 0 and info.device_clock_khz > 0
        assert info.wall_ms >= st_d.time_total * 0.5 and len(ms_d) == 300 and np.all(ms_d > 0)
        # the windows of one launch cannot overlap: their sum is at most the loop's wall time
        assert st_d.time_total <= info.wall_ms
        y_g, ms_g, st_g = fn(coo, m, n, iters=300, timing=sm.TIMING_DEVICE_GRAPH)
        info_g = sm.last_run_info()
        assert info_g.timing == sm.TIMING_DEVICE and info_g.graph_replays == 2 and info_g.repeat_launches == 0
        assert np.array_equal(y_d, y_g)
        assert st_d.time_avg <= st_g.time_avg * 1.25                       # warm caches, no dispatch ramp: not slower than a launch of its own
        assert info.wall_ms < info_g.wall_ms                               # and the loop is shorter: no launch boundary per product
        y_e, ms_e, st_e = fn(coo, m, n, iters=300, timing=sm.TIMING_EVENTS)
        assert sm.last_run_info().timing == sm.TIMING_EVENTS
        assert np.array_equal(y_d, y_e)
        assert 0.0005 < st_d.time_min <= st_d.time_avg < 0.05           # microseconds, not garbage ticks
        assert st_d.time_avg <= st_e.time_avg * 1.05 and st_e.time_avg - st_d.time_avg < 0.03
    # a launch of more than 4096 workgroups keeps the event pair under AUTO
    rng = np.random.default_rng(5)
    rows = 2_500_000                                                            # 5 M entries = 4883 tiles of 1024
    coo_big = sm.make_coo(np.repeat(np.arange(rows), 2), rng.integers(0, rows, 2 * rows), rng.random(2 * rows))
    y_auto, _, _ = sm.csr_compute(coo_big, rows, rows, iters=3)                 # (AUTO: scattered far columns -> the binned plan)
    assert sm.last_run_info().timing == sm.TIMING_EVENTS
    y_tile, _, _ = sm.csr_compute(coo_big, rows, rows, iters=3, kernel=sm.CSR_KERNEL_STREAM)
    assert sm.last_run_info().timing == sm.TIMING_EVENTS and np.allclose(y_auto, y_tile, rtol=0, atol=1e-12)
    sm.csr_compute(coo_big, rows, rows, iters=3, kernel=sm.CSR_KERNEL_STREAM, timing=sm.TIMING_DEVICE)   # but can be asked for
    assert sm.last_run_info().timing == sm.TIMING_DEVICE
    with pytest.raises(sm.SmvpError):                                           # not of a product of several launches
        sm.csr_compute(coo_big, rows, rows, iters=3, kernel=sm.CSR_KERNEL_BINNED, timing=sm.TIMING_DEVICE)
    with pytest.raises(sm.SmvpError):                                           # not with a changing operand
        sm.csr_compute(coo, m, n, iters=3, iterate=True, timing=sm.TIMING_DEVICE)


def test_config4_full_size_properties(torch):
    """BASELINE config 4 at its full size: 10 M x 10 M, 32 entries per row (320 M entries, 3.8 GB).

    AUTO picks the column sweep here.  Size-independent properties only: y(ones) = row sums (independent numpy
    computation), linearity, agreement of the sweep with the two tile kernels -- bit for bit, all three sum these rows in
    serial order --, run-to-run bit equality, and the oracle on the first 20 000 rows."""
    M = 10_000_000
    row_ptr, col_ind, val = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, M, M, param=32)
    assert len(col_ind) == 320_000_000
    A = sm.CsrMatrix(M, M, row_ptr, col_ind, val)
    assert A.get_kernel() == (sm.CSR_KERNEL_COLSWEEP, 19532)     # 512 workgroups = two FULL generations of 256, strips of 4883 rows (round 6; round 5: 7816 = five)
    ones = torch.ones(M, dtype=torch.float64, device="cuda")
    y1, y2, ya, yb, yab = (torch.empty(M, dtype=torch.float64, device="cuda") for _ in range(5))
    A.spmv(ones, y1)
    A.spmv(ones, y2)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    host = val.reshape(M, 32).sum(axis=1)
    scale = np.abs(val).reshape(M, 32).sum(axis=1)
    assert np.all(np.abs(y1.cpu().numpy() - host) <= TOL * scale)
    rng = np.random.default_rng(11)
    xa, xb = dev(torch, rng.random(M)), dev(torch, rng.random(M))
    A.spmv(xa, ya)
    A.spmv(xb, yb)
    A.spmv(xa + xb, yab)
    torch.cuda.synchronize()
    sc = torch.from_numpy(scale).cuda() * 2
    assert bool(((yab - (ya + yb)).abs() <= TOL * sc).all())
    k = 20_000
    ref = ob.csr_spmv(row_ptr[:k + 1].copy(), col_ind[:row_ptr[k]], val[:row_ptr[k]], xa.cpu().numpy())
    assert np.array_equal(ya.cpu().numpy()[:k], ref)        # ascending column order inside every row: the serial bits
    for kernel in (sm.CSR_KERNEL_STREAM, sm.CSR_KERNEL_STREAM_CARRY):
        A.set_kernel(kernel, 0)
        A.spmv(xa, y2)
        torch.cuda.synchronize()
        assert torch.equal(ya, y2)                          # 32 entries per row: one lane, serial order, same bits
    A.close()


@pytest.mark.parametrize("name", SAMPLES)
def test_device_built_arrays_against_committed_golden_arrays(torch, name):
    """COO -> CSR / TJDS on the GPU, compared directly with tests/golden/arrays/*.npz (not via the host converters):
    row_ptr, col_ind, perm, start_pos, row_ind bit for bit, and the quirk scalars."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arrays", name.replace(".mtx", ".npz")))
    m, n, coo = load(name)
    rng = np.random.default_rng(4)
    d_coo = _coo_to_device(torch, coo[rng.permutation(len(coo))])          # any input order
    rp, ci, v = sm.csr_from_coo_device(d_coo, m, n, len(coo))
    assert np.array_equal(rp.cpu().numpy(), g["row_ptr"]) and np.array_equal(ci.cpu().numpy(), g["col_ind"])
    t = sm.tjds_from_coo_device(d_coo, m, n, len(coo))
    for f in ("perm", "start_pos", "row_ind"):
        assert np.array_equal(getattr(t, f).cpu().numpy(), g[f]), f
    assert (t.num_diag, t.ref_num_tjdiag, t.last_diag_single) == \
        (int(g["num_diag"]), int(g["ref_num_tjdiag"]), int(g["last_diag_single"]))
    # and the values travel with their indices: the product of the device-built arrays is the golden vector
    A = sm.CsrMatrix(m, n, rp, ci, v)
    dy = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
    A.spmv(torch.ones(n, dtype=torch.float64, device="cuda"), dy)
    torch.cuda.synchronize()
    lens = np.diff(g["row_ptr"])
    assert np.array_equal(dy.cpu().numpy()[lens <= 32], g["y_csr"][lens <= 32])
    A.close()


@pytest.mark.parametrize("name", SAMPLES)
def test_gpu_y_against_committed_golden_vectors(torch, name):
    """tests/golden/arrays/*.npz: full-precision y of the serial loop (oracle, pinned against the reference's reports)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arrays", name.replace(".mtx", ".npz")))
    m, n, coo = load(name)
    y, _, _ = sm.csr_compute(coo, m, n, iters=2)
    lens = np.diff(g["row_ptr"])
    assert np.array_equal(y[lens <= 32], g["y_csr"][lens <= 32])          # one lane, serial order: identical bits
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    assert_close(y, g["y_csr"], row_scale(row_ptr, col_ind, val, np.ones(n)), exact=name in EXACT)
    yq, _, _ = sm.tjds_compute(coo, m, n, iters=1, ref_quirks=True)
    assert_close(yq, g["y_tjds_refquirks"], row_scale(row_ptr, col_ind, val, np.ones(n)), exact=name in EXACT)


def test_products_can_be_captured_in_a_callers_graph(torch):
    """A caller may record the products into a hipGraph of its own (one warm call first: plans and function attributes are set
    up outside the capture): the tile kernel, the column sweep, the binned plan's three kernels on their two streams, the vector
    kernel and the TJDS product (operand permutation + product) replay with new operands and give the oracle's result."""
    M = 1 << 16
    rp, ci, v = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, M, M)
    rng = np.random.default_rng(3)
    xs = [rng.random(M) for _ in range(3)]
    A = sm.CsrMatrix(M, M, rp, ci, v)
    coo = sm.make_coo(np.repeat(np.arange(M), np.diff(rp)), ci, v)
    T = sm.TjdsMatrix(sm.tjds_from_coo(coo, M, M))
    dx = torch.zeros(M, dtype=torch.float64, device="cuda")
    dy = torch.zeros(M, dtype=torch.float64, device="cuda")
    forms = [("stream", lambda: A.set_kernel(sm.CSR_KERNEL_STREAM, 0), lambda st: A.spmv(dx, dy, stream=st)),
             ("colsweep", lambda: A.set_kernel(sm.CSR_KERNEL_COLSWEEP, 0), lambda st: A.spmv(dx, dy, stream=st)),
             ("binned", lambda: A.set_kernel(sm.CSR_KERNEL_BINNED, 0), lambda st: A.spmv(dx, dy, stream=st)),
             ("vector", lambda: A.set_kernel(sm.CSR_KERNEL_VECTOR, 8), lambda st: A.spmv(dx, dy, stream=st)),
             ("tjds", lambda: None, lambda st: (T.set_x(dx, stream=st), T.spmv(dy, stream=st)))]
    for name, setup, run in forms:
        setup()
        dx.copy_(torch.from_numpy(xs[0]))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                run(s)
        for x in xs:
            dx.copy_(torch.from_numpy(x))
            dy.fill_(float("nan"))
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert_close(dy.cpu().numpy(), ob.csr_spmv(rp, ci, v, x), row_scale(rp, ci, v, x))
        del g
    A.close()
    T.close()


def test_host_threads_with_their_own_handles_and_streams(torch):
    """The library is re-entrant per handle (SURVEY 8(b) "Threading"): six host threads create, plan, multiply with and destroy
    their own CSR / TJDS handles on their own streams at the same time -- every kernel family, the binned plan's side stream
    included -- and call the reference-shaped entry point concurrently; every result is the oracle's."""
    import threading

    M = 1 << 16
    rp, ci, v = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, M, M)
    tj = sm.tjds_from_coo(sm.make_coo(np.repeat(np.arange(M), np.diff(rp)), ci, v), M, M)
    m32, n32, coo32 = load("ibm32.mtx")
    want32 = ob.report_y_lines(ob.read_report("smvp-toolbox_report_CSR_1615284655.txt"))
    kernels = [(sm.CSR_KERNEL_STREAM, 0), (sm.CSR_KERNEL_COLSWEEP, 0), (sm.CSR_KERNEL_BINNED, 0), (sm.CSR_KERNEL_VECTOR, 8)]
    errors = []

    def worker(k):
        try:
            torch.cuda.set_device(0)
            rng = np.random.default_rng(k)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for it in range(4):
                    A = sm.CsrMatrix(M, M, rp, ci, v)
                    T = sm.TjdsMatrix(tj)
                    A.set_kernel(*kernels[(k + it) % 4])
                    x = rng.random(M)
                    dx = torch.from_numpy(x).cuda()
                    dy = torch.full((M,), float("nan"), dtype=torch.float64, device="cuda")
                    dt = torch.full((M,), float("nan"), dtype=torch.float64, device="cuda")
                    for _ in range(5):
                        A.spmv(dx, dy, stream=s)
                        T.set_x(dx, stream=s)
                        T.spmv(dt, stream=s)
                    s.synchronize()
                    ref, scale = ob.csr_spmv(rp, ci, v, x), row_scale(rp, ci, v, x)
                    for what, got in (("csr", dy), ("tjds", dt)):
                        if not np.all(np.abs(got.cpu().numpy() - ref) <= 1e-9 * scale):
                            errors.append((what, k, it))
                    A.close()
                    T.close()
            y, ms, st = sm.csr_compute(coo32, m32, n32, iters=50)
            if ob.fmt_g(y) != want32:
                errors.append(("compute", k))
        except Exception as e:   # a thread's exception must fail the test, not vanish
            errors.append((k, type(e).__name__, str(e)[:200]))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert errors == []
