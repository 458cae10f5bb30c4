"""Property tests (hypothesis) for the host converters: structure invariants that hold for any input, CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle_binding as ob
import smvp_toolkit_amd as sm


@st.composite
def coo_matrices(draw):
    rows = draw(st.integers(1, 40))
    cols = draw(st.integers(1, 40))
    cells = draw(st.lists(st.integers(0, rows * cols - 1), min_size=0, max_size=200, unique=True))
    vals = draw(st.lists(st.floats(-1e3, 1e3, allow_nan=False, width=64), min_size=len(cells), max_size=len(cells)))
    order = draw(st.permutations(range(len(cells))))
    r = np.array([cells[i] // cols for i in order], dtype=np.int32)
    c = np.array([cells[i] % cols for i in order], dtype=np.int32)
    v = np.array([vals[i] for i in order], dtype=np.float64)
    return rows, cols, sm.make_coo(r, c, v)


@settings(max_examples=150, deadline=None)
@given(coo_matrices())
def test_csr_invariants(m):
    rows, cols, coo = m
    row_ptr, col_ind, val = sm.csr_from_coo(coo, rows)
    assert row_ptr[0] == 0 and row_ptr[-1] == len(coo) and np.all(np.diff(row_ptr) >= 0)
    for r in range(rows):
        seg = col_ind[row_ptr[r]:row_ptr[r + 1]]
        assert np.all(np.diff(seg) > 0)                                   # sorted, distinct
    # same multiset of entries
    rr = np.repeat(np.arange(rows), np.diff(row_ptr))
    got = sorted(zip(rr.tolist(), col_ind.tolist(), val.tolist()))
    want = sorted(zip(coo["row"].tolist(), coo["col"].tolist(), coo["val"].tolist()))
    assert got == want
    # bit-exact with the oracle
    for a, b in zip((row_ptr, col_ind, val), ob.csr_build(coo, rows)):
        assert a.tobytes() == b.tobytes()


@settings(max_examples=150, deadline=None)
@given(coo_matrices())
def test_tjds_invariants(m):
    rows, cols, coo = m
    t = sm.tjds_from_coo(coo, rows, cols)
    col_len = np.bincount(coo["col"], minlength=cols)
    assert sorted(t.perm.tolist()) == list(range(cols))
    plen = col_len[t.perm]
    assert np.all(plen[:-1] >= plen[1:])                                   # longest column first
    same = plen[:-1] == plen[1:]
    assert np.all(t.perm[:-1][same] < t.perm[1:][same])                    # ties by original index
    assert t.num_diag == (col_len.max() if len(coo) else 0)
    assert t.start_pos[0] == 0 and t.start_pos[-1] == len(coo)
    width = np.diff(t.start_pos)
    assert np.all(width[:-1] >= width[1:]) and (len(width) == 0 or width[0] == (col_len > 0).sum())
    assert t.ref_num_tjdiag == col_len[0]
    assert t.last_diag_single == int(len(width) > 0 and width[-1] == 1)
    # entry j of diagonal d sits in permuted column j - start_pos[d]; rows ascend down a column
    dense = np.zeros((rows, cols))
    dense[coo["row"], coo["col"]] = coo["val"]
    prev_row = np.full(cols, -1)
    for d in range(t.num_diag):
        for j in range(t.start_pos[d], t.start_pos[d + 1]):
            c = t.perm[j - t.start_pos[d]]
            r = t.row_ind[j]
            assert r > prev_row[c] and dense[r, c] == t.val[j]
            prev_row[c] = r
    # oracle agreement and product equivalence
    u = ob.tjds_build(coo, rows, cols)
    for f in ("perm", "start_pos", "row_ind", "val"):
        assert getattr(t, f).tobytes() == getattr(u, f).tobytes()
    x = np.linspace(0.5, 1.5, cols)
    y_csr = ob.csr_spmv(*ob.csr_build(coo, rows), x)
    assert np.allclose(ob.tjds_spmv(u, x), y_csr, rtol=0, atol=1e-9 * (1 + np.abs(dense).sum(axis=1).max()))


@settings(max_examples=50, deadline=None)
@given(st.integers(0, 3000), st.integers(1, 9), st.integers(0, 2 ** 32 - 1))
def test_partition_rows_is_a_partition(rows, parts, seed):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 20, rows)
    row_ptr = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(lens, out=row_ptr[1:])
    b = sm.partition_rows(row_ptr, parts)
    assert b[0] == 0 and b[-1] == rows and np.all(np.diff(b) >= 0) and len(b) == parts + 1
