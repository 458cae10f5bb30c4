"""The committed golden arrays (tests/golden/arrays, made by tests/golden/make_arrays.py) against the product's host
converters and against the oracle -- a pin that survives edits to either."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import smvp_toolkit_amd as sm
from conftest import SAMPLES

ARRAYS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arrays")


@pytest.mark.parametrize("name", SAMPLES)
def test_converters_match_committed_arrays(name):
    g = np.load(os.path.join(ARRAYS, name.replace(".mtx", ".npz")), allow_pickle=False)
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path(name))
    assert (m, n, tc) == (int(g["rows"]), int(g["cols"]), str(g["typecode"]))
    row_ptr, col_ind, val = sm.csr_from_coo(coo, m)
    assert np.array_equal(row_ptr, g["row_ptr"]) and np.array_equal(col_ind, g["col_ind"])
    t = sm.tjds_from_coo(coo, m, n)
    for f in ("perm", "start_pos", "row_ind"):
        assert np.array_equal(getattr(t, f), g[f]), f
    assert (t.num_diag, t.ref_num_tjdiag, t.last_diag_single) == \
        (int(g["num_diag"]), int(g["ref_num_tjdiag"]), int(g["last_diag_single"]))


@pytest.mark.parametrize("name", SAMPLES)
def test_oracle_matches_committed_vectors(name):
    g = np.load(os.path.join(ARRAYS, name.replace(".mtx", ".npz")), allow_pickle=False)
    rc, tc, m, n, coo = ob.mm_read_coo(ob.fixture_path(name))
    y = ob.csr_spmv(*ob.csr_build(coo, m), np.ones(n))
    assert y.tobytes() == g["y_csr"].tobytes()
    yq = ob.tjds_spmv(ob.tjds_build(coo, m, n), np.ones(n), refquirks=True)
    assert yq.tobytes() == g["y_tjds_refquirks"].tobytes()
