"""Synthetic workload generators (BASELINE.json configs 2-4): host only."""
import numpy as np
import pytest

import smvp_toolkit_amd as sm

MEMPLUS_MEAN = 126150 / 17758


def test_block_independence_and_determinism():
    M = 50_000
    rp, ci, v = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, M, M)
    rp2, ci2, v2 = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, M, M, threads=1)
    assert rp.tobytes() == rp2.tobytes() and ci.tobytes() == ci2.tobytes() and v.tobytes() == v2.tobytes()
    # any row block generated alone equals the matching slice of the whole
    a, b = 12_345, 30_001
    brp, bci, bv = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, M, M, row_begin=a, row_end=b)
    assert np.array_equal(brp, rp[a:b + 1] - rp[a])
    assert np.array_equal(bci, ci[rp[a]:rp[b]]) and np.array_equal(bv, v[rp[a]:rp[b]])
    # another seed gives another matrix
    assert sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 1, M, M)[1].tobytes() != ci.tobytes()


def test_memplus_shape():
    M = 200_000
    rp, ci, v = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, M, M)
    lens = np.diff(rp)
    assert abs(lens.mean() - MEMPLUS_MEAN) < 0.35 and lens.min() >= 2 and lens.max() <= 574
    assert (lens <= 8).mean() == pytest.approx(0.856, abs=0.01)          # memplus: 86 % of rows <= 8 entries
    rows = np.repeat(np.arange(M), lens)
    dist = np.abs(rows - ci)
    # every row keeps its diagonal, columns sorted and distinct inside a row
    assert (dist == 0).sum() == M
    inner = np.ones(len(ci), bool)
    inner[rp[1:-1]] = False
    assert np.all(np.diff(ci)[inner[1:]] > 0)
    assert ci.min() >= 0 and ci.max() < M
    # band profile of memplus (cumulative share of entries within a distance)
    for d, share in ((8, 0.295), (64, 0.334), (512, 0.421), (4096, 0.615)):
        assert (dist <= d).mean() == pytest.approx(share, abs=0.03)
    assert v.min() >= -1.0 and v.max() < 1.0 and abs(v.mean()) < 0.01


def test_uniform_kind():
    M, N, K = 10_000, 40_000, 32
    rp, ci, v = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, M, N, param=K)
    assert np.all(np.diff(rp) == K) and len(ci) == M * K
    c = ci.reshape(M, K)
    assert np.all(np.diff(c, axis=1) > 0) and c.min() >= 0 and c.max() < N
    assert abs(c.mean() / N - 0.5) < 0.01


def test_small_column_count_is_clipped():
    rp, ci, v = sm.synth_csr(sm.SYNTH_UNIFORM, 1, 5, 3, param=8)
    assert np.all(np.diff(rp) == 3) and ci.reshape(5, 3).tolist() == [[0, 1, 2]] * 5


def test_bad_arguments():
    with pytest.raises(sm.SmvpError):
        sm.synth_csr(99, 1, 10, 10)
    with pytest.raises(sm.SmvpError):
        sm.synth_csr(sm.SYNTH_UNIFORM, 1, 10, 10, param=4, row_begin=5, row_end=20)
