"""bench.py's workload builder: the row blocks of N ranks must tile the 1-rank matrix exactly (CPU only)."""
import argparse
import sys

import numpy as np
import pytest

import smvp_toolkit_amd as sm
from smvp_toolkit_amd import sharding

import bench


def args_for(**kw):
    a = argparse.Namespace(copies=16, rows_log2=14, rows=5000)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("workload", ["memplus_tiled", "pwt_tiled", "memplus_shaped", "uniform32"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_blocks_tile_the_whole(workload, world):
    a = args_for()
    whole = bench.build_block(sm, sharding, workload, a, 0, 1)
    rows_seen, parts = 0, []
    for rank in range(world):
        blk = bench.build_block(sm, sharding, workload, a, rank, world)
        assert blk["rows_total"] == whole["rows_total"] and blk["cols_total"] == whole["cols_total"]
        assert blk["r0"] == rows_seen and blk["r1"] - blk["r0"] == blk["rows"] == len(blk["row_ptr"]) - 1
        assert blk["bounds"][rank] == blk["r0"] and blk["bounds"][rank + 1] == blk["r1"]
        rows_seen = blk["r1"]
        parts.append(blk)
    assert rows_seen == whole["rows_total"]
    assert np.array_equal(np.concatenate([p["col_ind"] for p in parts]), whole["col_ind"])
    assert np.array_equal(np.concatenate([p["val"] for p in parts]), whole["val"])
    lens = np.concatenate([np.diff(p["row_ptr"]) for p in parts])
    assert np.array_equal(lens, np.diff(whole["row_ptr"]))
    # equal blocks: the all-gather needs no padding
    assert len(set(np.diff(parts[0]["bounds"]).tolist())) == 1


def test_tiled_block_is_kron_identity_memplus():
    a = args_for(copies=3)
    blk = bench.build_block(sm, sharding, "memplus_tiled", a, 0, 1)
    m, n, rp, ci, v, ncopies, report = blk["base"]
    assert ncopies == 3 and blk["rows"] == 3 * m and blk["nnz"] == 3 * len(ci)
    for c in range(3):
        a0, a1 = c * len(ci), (c + 1) * len(ci)
        assert np.array_equal(blk["col_ind"][a0:a1], ci + c * n)
        assert np.array_equal(blk["val"][a0:a1], v)
        assert np.array_equal(blk["row_ptr"][c * m:(c + 1) * m + 1] - c * len(ci), rp)


def test_host_check_catches_a_wrong_row():
    a = args_for(copies=2)
    blk = bench.build_block(sm, sharding, "memplus_tiled", a, 0, 1)
    x = np.ones(blk["cols_total"])
    good = np.add.reduceat(blk["val"], blk["row_ptr"][:-1])
    ok, worst, scale = bench.host_check(blk, x, good)
    assert ok and worst < 1e-12
    bad = good.copy()
    bad[123] += 1e-6 * scale[123]
    assert not bench.host_check(blk, x, bad)[0]


def test_c_layer_child_is_given_up_on_when_it_hangs(tmp_path, monkeypatch):
    """N > 1: bench.py runs the C-layer leg in a child process of rank 0 under a wall-clock budget -- a hang there must cost
    that leg ({"error": "timeout"}), not the run; a child that dies reports its last line; a child that answers is parsed."""
    import json
    import time

    def child(body):
        f = tmp_path / "child.py"
        f.write_text(body)
        monkeypatch.setattr(bench.os.path, "abspath", lambda p: str(f) if str(p).endswith("bench.py") else p)
        return bench.c_layer_in_child(argparse.Namespace(rows=1000, c_layer_budget=1.5), 8, 5, rank=1)

    t0 = time.time()
    out = child("import time\ntime.sleep(60)\n")
    assert out["error"] == "timeout" and out["n_gpus"] == 8 and time.time() - t0 < 20
    out = child("import sys\nprint('no GPU here', file=sys.stderr)\nsys.exit(3)\n")
    assert "error" in out and "3" in out["error"]
    out = child("import json\nprint(json.dumps({'n_gpus': 8, 'chunks_1': {'overlapped': {'event_ms': 1.0}}}))\n")
    assert out["n_gpus"] == 8 and "child process of rank 0" in out["ran_in"] and out["chunks_1"]["overlapped"]["event_ms"] == 1.0
