"""bench.py's workload builder: the row blocks of N ranks must tile the 1-rank matrix exactly (CPU only)."""
import argparse
import sys

import numpy as np
import pytest

import smvp_toolkit_amd as sm
from smvp_toolkit_amd import sharding

import bench
import bench_core
import bench_legs


def args_for(**kw):
    a = argparse.Namespace(copies=16, rows_log2=14, rows=5000)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("workload", ["memplus_tiled", "pwt_tiled", "memplus_shaped", "uniform32"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_blocks_tile_the_whole(workload, world):
    a = args_for()
    whole = bench_core.build_block(sm, sharding, workload, a, 0, 1)
    rows_seen, parts = 0, []
    for rank in range(world):
        blk = bench_core.build_block(sm, sharding, workload, a, rank, world)
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
    blk = bench_core.build_block(sm, sharding, "memplus_tiled", a, 0, 1)
    m, n, rp, ci, v, ncopies, report = blk["base"]
    assert ncopies == 3 and blk["rows"] == 3 * m and blk["nnz"] == 3 * len(ci)
    for c in range(3):
        a0, a1 = c * len(ci), (c + 1) * len(ci)
        assert np.array_equal(blk["col_ind"][a0:a1], ci + c * n)
        assert np.array_equal(blk["val"][a0:a1], v)
        assert np.array_equal(blk["row_ptr"][c * m:(c + 1) * m + 1] - c * len(ci), rp)


def test_host_check_catches_a_wrong_row():
    a = args_for(copies=2)
    blk = bench_core.build_block(sm, sharding, "memplus_tiled", a, 0, 1)
    x = np.ones(blk["cols_total"])
    good = np.add.reduceat(blk["val"], blk["row_ptr"][:-1])
    ok, worst, scale = bench_core.host_check(blk, x, good)
    assert ok and worst < 1e-12
    bad = good.copy()
    bad[123] += 1e-6 * scale[123]
    assert not bench_core.host_check(blk, x, bad)[0]


def test_c_layer_child_is_given_up_on_when_it_hangs(tmp_path, monkeypatch):
    """N > 1: bench.py runs the C-layer leg in a child process of rank 0 under a wall-clock budget -- a hang there must cost
    that leg ({"error": "timeout"}), not the run; a child that dies reports its last line; a child that answers is parsed."""
    import json
    import time

    def child(body):
        f = tmp_path / "child.py"
        f.write_text(body)
        monkeypatch.setattr(bench_core, "BENCH_SCRIPT", str(f))
        return bench_legs.c_layer_in_child(argparse.Namespace(rows=1000, c_layer_budget=1.5), 8, 5, rank=1)

    t0 = time.time()
    out = child("import time\ntime.sleep(60)\n")
    assert out["error"] == "timeout" and out["n_gpus"] == 8 and time.time() - t0 < 20
    out = child("import sys\nprint('no GPU here', file=sys.stderr)\nsys.exit(3)\n")
    assert "error" in out and "3" in out["error"]
    out = child("import json\nprint(json.dumps({'n_gpus': 8, 'chunks_1': {'overlapped': {'event_ms': 1.0}}}))\n")
    assert out["n_gpus"] == 8 and "child process of rank 0" in out["ran_in"] and out["chunks_1"]["overlapped"]["event_ms"] == 1.0


def test_plain_invocation_with_gpus_2_starts_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus 2` without torch.distributed.run around it (the driver's command shape) must not exit 1
    with "needs one process per GPU": the parent -- before torch or HIP are touched -- starts one child per rank with the
    rendezvous environment set, relays rank 0's JSON line and returns the children's worst exit code."""
    import io
    import subprocess

    started = []

    class Child:
        def __init__(self, cmd, env=None, stdout=None, **kw):
            started.append((cmd, env, kw))
            self.rank = int(env["RANK"])
            self.pid = 10 ** 7 + self.rank          # no such process group: end() must cope
            self.returncode = None
            self.stdout = io.StringIO('{"n_gpus": 2, "value": 1.0}\n') if stdout == subprocess.PIPE else None

        def poll(self):
            self.returncode = 0 if self.rank == 0 else rc_other[0]
            return self.returncode

        def wait(self, timeout=None):
            return self.poll()

        def kill(self):
            pass

    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(subprocess, "Popen", Child)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    rc_other = [0]
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert len(started) == 2
    for rank, (cmd, env, kw) in enumerate(started):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "2", "--steps", "3"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(rank) and env["WORLD_SIZE"] == "2" and env["MASTER_ADDR"] == "127.0.0.1"
        assert int(env["MASTER_PORT"]) > 0 and kw.get("start_new_session") is True
    assert started[0][1]["MASTER_PORT"] == started[1][1]["MASTER_PORT"]
    assert capsys.readouterr().out.count('{"n_gpus": 2') == 1
    # a rank that fails: its exit code comes out
    started.clear()
    rc_other[0] = 3
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3


def test_self_launched_ranks_really_run(tmp_path):
    """The same end to end on this CPU-only box: both ranks start, find no GPU, say so, and the launcher returns non-zero
    quickly with nothing left running."""
    import os
    import subprocess
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    t0 = time.time()
    p = subprocess.run([sys.executable, bench_core.BENCH_SCRIPT, "--gpus", "2", "--no-c-layer",
                        "--no-config4", "--copies", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present: the gpu-marked test covers the real run")
    assert p.returncode != 0 and "needs one process per GPU" not in p.stderr
    assert p.stderr.count("bench.py needs a GPU") == 2 and "starting 2 rank processes" in p.stderr
    assert time.time() - t0 < 120


def test_flat_keys_are_scalars_and_complete():
    """roofline's flat keys: what the driver's parse keeps (it drops nested objects)."""
    roof = {}
    others = {"tjds": {"frac": 0.5, "ms_per_product": 0.4, "traffic_over_algorithmic": 1.44, "moved_frac_of_peak": 0.7},
              "config4": {"frac": 0.22, "ms_per_product": 2.2, "t1_ms": 2.2, "tN_step_ms": 0.6, "tN_step_after_ms": 0.7,
                          "tN_products_only_ms": 0.4, "speedup_overlapped": 3.6, "speedup_after": 3.1, "speedup_products_only": 5.5,
                          "chunks_chosen": 2, "eighth_of_n8": {"chosen": 4, "inputs": {"product_ms_by_chunks": {"1": 0.41, "2": 0.5, "4": 0.54}}}},
              "config4_c_layer": {"chunks_1": {"overlapped": {"event_ms": 2.3}, "products_only": {"event_ms": 2.2}},
                                  "exchange_rccl_ms": 0.1, "exchange_direct_ms": 0.05, "exchange_chosen": "direct", "rccl_ranks": 8},
              "sample_matrices_us_per_product": {"memplus.mtx": {"csr_avg_ms": 3.1, "tjds_loop_wall_ms_per_product": 6.0}, "note": "x"}}
    extra = {"tjds_two_phase": {"frac_of_hbm_peak": 0.25}, "config5_pwt": {"csr_ms_per_step": 0.0037}}
    bench_legs.flat_keys(roof, others, extra, 8, {"backend": "nccl (RCCL)", "rccl_ranks": 8, "exchange": "e" * 300, "self_launched": True})
    assert all(not isinstance(v, (dict, list)) for v in roof.values())
    assert roof["frac_tjds"] == 0.5 and roof["traffic_over_alg_tjds"] == 1.44 and roof["frac_tjds_colmajor"] == 0.25
    assert roof["config4_speedup_overlapped"] == 3.6 and roof["config4_chunks_chosen"] == 2 and roof["config4_eighth_ms_1chunk"] == 0.41
    assert roof["config4_c_layer_overlapped_ms_1chunk"] == 2.3 and roof["exchange_direct_ms"] == 0.05 and roof["c_layer_rccl_ranks"] == 8
    assert roof["config4_c_layer_step_best_ms"] == 2.3 and roof["config4_c_layer_speedup_best"] == round(2.2 / 2.3, 3)
    assert roof["memplus_csr_us"] == 3.1 and roof["config5_csr_us"] == 3.7 and roof["rccl_ranks"] == 8 and roof["n_gpus"] == 8


def _state(**roofline_more):
    import time

    roof = {"bound": "hbm", "kernel": "csr_stream_owner<8, 5, false>", "achieved": 6230.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.7787,
            "traffic": 1601886515.1999998, "alg_bytes_per_launch": 1764298244.0, "ms_per_launch": 0.28319, "launches_per_product": 1,
            "ms_per_product": 0.28319, "plan": {"plan_bytes": 1}, "others": {"tjds": {"frac": 0.6}}, "note": "n" * 500}
    roof.update(roofline_more)
    return {"t0": time.time(), "errors": {}, "leg_seconds": {"headline": 12.3, "tjds": 4.0}, "extra": {"dist": {}}, "others": {"tjds": {"frac": 0.6}},
            "cpu": {"value": 0.9, "unit": "GFLOP/s", "cores": 1, "kind": "port", "host_cpu": "AMD EPYC", "sample": "s" * 400},
            "head": {"metric": "fp64 CSR SpMV GFLOP/s", "value": 850.0, "unit": "GFLOP/s", "n_gpus": 8}, "config": {"workload": "w" * 400, "rows": 1},
            "roofline": roof, "detail_path": "/nonexistent/bench_detail.json", "out": sys.stdout}


def test_compact_line_is_short_flat_and_keeps_the_contract():
    """The ONE stdout line: nothing nested below roofline / cpu_baseline / config, long strings cut, far below 8000 characters
    whatever the legs put into `roofline`; what has to go goes in DROP_ORDER, the contract's own keys never."""
    import json

    flat = {"frac_tjds": 0.61, "frac_survey_random_model": 0.33, "frac_config4": 0.26, "config4_t1_ms": 1.96, "config4_speedup_overlapped": 5.5,
            "rccl_ranks": 8}
    text = bench.compact_line(_state(**flat, frac_broken=float("nan"), ms_broken=float("inf")))
    j = json.loads(text, parse_constant=lambda c: pytest.fail("%s on the line: not JSON" % c))
    assert j["roofline"]["frac_broken"] is None and j["roofline"]["ms_broken"] is None
    assert len(text) < 3000 and "\n" not in text
    for obj in (j["roofline"], j["cpu_baseline"], j["config"]):
        assert all(not isinstance(v, (dict, list)) for v in obj.values())
        assert all(len(v) <= bench.STRING_LIMIT for v in obj.values() if isinstance(v, str))
    assert j["roofline"]["frac"] == 0.7787 and j["roofline"]["traffic"] == 1601890000.0 and j["cpu_baseline"]["cores"] == 1
    assert all(j["roofline"][k] == v for k, v in flat.items()) and "dropped_for_length" not in j["roofline"]
    assert j["roofline"]["leg_seconds"] == "headline 12, tjds 4" and j["value"] == 850.0 and j["n_gpus"] == 8
    # far too many keys: the least important prefixes leave, the line fits, the contract's keys and the fractions stay
    many = dict(flat, **{"memplus_k%d" % i: 0.123456 for i in range(300)}, **{"config4_c_layer_overlapped_ms_%d" % i: 1.5 for i in range(200)})
    st = _state(**many)
    st["errors"]["config4_c_layer"] = "timeout " * 100
    text = bench.compact_line(st)
    j = json.loads(text)
    assert len(text) <= bench.LINE_LIMIT < 8000 and j["roofline"]["dropped_for_length"] is True
    assert all(j["roofline"][k] == v for k, v in flat.items()) and j["roofline"]["ms_per_launch"] == 0.28319
    assert j["roofline"]["config4_c_layer_error"].startswith("timeout") and len(j["roofline"]["config4_c_layer_error"]) <= bench.STRING_LIMIT


def test_watchdog_prints_the_line_when_a_leg_hangs(tmp_path):
    """A leg that never returns costs that leg: at the hard deadline the watchdog prints the compact line from what has been
    measured (exit 0); with nothing measured yet it exits 3 and prints no line."""
    import json
    import subprocess
    import textwrap

    body = textwrap.dedent("""
        import sys, time
        sys.argv = ["bench.py"]
        sys.path.insert(0, %r)
        import bench
        from test_bench_blocks import _state
        st = _state(frac_tjds=0.61) if %s else {"t0": time.time(), "errors": {}, "leg_seconds": {}, "detail_path": %r, "out": sys.stdout}
        st["detail_path"], st["leg"] = %r, "config4"
        bench.Emitter(st, 0, 1.0)
        time.sleep(60)
    """)
    import os
    root = os.path.dirname(bench_core.BENCH_SCRIPT)
    for measured in (True, False):
        f = tmp_path / ("hang_%s.py" % measured)
        detail = str(tmp_path / "detail.json")
        f.write_text(body % (root, measured, detail, detail))
        p = subprocess.run([sys.executable, str(f)], capture_output=True, text=True, timeout=60,
                           env=dict(os.environ, PYTHONPATH=os.pathsep.join([root, os.path.join(root, "tests"), os.path.join(root, "smvp-toolkit_amd", "python")])))
        if measured:
            assert p.returncode == 0, p.stderr
            j = json.loads(p.stdout.splitlines()[-1])
            assert "config4" in j["roofline"]["watchdog"] and j["roofline"]["config4_error"] and j["roofline"]["frac_tjds"] == 0.61
            assert j["detail"] == "detail.json" and json.load(open(detail))["roofline"]["plan"] == {"plan_bytes": 1}
        else:
            assert p.returncode == 3 and p.stdout == ""
