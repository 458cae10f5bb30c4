#!/usr/bin/env python3
"""Time the TJDS product forms (and the CSR product beside them) on HBM-sized workloads; check each against CSR.

    python3 tools/exp_tjds.py --workloads memplus_tiled,pwt_tiled,random,uniform --steps 20
Variants: mode[:index[:tile[:cache]]]  with mode in two_phase | atomic | gather, index in sorted | half | k32, tile 256|1024|2048
(0 = default), cache = tiles per val line from which on the values are cached (0 = none; default 8).
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def workload(sm, sharding, ob, name, scale):
    if name in ("memplus_tiled", "pwt_tiled"):
        f = "memplus.mtx" if name == "memplus_tiled" else "pwt.mtx"
        tc, m, n, coo = sm.mm_read_coo(ob.fixture_path(f))
        rp, ci, v = sm.csr_from_coo(coo, m)
        copies = max(1, int(scale * (1 << 24)) // m)
        RP, CI, V = sharding.tile_block_diagonal(rp, ci, v, n, 0, copies)
        return "%s x%d" % (f, copies), m * copies, n * copies, RP, CI, V
    if name == "random":
        rows = int(scale * (1 << 24))
        RP, CI, V = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, rows, rows, 0, 0, rows)
        return "memplus_shaped random model rows=%d" % rows, rows, rows, RP, CI, V
    rows = int(scale * 10_000_000)
    RP, CI, V = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, rows, rows, 32, 0, rows)
    return "uniform32 rows=%d" % rows, rows, rows, RP, CI, V


def timeit(torch, fn, steps, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        tms.append(e0.elapsed_time(e1) / steps)
    return sorted(tms)[len(tms) // 2], min(tms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="memplus_tiled")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--variants", default="two_phase,gather:sorted:1024,gather:sorted:2048,gather:k32:1024")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--x", default="ones")
    a = ap.parse_args()
    import torch
    import smvp_toolkit_amd as sm
    from smvp_toolkit_amd import sharding
    import oracle_binding as ob

    modes = {"two_phase": sm.TJDS_MODE_TWO_PHASE, "atomic": sm.TJDS_MODE_ATOMIC, "gather": sm.TJDS_MODE_ROW_GATHER,
             }
    st = torch.cuda.current_stream()
    for w in a.workloads.split(","):
        t0 = time.perf_counter()
        name, rows, cols, RP, CI, V = workload(sm, sharding, ob, w, a.scale)
        nnz = int(RP[-1])
        print("# %s: rows=%d nnz=%d (built in %.1f s)" % (name, rows, nnz, time.perf_counter() - t0), flush=True)
        x_host = np.ones(cols) if a.x == "ones" else np.random.default_rng(67890).random(cols)
        x = torch.from_numpy(x_host).cuda()
        A = sm.CsrMatrix(rows, cols, torch.from_numpy(RP).cuda(), torch.from_numpy(CI).cuda(), torch.from_numpy(V).cuda())
        y = torch.empty(rows, dtype=torch.float64, device="cuda")
        A.spmv(x, y, stream=st)
        torch.cuda.synchronize()
        y_csr = y.clone()
        scale = torch.empty(rows, dtype=torch.float64, device="cuda")
        Aabs = sm.CsrMatrix(rows, cols, torch.from_numpy(RP).cuda(), torch.from_numpy(CI).cuda(), torch.from_numpy(np.abs(V)).cuda())
        Aabs.spmv(x.abs(), scale, stream=st)
        torch.cuda.synchronize()
        Aabs.close()
        kname, cbytes = A.describe()
        ms, mn = timeit(torch, lambda: A.spmv(x, y, stream=st), a.steps)
        print("%-44s %8.4f ms (min %.4f)  %5.1f %% of 8 TB/s  %7.1f GFLOP/s" % ("CSR " + kname, ms, mn, cbytes / ms * 1e-6 / 80.0, 2 * nnz / ms * 1e-6), flush=True)
        A.close()
        del A
        coo = np.zeros(nnz, dtype=sm.COO_DTYPE)
        coo["row"] = np.repeat(np.arange(rows, dtype=np.int32), np.diff(RP))
        coo["col"], coo["val"] = CI, V
        d_coo = torch.from_numpy(coo.view(np.uint8)).cuda()
        del coo
        tj = sm.tjds_from_coo_device(d_coo, rows, cols, nnz)
        del d_coo
        print("# TJDS: %d jagged diagonals" % tj.num_diag, flush=True)
        for var in a.variants.split(","):
            parts = var.split(":")
            mode = modes[parts[0]]
            sm.set_option("tjds_index", {"half": 0, "sorted": 1, "k32": 2}[parts[1] if len(parts) > 1 else "half"])
            t0 = time.perf_counter()
            T = sm.TjdsMatrix(tj)          # default plan is built here ...
            T.set_mode(mode)               # ... and the variant's own one here
            if len(parts) > 2 and int(parts[2]) and mode >= sm.TJDS_MODE_ROW_GATHER:
                T.set_tile(int(parts[2]))
            if len(parts) > 3 and mode >= sm.TJDS_MODE_ROW_GATHER:
                T.set_value_cache(int(parts[3]))
            torch.cuda.synchronize()
            plan_s = time.perf_counter() - t0
            T.set_x(x, stream=st)
            y.fill_(float("nan"))

            def step():
                T.zero_y(y, stream=st)
                T.spmv(y, stream=st)

            step()
            torch.cuda.synchronize()
            err = float(((y - y_csr).abs() / scale.clamp_min(1e-300)).max())
            tname, tbytes = T.describe()
            ms, mn = timeit(torch, step, a.steps)
            cache = T.get_value_cache()
            print("%-44s %8.4f ms (min %.4f)  %5.1f %% of 8 TB/s  %7.1f GFLOP/s  err %.1e  plan %.2f s  [%s] cache >= %d tiles: %.1f %% of the values" % (
                var, ms, mn, tbytes / ms * 1e-6 / 80.0, 2 * nnz / ms * 1e-6, err, plan_s, tname, cache[0], 100.0 * cache[1] / max(nnz, 1)), flush=True)
            T.close()
            del T
        del tj, y, y_csr, scale, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
