#!/usr/bin/env python3
"""One-off stress run (not part of the suite): the binned plan's two near engines -- K6's window plan and the tile kernel --
on random multi-block structures, against the oracle and against each other.

    python3 tools/stress_near_window.py --seeds 200
Shapes: 1 ... 70 000 rows (ragged last block of 8192), square and rectangular, bands 0 (= 4096) / 1 / 5 / 64 / 1000 / 4096 /
5000 (too wide: the window plan must step aside), row lengths from several distributions (all short, many of 17 ... 400
entries, whole stretches empty, enough long rows in one block to exceed the plan's list), a share of entries anywhere (the far
passes), a few rows with more far entries than the cap (kept near: outside the window).
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def matrix(seed):
    rng = np.random.default_rng(90000 + seed)
    rows = int(rng.choice([1, 63, 64, 65, 8191, 8192, 8193, int(rng.integers(2, 70000))]))
    cols = rows if rng.random() < 0.5 else int(rng.integers(1, 70000))
    band = int(rng.choice([0, 1, 5, 64, 1000, 4096, 5000]))
    reach = band if band else 4096
    style = seed % 5
    if style == 0:
        lens = rng.integers(0, 17, rows)
    elif style == 1:
        lens = rng.choice([0, 1, 2, 3, 5, 8, 16, 17, 40, 400], rows, p=[.15, .2, .2, .15, .1, .08, .05, .03, .03, .01])
    elif style == 2:
        lens = np.where(rng.random(rows) < 0.8, 0, rng.integers(1, 30, rows))
    elif style == 3:                                  # a stretch with more long rows than a block's list holds
        lens = rng.integers(0, 6, rows)
        a = int(rng.integers(0, max(1, rows - 1500)))
        lens[a:a + 1500] = rng.integers(17, 60, min(1500, rows - a))
    else:
        lens = np.minimum((rng.pareto(1.2, rows) * 3).astype(np.int64), 3000)
    row_of = np.repeat(np.arange(rows), lens)
    off = rng.integers(-reach, reach + 1, row_of.size)
    col = np.clip(row_of + off, 0, cols - 1)
    far = rng.random(row_of.size) < rng.choice([0.0, 0.1, 0.5])
    col[far] = rng.integers(0, cols, int(far.sum()))
    extra_r, extra_c = [], []
    for r in rng.integers(0, rows, int(rng.integers(0, 3))):          # rows with more far entries than the cap
        extra_r.append(np.full(1200, r))
        extra_c.append(rng.integers(0, cols, 1200))
    all_r = np.concatenate([row_of] + extra_r).astype(np.int64)
    all_c = np.concatenate([col] + extra_c).astype(np.int64)
    if rng.random() < 0.7:                                              # columns ascending inside a row (not always)
        order = np.lexsort((all_c, all_r))
    else:
        order = np.argsort(all_r, kind="stable")
    all_r, all_c = all_r[order], all_c[order]
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(all_r, minlength=rows))]).astype(np.int32)
    val = rng.uniform(-1, 1, all_c.size) * 10.0 ** rng.integers(-3, 4, all_c.size)
    return rows, cols, band, row_ptr, all_c.astype(np.int32), val, rng.standard_normal(cols)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    a = ap.parse_args()
    import torch
    import smvp_toolkit_amd as sm
    import oracle_binding as ob
    bad = 0
    used = {"window": 0, "tile": 0}
    for seed in range(a.seeds):
        rows, cols, band, row_ptr, col_ind, val, x = matrix(seed)
        ref = ob.csr_spmv(row_ptr, col_ind, val, x)
        scale = ob.csr_spmv(row_ptr, col_ind, np.abs(val), np.abs(x))
        dx = torch.from_numpy(x).cuda()
        got = {}
        for near in ("window", "tile"):
            sm.set_option("binned_near", 1 if near == "tile" else 0)
            A = sm.CsrMatrix(rows, cols, row_ptr, col_ind, val)
            A.set_kernel(sm.CSR_KERNEL_BINNED, band)
            name = A.describe()[0]
            if near == "window":
                used["window" if "csr_near_window" in name else "tile"] += 1
            ys = []
            for _ in range(2):
                dy = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
                A.spmv(dx, dy)
                torch.cuda.synchronize()
                ys.append(dy.cpu().numpy())
            A.close()
            y = ys[0]
            if np.isnan(y).any() or not np.all(np.abs(y - ref) <= 1e-9 * scale) or not np.array_equal(ys[0], ys[1]):
                bad += 1
                print("MISMATCH seed %d (%d x %d, band %d, %d entries): near = %s (%s)" % (seed, rows, cols, band, int(row_ptr[-1]), near, name), flush=True)
            got[near] = y
        if seed % 20 == 19:
            print("seed %d done, %d mismatches so far; the window plan was used %d times, stepped aside %d times" % (seed, bad, used["window"], used["tile"]), flush=True)
    print("%d seeds, %d mismatches; window plan used %d, stepped aside %d" % (a.seeds, bad, used["window"], used["tile"]))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
