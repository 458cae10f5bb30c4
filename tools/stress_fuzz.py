#!/usr/bin/env python3
"""One-off stress run (not part of the suite): many random structures through every product form, against the oracle.

    python3 tools/stress_fuzz.py --seeds 400
Shapes are biased towards the corners of the tile logic: entry counts at and around multiples of 256 / 1024 / 2048, rows
that end exactly on tile edges, rows longer than the LDS overflow area, empty leading / trailing rows, one column.
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def matrix(seed):
    rng = np.random.default_rng(70000 + seed)
    style = seed % 9
    cols = int(rng.integers(1, 9000))
    if style == 0:      # total entries exactly a multiple of a tile size
        target = int(rng.choice([256, 512, 1024, 2048, 3072, 4096, 6144])) + int(rng.integers(-2, 3))
        lens = []
        while sum(lens) < target:
            lens.append(int(min(cols, rng.integers(0, 40), target - sum(lens))))
        lens += [0] * int(rng.integers(0, 5))
    elif style == 1:    # rows ending exactly on tile edges
        tile = int(rng.choice([256, 1024, 2048]))
        cols = max(cols, tile)
        lens = [tile, tile - 1, 1, tile, 3, tile - 3] * int(rng.integers(1, 4))
    elif style == 2:    # very long rows between short ones
        cols = max(cols, 6000)
        lens = rng.integers(0, 4, int(rng.integers(10, 400))).tolist()
        for _ in range(int(rng.integers(1, 4))):
            lens[int(rng.integers(0, len(lens)))] = int(rng.integers(1025, 6000))
    elif style == 3:    # mostly empty
        lens = np.where(rng.random(int(rng.integers(1, 5000))) < 0.9, 0, rng.integers(1, 9, 1)).tolist()
    elif style == 4:    # one column
        cols = 1
        lens = rng.integers(0, 2, int(rng.integers(1, 3000))).tolist()
    elif style == 5:    # uniform rows of 33 / 32 / 31 entries (the serial-lane limit)
        cols = max(cols, 40)
        lens = [int(rng.choice([31, 32, 33]))] * int(rng.integers(1, 300))
    elif style == 6:    # power law
        lens = np.minimum((rng.pareto(1.1, int(rng.integers(1, 3000))) * 2).astype(int), cols).tolist()
    elif style == 7:    # a single row
        lens = [int(rng.integers(0, min(cols, 5000) + 1))]
    else:               # wide and large: positions and permuted columns reach over many blocks of 2^16 (the TJDS plan's 16-bit words)
        cols = int(rng.integers(70_000, 400_000))
        lens = rng.integers(0, 9, int(rng.integers(30_000, 90_000))).tolist()
        for _ in range(int(rng.integers(0, 3))):
            lens[int(rng.integers(0, len(lens)))] = int(rng.integers(100, 3000))
    lens = np.minimum(np.array(lens, dtype=np.int64), cols)
    rows = len(lens)
    row_ptr = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(lens, out=row_ptr[1:])
    if style == 8:      # (rng.choice without replacement over 400 000 columns per row is slow: draw, sort, drop repeats)
        parts = []
        for l in lens:
            c = np.unique(rng.integers(0, cols, int(l)))
            parts.append(c)
        lens = np.array([len(c) for c in parts], dtype=np.int64)
        np.cumsum(lens, out=row_ptr[1:])
        col_ind = np.concatenate(parts + [np.zeros(0, int)]).astype(np.int32)
    else:
        col_ind = np.concatenate([np.sort(rng.choice(cols, size=int(l), replace=False)) for l in lens] + [np.zeros(0, int)]).astype(np.int32)
    val = rng.uniform(-1, 1, int(row_ptr[-1])) * 10.0 ** rng.integers(-6, 6, int(row_ptr[-1]))
    x = rng.standard_normal(cols)
    return rows, cols, row_ptr, col_ind, val, x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    a = ap.parse_args()
    import torch
    import smvp_toolkit_amd as sm
    import oracle_binding as ob
    csr_variants = [(sm.CSR_KERNEL_STREAM, p) for p in (256, 1024, 2048)] + [(sm.CSR_KERNEL_STREAM_CARRY, p) for p in (1024, 2048)] + \
                   [(sm.CSR_KERNEL_VECTOR, p) for p in (2, 16, 64)] + [(sm.CSR_KERNEL_COLSWEEP, p) for p in (0, 1024, 8192, 300, 1028)] + \
                   [(sm.CSR_KERNEL_BINNED, p) for p in (0, 1, 7, 100)]
    bad = 0
    for seed in range(a.seeds):
        rows, cols, row_ptr, col_ind, val, x = matrix(seed)
        ref = ob.csr_spmv(row_ptr, col_ind, val, x)
        scale = ob.csr_spmv(row_ptr, col_ind, np.abs(val), np.abs(x))
        dx = torch.from_numpy(x).cuda()

        def check(y, what):
            nonlocal bad
            if not np.all(np.abs(y - ref) <= 1e-9 * scale) or np.isnan(y).any():
                bad += 1
                print("MISMATCH seed %d (%d x %d, %d entries): %s" % (seed, rows, cols, int(row_ptr[-1]), what), flush=True)

        A = sm.CsrMatrix(rows, cols, row_ptr, col_ind, val)
        for k, p in csr_variants:
            A.set_kernel(k, p)
            dy = torch.full((max(rows, 1),), float("nan"), dtype=torch.float64, device="cuda")
            A.spmv(dx, dy)
            torch.cuda.synchronize()
            check(dy.cpu().numpy()[:rows], "csr kernel %d param %d" % (k, p))
            if k == sm.CSR_KERNEL_COLSWEEP and not np.array_equal(dy.cpu().numpy()[:rows], ref):   # the sweep is the serial order
                bad += 1
                print("NOT BIT-IDENTICAL seed %d: colsweep param %d" % (seed, p), flush=True)
        A.close()
        coo = sm.make_coo(np.repeat(np.arange(rows), np.diff(row_ptr)), col_ind, val)
        coo = coo[np.random.default_rng(seed).permutation(len(coo))]
        t = sm.tjds_from_coo(coo, rows, cols)
        for index in ("half", "sorted", "k32"):
            sm.set_option("tjds_index", {"half": 0, "sorted": 1, "k32": 2}[index])
            T = sm.TjdsMatrix(t)
            T.set_x(dx)
            for tile in (0, 256, 1024, 2048):
                if tile:
                    T.set_tile(tile)
                dy = torch.full((max(rows, 1),), float("nan"), dtype=torch.float64, device="cuda")
                T.spmv(dy)
                torch.cuda.synchronize()
                check(dy.cpu().numpy()[:rows], "tjds %s tile %d" % (index, tile))
                if index != "k32":           # the value cache never changes a bit
                    first = dy.clone()
                    for cache in (0, 1, 2, 8):
                        T.set_value_cache(cache)
                        dy.fill_(float("nan"))
                        T.spmv(dy)
                        torch.cuda.synchronize()
                        if not torch.equal(dy[:rows], first[:rows]):
                            bad += 1
                            print("NOT BIT-IDENTICAL seed %d: tjds tile %d cache %d" % (seed, tile, cache), flush=True)
                    T.set_value_cache(4)
            for mode in (sm.TJDS_MODE_TWO_PHASE, sm.TJDS_MODE_ATOMIC):
                T.set_mode(mode)
                dy = torch.full((max(rows, 1),), float("nan"), dtype=torch.float64, device="cuda")
                T.zero_y(dy)
                T.spmv(dy)
                torch.cuda.synchronize()
                check(dy.cpu().numpy()[:rows], "tjds mode %d" % mode)
            T.close()
        sm.set_option("tjds_index", None)
        y, _, _ = sm.csr_compute(coo, rows, cols, iters=3, x=x)        # (device-timed: the repeating kernel, three products in one launch)
        check(y, "csr_compute")
        y, _, _ = sm.tjds_compute(coo, rows, cols, iters=3, x=x, device_convert=True)
        check(y, "tjds_compute (device conversion)")
        if rows > 0:
            y, _, _ = sm.csr_compute(coo, rows, cols, iters=2, x=x, timing=sm.TIMING_DEVICE_GRAPH if seed % 2 else sm.TIMING_EVENTS)
            check(y, "csr_compute, other timing forms")
        if seed % 50 == 49:
            print("... %d seeds, %d mismatches" % (seed + 1, bad), flush=True)
    print("done: %d seeds, %d mismatches" % (a.seeds, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
