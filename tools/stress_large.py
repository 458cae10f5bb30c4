#!/usr/bin/env python3
"""One-off stress run at HBM sizes (not part of the suite): the default plans against their plainer forms, bit for bit.

    python3 tools/stress_large.py [--cases N]
For synthetic matrices of 1 ... 60 M entries (memplus-shaped, uniform rows of various lengths and widths, bands):
  CSR   default plan (16-bit column offsets / column sweep where they apply) == 32-bit columns, tile kernel (the binned plan,
        where AUTO picks it: within the rounding bound of it);
        the column sweep == the serial order (checked against the oracle on a slice of rows);
  TJDS  16-bit second word == 32-bit second word == no value cache, and within the bound of the CSR product.
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def cases(n):
    out = [("memplus_shaped", 1 << 20, 1 << 20, 0), ("memplus_shaped", 1 << 22, 1 << 22, 0),
           ("uniform", 600_000, 5_000_000, 24), ("uniform", 2_000_000, 2_000_000, 9), ("uniform", 1_500_000, 40_000, 16),
           ("uniform", 300_000, 70_000_000, 40), ("band", 3_000_000, 3_000_000, 11), ("band", 9_000_000, 9_000_000, 5),
           ("uniform", 4_000_000, 4_000_000, 4), ("memplus_shaped", 1 << 23, 1 << 23, 0)]
    return out[:n]


def build(sm, kind, rows, cols, per_row, seed):
    if kind == "memplus_shaped":
        return sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, seed, rows, cols, 0, 0, rows, threads=16)
    if kind == "uniform":
        return sm.synth_csr(sm.SYNTH_UNIFORM, seed, max(rows, cols), cols, per_row, 0, rows, threads=16)
    rng = np.random.default_rng(seed)
    off = np.sort(rng.choice(np.arange(-3 * per_row, 3 * per_row + 1), size=per_row, replace=False))
    ci = np.clip(np.arange(rows, dtype=np.int64)[:, None] + off[None, :], 0, cols - 1).astype(np.int32)
    ci.sort(axis=1)
    return (np.arange(rows + 1, dtype=np.int64) * per_row).astype(np.int32), ci.ravel(), rng.uniform(-1, 1, rows * per_row)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=10)
    a = ap.parse_args()
    import torch
    import smvp_toolkit_amd as sm
    import oracle_binding as ob
    bad = 0
    for i, (kind, rows, cols, per_row) in enumerate(cases(a.cases)):
        t0 = time.perf_counter()
        rp, ci, v = build(sm, kind, rows, cols, per_row, 100 + i)
        nnz = int(rp[-1])
        x = np.random.default_rng(i).random(cols)
        dx = torch.from_numpy(x).cuda()
        d = [torch.from_numpy(t).cuda() for t in (rp, ci, v)]
        ys, names = [], []
        for env in ({}, {"csr_col16": 0}):
            for k, val_ in env.items():
                sm.set_option(k, val_)
            A = sm.CsrMatrix(rows, cols, *d)
            if env:
                A.set_kernel(sm.CSR_KERNEL_STREAM, 0)
            for k in env:
                sm.set_option(k, None)
            y = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
            A.spmv(dx, y)
            torch.cuda.synchronize()
            ys.append(y)
            names.append(A.describe()[0])
            if not env and A.get_kernel()[0] != sm.CSR_KERNEL_COLSWEEP and nnz >= 4 << 20:   # the sweep too, where AUTO did not pick it
                A.set_kernel(sm.CSR_KERNEL_COLSWEEP, 0)
                ysw = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
                A.spmv(dx, ysw)
                torch.cuda.synchronize()
                ys.append(ysw)
                names.append(A.describe()[0])
            A.close()
        k = min(rows, 30_000)
        ref = ob.csr_spmv(rp[:k + 1].copy(), ci[:rp[k]], v[:rp[k]], x)
        scale = ob.csr_spmv(rp[:k + 1].copy(), ci[:rp[k]], np.abs(v[:rp[k]]), x)
        lens = np.diff(rp[:k + 1])
        msg = []
        for y, nm in zip(ys, names):
            got = y[:k].cpu().numpy()
            ok = np.all(np.abs(got - ref) <= 1e-9 * scale)
            binned = "csr_binned" in nm or "csr_binned" in names[0]     # near sum + far sum: the serial loop's bits only on rows without far entries
            serial = np.array_equal(got, ref) if "colsweep" in nm else binned or np.array_equal(got[lens <= 32], ref[lens <= 32])
            same = torch.equal(y, ys[0]) if "colsweep" not in nm and not binned else bool(((y - ys[0]).abs() <= 1e-9 * (ys[0].abs() + 1)).all())
            msg.append("%s %s" % (nm, "ok" if ok and serial and same else "MISMATCH"))
            bad += 0 if ok and serial and same else 1
        # TJDS of the same matrix (device conversion), three plans
        coo = np.zeros(nnz, dtype=sm.COO_DTYPE)
        coo["row"] = np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp))
        coo["col"], coo["val"] = ci, v
        d_coo = torch.from_numpy(coo.view(np.uint8)).cuda()
        del coo
        tj = sm.tjds_from_coo_device(d_coo, rows, cols, nnz)
        del d_coo
        yt = []
        for index, cache in (("half", 2), ("sorted", 4), ("half", 0)):
            sm.set_option("tjds_index", {"half": 0, "sorted": 1, "k32": 2}[index])
            T = sm.TjdsMatrix(tj)
            sm.set_option("tjds_index", None)
            T.set_value_cache(cache)
            T.set_x(dx)
            y = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
            T.spmv(y)
            torch.cuda.synchronize()
            yt.append(y)
            nm = "%s/%d %s" % (index, cache, T.describe()[0])
            ok = bool(((y - ys[0]).abs()[:k].cpu().numpy() <= 2e-9 * scale).all()) and torch.equal(y, yt[0])
            msg.append("%s %s" % (nm, "ok" if ok else "MISMATCH"))
            bad += 0 if ok else 1
            T.close()
        print("[%d] %s %d x %d, %d entries (%.1f s): %s" % (i, kind, rows, cols, nnz, time.perf_counter() - t0, "; ".join(msg)), flush=True)
        del ys, yt, d, dx, tj
        torch.cuda.empty_cache()
    print("done: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
