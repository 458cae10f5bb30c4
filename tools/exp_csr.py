#!/usr/bin/env python3
"""Kernel experiments on one GPU: time CSR / TJDS products on a synthetic matrix.

    python tools/exp_csr.py --kind memplus --rows-log2 24 [--cols N] --variants stream:2048,stream:1024,vector:8

Prints one line per variant: ms per product, algorithmic GB/s, % of the 8 TB/s HBM peak.
Development aid only; bench.py is the measured contract.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="memplus", choices=["memplus", "uniform"])
    ap.add_argument("--rows-log2", type=int, default=24)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--cols", type=int, default=0, help="0 = square")
    ap.add_argument("--param", type=int, default=32)
    ap.add_argument("--variants", default="stream:2048,stream:1024,vector:8")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--x", default="ones")
    ap.add_argument("--tjds", action="store_true")
    a = ap.parse_args()

    import torch
    import smvp_toolkit_amd as sm

    rows = a.rows or (1 << a.rows_log2)
    cols = a.cols or rows
    kind = sm.SYNTH_MEMPLUS_SHAPED if a.kind == "memplus" else sm.SYNTH_UNIFORM
    t0 = time.time()
    rp, ci, v = sm.synth_csr(kind, 12345, rows, cols, a.param)
    nnz = int(rp[-1])
    print("# %s rows=%d cols=%d nnz=%d (%.1f/row) gen %.1fs" % (a.kind, rows, cols, nnz, nnz / rows, time.time() - t0), flush=True)
    A = sm.CsrMatrix(rows, cols, torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(v).cuda())
    xh = np.ones(cols) if a.x == "ones" else np.random.default_rng(1).random(cols)
    x = torch.from_numpy(xh).cuda()
    y = torch.empty(rows, dtype=torch.float64, device="cuda")
    ref = None
    for var in a.variants.split(","):
        fam, par = var.split(":")
        A.set_kernel({"stream": 2, "vector": 1, "carry": 3}[fam], int(par))
        name, nbytes = A.describe()
        for _ in range(3):
            A.spmv(x, y)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            A.spmv(x, y, stream=torch.cuda.current_stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        gbs = nbytes / ms * 1e-6
        print("%-22s %8.4f ms  %8.1f GB/s  %5.1f %% of 8 TB/s   %7.1f GFLOP/s  maxdiff %.1e" %
              (name, ms, gbs, gbs / 80.0, 2 * nnz / ms * 1e-6, err), flush=True)
    if a.tjds:
        coo = np.zeros(nnz, dtype=sm.COO_DTYPE)
        coo["row"] = np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp))
        coo["col"], coo["val"] = ci, v
        tj = sm.tjds_from_coo(coo, rows, cols)
        T = sm.TjdsMatrix(tj)
        T.set_x(x)
        name, nbytes = T.describe()
        for _ in range(2):
            T.zero_y(y)
            T.spmv(y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(max(3, a.steps // 5)):
            T.zero_y(y, stream=torch.cuda.current_stream())
            T.spmv(y, stream=torch.cuda.current_stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / max(3, a.steps // 5)
        gbs = nbytes / ms * 1e-6
        print("%-22s %8.4f ms  %8.1f GB/s  %5.1f %% of 8 TB/s   D=%d  maxdiff %.1e" %
              (name, ms, gbs, gbs / 80.0, tj.num_diag, float((y - ref).abs().max())), flush=True)


if __name__ == "__main__":
    main()
