#!/usr/bin/env python3
"""tools/exp_binned.py -- time the binned CSR plan (near part + pass A + pass B) on the SURVEY 8(d) random model.

    python tools/exp_binned.py [--rows-log2 24] [--band 0] [--steps 50] [--no-check] [--kernel binned|auto|stream]

Prints ms per product (HIP events) and, unless --no-check (for the library's SMVP_BINNED_DBG timing variants, whose
results are wrong on purpose), the worst row-normwise error against numpy.  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.  Development aid, not part of the library or the tests.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-log2", type=int, default=24)
    ap.add_argument("--kind", default="memplus", choices=["memplus", "uniform"])
    ap.add_argument("--per-row", type=int, default=32)
    ap.add_argument("--band", type=int, default=0)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--kernel", default="binned", choices=["binned", "auto", "stream", "colsweep"])
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--churn", type=int, default=0, help="allocate and free this many odd-sized device buffers first (memory placement experiment)")
    args = ap.parse_args()
    import torch
    import smvp_toolkit_amd as sm

    rows = 1 << args.rows_log2
    if args.churn:
        rng = np.random.default_rng(1)
        held = []
        for i in range(args.churn):
            held.append(torch.empty(int(rng.integers(1 << 20, 1 << 29)), dtype=torch.uint8, device="cuda"))
            if len(held) > 24:
                del held[int(rng.integers(0, len(held)))]
        keep = held[::3]                 # every third stays: what comes next is allocated around them
        del held
        torch.cuda.empty_cache()
    if args.kind == "memplus":
        rp, ci, v = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, rows, rows, threads=16)
    else:
        rp, ci, v = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, rows, rows, args.per_row, threads=16)
    nnz = int(rp[-1])
    A = sm.CsrMatrix(rows, rows, torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(v).cuda())
    if args.kernel != "auto":
        A.set_kernel({"binned": sm.CSR_KERNEL_BINNED, "stream": sm.CSR_KERNEL_STREAM, "colsweep": sm.CSR_KERNEL_COLSWEEP}[args.kernel], args.band)
    name, alg = A.describe()
    x = sm.vector_random(rows)
    dx = torch.from_numpy(x).cuda()
    dy = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    A.spmv(dx, dy)
    torch.cuda.synchronize()
    if not args.no_check:
        prod = v * x[ci]
        ref = np.add.reduceat(prod, np.minimum(rp[:-1], nnz - 1)) * (np.diff(rp) > 0)
        scale = np.add.reduceat(np.abs(prod), np.minimum(rp[:-1], nnz - 1)) * (np.diff(rp) > 0)
        err = np.abs(dy.cpu().numpy() - ref) / np.maximum(scale, 1e-300)
        print("max normwise error vs numpy: %.3g" % err.max())
        assert err.max() < 1e-9
    for _ in range(5):
        A.spmv(dx, dy)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        A.spmv(dx, dy)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    pi = A.plan_info()
    print("%s\n  rows %d nnz %d: %.4f ms per product = %.1f GFLOP/s, %.4f of 8 TB/s (algorithmic); plan %.2f x matrix, built in %.0f ms"
          % (name, rows, nnz, ms, 2.0 * nnz / ms * 1e-6, alg / ms * 1e-6 / 8000.0, pi["plan_bytes"] / pi["matrix_bytes"], pi["build_ms"]))
    A.close()


if __name__ == "__main__":
    main()
