#!/usr/bin/env python3
"""The far gathers of the SURVEY 8(d) random model: where the time goes, and what ANY plan-level split could reach.

The model's rows are cut at |row - col| > 4096 into a near and a far part (39 % of the entries are far: uniform over
all 2^24 columns).  Measured with the product's own CSR kernel (tile sizes 1024 / 2048 = 4 / 8 gathers in flight per
lane), each on a matrix of the same row structure:
  full        the model as it is
  near_only   far entries dropped                      -> cost of everything but the far gathers
  far_only    near entries dropped                     -> cost of the 46 M far gathers (+ their 12 B/entry stream)
  far_folded  far_only with every column folded into one 2 MB window of x (col mod 262144): the far part as it would
              run if a 2-D blocking scheme delivered every far gather from L2 -- an upper bound for any such scheme
              (cells of 8 K rows x 256 K columns hold ~350 entries each; y cannot be blocked any coarser in LDS)
best case of a split = near_only + far_folded; realistic split = near_only + far_only.
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import smvp_toolkit_amd as sm
    rows = 1 << 24
    rp, ci, v = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, rows, rows, 0, 0, rows, threads=32)
    nnz = int(rp[-1])
    r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rp))
    far = np.abs(r - ci) > 4096
    print("# SURVEY 8(d) model: rows=%d nnz=%d far=%d (%.1f %%)" % (rows, nnz, far.sum(), 100.0 * far.mean()), flush=True)
    alg_full = 12.0 * nnz + 4.0 * (rows + 1) + 16.0 * rows

    def sub(mask, fold=False):
        cnt = np.zeros(rows + 1, dtype=np.int64)
        np.add.at(cnt, r[mask] + 1, 1)
        rp2 = np.cumsum(cnt).astype(np.int32)
        c2 = ci[mask].copy()
        if fold:
            c2 = (c2 % 262144).astype(np.int32)
        return rp2, c2, v[mask]

    x = torch.ones(rows, dtype=torch.float64, device="cuda")
    y = torch.empty(rows, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream()
    res = {}
    for name, (rp2, c2, v2) in (("full", (rp, ci, v)), ("near_only", sub(~far)), ("far_only", sub(far)),
                                ("far_folded", sub(far, True))):
        n2 = int(rp2[-1])
        A = sm.CsrMatrix(rows, rows, torch.from_numpy(rp2).cuda(), torch.from_numpy(c2).cuda(), torch.from_numpy(v2).cuda())
        for tile in (1024, 2048):
            A.set_kernel(sm.CSR_KERNEL_STREAM, tile)
            for _ in range(3):
                A.spmv(x, y, stream=st)
            torch.cuda.synchronize()
            tms = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    A.spmv(x, y, stream=st)
                e1.record()
                torch.cuda.synchronize()
                tms.append(e0.elapsed_time(e1) / 10)
            ms = sorted(tms)[2]
            res[(name, tile)] = ms
            print("%-11s tile %4d  %9d entries  %7.4f ms  %6.1f G gathers/s" % (name, tile, n2, ms, n2 / ms * 1e-6), flush=True)
        A.close()
    for tile in (1024, 2048):
        real = res[("near_only", tile)] + res[("far_only", tile)]
        best = res[("near_only", tile)] + res[("far_folded", tile)]
        print("tile %d: full %.4f ms = %.1f %% of 8 TB/s | split, far gathers as they are: %.4f ms = %.1f %% | split, every far "
              "gather an L2 hit (bound for any 2-D blocking): %.4f ms = %.1f %%" % (
                  tile, res[("full", tile)], alg_full / res[("full", tile)] * 1e-6 / 80.0, real, alg_full / real * 1e-6 / 80.0,
                  best, alg_full / best * 1e-6 / 80.0), flush=True)


if __name__ == "__main__":
    main()
