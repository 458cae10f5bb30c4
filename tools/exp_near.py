#!/usr/bin/env python3
"""tools/exp_near.py -- what the near part of the binned plan (entries within `band` of the diagonal, on the tile kernel)
waits for: its own structure against the same rows with the gathers pulled in towards the diagonal.

    python tools/exp_near.py [--rows-log2 24] [--band 4096] [--steps 50]

Variants (same row lengths, same values, tile kernel `SMVP_CSR_KERNEL_STREAM`):
    as is          the near part of the SURVEY 8(d) random model
    within W       every column clipped to row +- W (W = 8, 64, 512): same streams, gathers nearer
    fused rows     the near rows padded with the row's far entry COUNT as extra diagonal entries: the entries per row of a fused
                   near + pass-B kernel, gathers as cheap as they get
Development aid, not part of the library or the tests.
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
    ap.add_argument("--band", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    import torch
    import smvp_toolkit_amd as sm

    rows = 1 << args.rows_log2
    rp, ci, v = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, rows, rows, threads=16)
    row_of = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rp))
    dist = np.abs(ci.astype(np.int64) - row_of)
    near = dist <= args.band
    lens = np.bincount(row_of[near], minlength=rows)
    nrp = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(lens, out=nrp[1:])
    nci, nv, nrow = ci[near], v[near], row_of[near]
    far_count = np.diff(rp) - lens
    print("near part: %d of %d entries, %.2f per row; |col - row| <= 8: %.1f %%, <= 64: %.1f %%, <= 512: %.1f %%" % (
        nci.size, ci.size, nci.size / rows, 100.0 * np.mean(dist[near] <= 8), 100.0 * np.mean(dist[near] <= 64), 100.0 * np.mean(dist[near] <= 512)))
    x = sm.vector_random(rows)
    dx = torch.from_numpy(x).cuda()
    dy = torch.empty(rows, dtype=torch.float64, device="cuda")

    def time_it(label, p, c, val):
        A = sm.CsrMatrix(rows, rows, torch.from_numpy(p).cuda(), torch.from_numpy(np.ascontiguousarray(c, dtype=np.int32)).cuda(),
                         torch.from_numpy(val).cuda())
        A.set_kernel(sm.CSR_KERNEL_STREAM, 0)
        name, alg = A.describe()
        for _ in range(5):
            A.spmv(dx, dy)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            A.spmv(dx, dy)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        print("%-28s %-34s %9d entries  %.4f ms  %.0f entries/us  %.0f GB/s of its algorithmic bytes" % (
            label, name, int(p[-1]), ms, int(p[-1]) / ms / 1e3, alg / ms / 1e6), flush=True)
        A.close()

    time_it("as is", nrp, nci, nv)
    for w in (512, 64, 8):
        d = np.clip(nci.astype(np.int64) - nrow, -w, w)
        c = np.clip(nrow + d, 0, rows - 1)
        order = np.lexsort((c, nrow))          # columns ascending inside a row again
        time_it("within %d" % w, nrp, c[order], nv[order])
    # the rows of a fused kernel: near entries + one diagonal entry per far entry
    flens = lens + far_count
    frp = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(flens, out=frp[1:])
    fci = np.repeat(np.arange(rows, dtype=np.int32), flens)
    fv = np.ones(fci.size)
    pos = frp[:-1][nrow] + (np.arange(nci.size) - nrp[:-1][nrow])
    fci[pos] = nci
    fv[pos] = nv
    order = np.lexsort((fci, np.repeat(np.arange(rows), flens)))
    time_it("fused rows (far -> diagonal)", frp, fci[order], fv[order])


if __name__ == "__main__":
    main()
