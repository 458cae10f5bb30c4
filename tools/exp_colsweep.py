#!/usr/bin/env python3
"""The column-swept CSR kernel (SMVP_CSR_KERNEL_COLSWEEP) against the tile kernel on matrices with scattered columns.

    python3 tools/exp_colsweep.py --workload uniform|random|random_far [--rows N --local-rows L --per-row P] --rb 0,8192,4096
uniform = BASELINE config 4 (optionally only the first L rows: one rank's block; N columns, P entries per row);
random = the SURVEY 8(d) model (N rows);
random_far = its entries beyond distance 4096 through the sweep, the rest through the tile kernel.
(The per-cell and thread-count variants recorded in profiles/r02_colsweep_measured.txt were run with earlier
revisions of this tool and of the kernel.)
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def timeit(torch, fn, steps=10, reps=5):
    for _ in range(2):
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
    return sorted(tms)[len(tms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="uniform")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--local-rows", type=int, default=0)
    ap.add_argument("--per-row", type=int, default=32)
    ap.add_argument("--rb", default="0")
    ap.add_argument("--g", default="0", help="chunks in flight per wavefront (0 = the library's rule, 1, 2, 4), comma list")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--parts", default="1", help="column parts per strip (1, 2, 4), comma list")
    a = ap.parse_args()
    import torch
    import smvp_toolkit_amd as sm
    if a.workload == "uniform":
        rows, cols = a.local_rows or a.rows, a.rows
        rp, ci, v = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, a.rows, a.rows, a.per_row, 0, rows, threads=32)
    else:
        rows = cols = a.rows if a.rows != 10_000_000 else 1 << 24
        rp, ci, v = sm.synth_csr(sm.SYNTH_MEMPLUS_SHAPED, 12345, rows, rows, 0, 0, rows, threads=32)
    nnz = int(rp[-1])
    alg = 12.0 * nnz + 4.0 * (rows + 1) + 8.0 * rows + 8.0 * cols
    st = torch.cuda.current_stream()
    x = torch.ones(cols, dtype=torch.float64, device="cuda")
    y = torch.empty(rows, dtype=torch.float64, device="cuda")
    d_rp, d_ci, d_v = torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(v).cuda()
    A = sm.CsrMatrix(rows, cols, d_rp, d_ci, d_v)
    print("# AUTO picks kernel %s; gather spread %.3f" % (A.get_kernel(), A.gather_spread()), flush=True)
    A.set_kernel(sm.CSR_KERNEL_STREAM, 0)
    A.spmv(x, y, stream=st)
    torch.cuda.synchronize()
    y_ref = y.clone()
    t_full = timeit(torch, lambda: A.spmv(x, y, stream=st))
    print("# %s rows=%d cols=%d nnz=%d: tile kernel %.4f ms = %.1f %% of 8 TB/s, %.1f G gathers/s" % (
        a.workload, rows, cols, nnz, t_full, alg / t_full * 1e-6 / 80, nnz / t_full * 1e-6), flush=True)
    t_near, S = 0.0, A
    if a.workload == "random_far":
        d_row = torch.repeat_interleave(torch.arange(rows, dtype=torch.int32, device="cuda"), (d_rp[1:] - d_rp[:-1]).long())
        far = (d_row.long() - d_ci.long()).abs() > 4096

        def part(mask):
            cnt = torch.zeros(rows + 1, dtype=torch.int64, device="cuda")
            cnt[1:] = torch.bincount(d_row[mask].long(), minlength=rows)
            return sm.CsrMatrix(rows, cols, torch.cumsum(cnt, 0).int(), d_ci[mask].contiguous(), d_v[mask].contiguous())
        N, S = part(~far), part(far)
        y_near = torch.empty_like(y)
        N.spmv(x, y_near, stream=st)
        torch.cuda.synchronize()
        t_near = timeit(torch, lambda: N.spmv(x, y_near, stream=st))
        print("# near part (%d entries) on the tile kernel: %.4f ms" % (int((~far).sum()), t_near), flush=True)
        y_ref = y_ref - y_near
        nnz = int(far.sum())
    for rb, g, parts in [(int(s), g, int(p)) for s in a.rb.split(",") for p in a.parts.split(",") for g in a.g.split(",") * a.repeat]:
        S.set_kernel(sm.CSR_KERNEL_COLSWEEP, sm.sweep_parts(rb, parts, int(g)))
        y.fill_(float("nan"))
        S.spmv(x, y, stream=st)
        torch.cuda.synchronize()
        err = float((y - y_ref).abs().max() / y_ref.abs().max())
        y2 = y.clone()
        S.spmv(x, y, stream=st)
        torch.cuda.synchronize()
        assert torch.equal(y, y2), "not the same bits from run to run"
        ms = timeit(torch, lambda: S.spmv(x, y, stream=st))
        tot = ms + t_near
        print("colsweep G=%s parts %d rows/block %5d (asked %d): %.4f ms  %.1f G gathers/s  max err %.1e  |  whole product %.4f ms = %.1f %% of 8 TB/s" % (
            g, 1 << ((S.get_kernel()[1] >> 24) & 3), S.get_kernel()[1] & 0xffffff, rb, ms, nnz / ms * 1e-6, err, tot, alg / tot * 1e-6 / 80), flush=True)


if __name__ == "__main__":
    main()
