#!/usr/bin/env python3
"""Does the headline kernel's time depend on WHERE its streams lie?  (bench.py's headline read 0.269 or 0.280 ms from process to
process on one box, profiles/r06_headline_variance.txt.)  kron(I_944, memplus) with val at different offsets inside one big
device buffer, the plan rebuilt for each; every placement measured twice.

    python3 tools/exp_placement.py [--copies 944] [--offsets-kb 0,4,64,...]
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=944)
    ap.add_argument("--offsets-kb", default="0,4,8,16,32,64,128,256,512,1024,2048,4096,1028,2052,68")
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    import torch
    import smvp_toolkit_amd as sm
    from smvp_toolkit_amd import sharding
    import bench_core as core
    blk = core.build_block(sm, sharding, "memplus_tiled", argparse.Namespace(copies=a.copies, scaling="strong"), 0, 1)
    rows, cols, nnz = blk["rows"], blk["cols_total"], blk["nnz"]
    st = torch.cuda.current_stream()
    x = torch.ones(cols, dtype=torch.float64, device="cuda")
    y = torch.empty(rows, dtype=torch.float64, device="cuda")
    d_rp, d_ci = torch.from_numpy(blk["row_ptr"]).cuda(), torch.from_numpy(blk["col_ind"]).cuda()
    arena = torch.empty(nnz * 8 + (64 << 20), dtype=torch.uint8, device="cuda")
    print("# arena at 0x%x (mod 2 MB = %d KB), x at 0x%x, y at 0x%x" % (arena.data_ptr(), (arena.data_ptr() % (2 << 20)) >> 10, x.data_ptr(), y.data_ptr()))
    h_val = torch.from_numpy(blk["val"])
    alg = 12.0 * nnz + 4.0 * (rows + 1) + 8.0 * rows + 8.0 * cols

    def measure(A):
        for _ in range(30):
            A.spmv(x, y, stream=st)
        torch.cuda.synchronize()
        best = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                A.spmv(x, y, stream=st)
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / 50)
        return sorted(best)[2]

    for rep in range(a.repeat):
        for kb in [int(s) for s in a.offsets_kb.split(",")]:
            off = kb << 10
            d_val = arena[off:off + nnz * 8].view(torch.float64)
            d_val.copy_(h_val)
            A = sm.CsrMatrix(rows, cols, d_rp, d_ci, d_val)
            ms = measure(A)
            print("val at arena + %5d KB: %.4f ms = %.4f of 8 TB/s  (%s)" % (kb, ms, alg / ms * 1e-6 / 8000, A.describe()[0]), flush=True)
            A.close()


if __name__ == "__main__":
    main()
