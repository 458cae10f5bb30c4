#!/usr/bin/env python3
"""Time the CSR kernels on kron(I_k, A) for a sample matrix A (block-diagonal replication)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))

def tile_csr(rp, ci, v, rows, cols, copies):
    nnz = int(rp[-1])
    RP = (rp[:-1][None, :].astype(np.int64) + (np.arange(copies, dtype=np.int64) * nnz)[:, None]).reshape(-1)
    RP = np.concatenate([RP, [copies * nnz]]).astype(np.int32)
    CI = (ci[None, :].astype(np.int64) + (np.arange(copies, dtype=np.int64) * cols)[:, None]).reshape(-1).astype(np.int32)
    V = np.tile(v, copies)
    return RP, CI, V

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="memplus.mtx"); ap.add_argument("--rows-log2", type=int, default=24)
    ap.add_argument("--variants", default="stream:2048,stream:1024,vector:8"); ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    import torch, smvp_toolkit_amd as sm, oracle_binding as ob
    tc, m, n, coo = sm.mm_read_coo(ob.fixture_path(a.matrix))
    rp, ci, v = sm.csr_from_coo(coo, m)
    copies = max(1, (1 << a.rows_log2) // m)
    RP, CI, V = tile_csr(rp, ci, v, m, n, copies)
    rows, cols, nnz = m * copies, n * copies, int(RP[-1])
    print("# %s x %d copies: rows=%d nnz=%d" % (a.matrix, copies, rows, nnz), flush=True)
    A = sm.CsrMatrix(rows, cols, torch.from_numpy(RP).cuda(), torch.from_numpy(CI).cuda(), torch.from_numpy(V).cuda())
    x = torch.ones(cols, dtype=torch.float64, device="cuda"); y = torch.empty(rows, dtype=torch.float64, device="cuda")
    ybase = ob.csr_spmv(rp, ci, v, np.ones(n))
    for var in a.variants.split(","):
        fam, par = var.split(":"); A.set_kernel({"stream": 2, "vector": 1, "carry": 3}[fam], int(par)); name, nbytes = A.describe()
        for _ in range(100): A.spmv(x, y)
        torch.cuda.synchronize()
        err = float((y.view(copies, m) - torch.from_numpy(ybase).cuda()[None, :]).abs().max())
        tms = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps): A.spmv(x, y, stream=torch.cuda.current_stream())
            e1.record(); torch.cuda.synchronize()
            tms.append(e0.elapsed_time(e1) / a.steps)
        ms = sorted(tms)[2]; gbs = nbytes / ms * 1e-6
        name += " min %.4f" % min(tms)
        print("%-22s %8.4f ms  %8.1f GB/s  %5.1f %% of 8 TB/s  %7.1f GFLOP/s  max|y - tile(y_base)| %.1e" % (name, ms, gbs, gbs / 80.0, 2 * nnz / ms * 1e-6, err), flush=True)
if __name__ == "__main__":
    main()
