#!/usr/bin/env python3
"""One process: the headline product's time beside the addresses of its arrays (is the fast / slow mode of profiles/r06_headline_variance.txt
a matter of where the allocations land?).  Run several times in one gpurun call.  --arena N: allocate N GB first and free them (shifts the layout)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--arena", type=float, default=0.0)
ap.add_argument("--hold", type=float, default=0.0, help="GB allocated first and KEPT")
a = ap.parse_args()
import torch
import smvp_toolkit_amd as sm
from smvp_toolkit_amd import sharding
import bench_core as core
keep = None
if a.hold:
    keep = torch.empty(int(a.hold * (1 << 30)), dtype=torch.uint8, device="cuda")
if a.arena:
    t = torch.empty(int(a.arena * (1 << 30)), dtype=torch.uint8, device="cuda"); del t; torch.cuda.empty_cache()
blk = core.build_block(sm, sharding, "memplus_tiled", argparse.Namespace(copies=944, scaling="strong"), 0, 1)
rp, ci, v = (torch.from_numpy(blk[k]).cuda() for k in ("row_ptr", "col_ind", "val"))
x = torch.ones(blk["cols_total"], dtype=torch.float64, device="cuda")
y = torch.empty(blk["rows"], dtype=torch.float64, device="cuda")
A = sm.CsrMatrix(blk["rows"], blk["cols_total"], rp, ci, v)
st = torch.cuda.current_stream()
for _ in range(150):
    A.spmv(x, y, stream=st)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        A.spmv(x, y, stream=st)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 50)
free, total = torch.cuda.mem_get_info()
print("ms %.4f  (min %.4f max %.4f)  val 0x%x col 0x%x rp 0x%x x 0x%x y 0x%x  free %.1f GB" % (sorted(ts)[2], min(ts), max(ts), v.data_ptr(), ci.data_ptr(), rp.data_ptr(), x.data_ptr(), y.data_ptr(), free / 2**30), flush=True)
