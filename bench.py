#!/usr/bin/env python3
"""bench.py -- the headline benchmark: fp64 SpMV on MI355X, CSR (and TJDS beside it).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one product y = A x over the whole (sharded) matrix, inputs resident in
HBM.  Default workload: the memplus-shaped synthetic of SURVEY 8(d) at 2^24 rows
(~119 M entries, ~1.8 GB of algorithmic traffic -- memplus.mtx itself is 1.9 MB and
lives in L2, so it says nothing about HBM), CSR, x = ones like the reference
(main-cli.c:368-369).  With N > 1 the fixed matrix is cut into N row blocks, one
process per GPU, and a step is the local product plus the RCCL all-gather of the y
blocks over xGMI (strong scaling).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against HBM
(algorithmic bytes of SURVEY 8(d) / measured time per launch); `cpu_baseline` is the
reference's serial loop (the C oracle's restatement of main-cli.c:410-416) timed on
this box's host, one thread.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="memplus_shaped", choices=["memplus_shaped", "uniform32"])
    ap.add_argument("--rows-log2", type=int, default=24, help="memplus_shaped: total rows = 2^k")
    ap.add_argument("--rows", type=int, default=10_000_000, help="uniform32: total rows (BASELINE config 4)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "stream", "vector"])
    ap.add_argument("--kernel-param", type=int, default=0)
    ap.add_argument("--x", default="ones", choices=["ones", "random"])
    ap.add_argument("--no-tjds", action="store_true", help="skip the TJDS leg (extra.tjds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=0, help="0 = sized for about 15 s")
    ap.add_argument("--no-allgather", action="store_true", help="N > 1: time the local products only")
    return ap.parse_args()


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def timed_region(torch, dist, world, steps, body):
    """barrier + sync, `steps` x body(), sync + barrier; returns (wall seconds max over ranks, event ms)."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        body()
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([wall, ev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(t[0]), float(t[1])
    return wall, ev_ms


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with torch.distributed.run" % args.gpus)
        args.gpus = world

    import torch
    import torch.distributed as dist

    import smvp_toolkit_amd as sm
    from smvp_toolkit_amd import sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev_name, cus, hbm = sm.device_info(local_rank)

    # ------------------------------------------------------------ the workload
    if args.workload == "memplus_shaped":
        kind, seed, param = sm.SYNTH_MEMPLUS_SHAPED, 12345, 0
        rows_total = cols_total = 1 << args.rows_log2
        wl_name = "memplus_shaped_synthetic rows=2^%d seed=%d" % (args.rows_log2, seed)
    else:
        kind, seed, param = sm.SYNTH_UNIFORM, 2024, 32
        rows_total = cols_total = args.rows
        wl_name = "uniform32_synthetic rows=%d seed=%d" % (rows_total, seed)
    bounds = sharding.equal_row_bounds(rows_total, world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    t_gen = time.perf_counter()
    row_ptr, col_ind, val = sm.synth_csr(kind, seed, rows_total, cols_total, param, r0, r1,
                                         threads=max(1, min(16, (os.cpu_count() or 8) // max(1, world))))
    rows_local, nnz_local = r1 - r0, int(row_ptr[-1])
    log(rank, "generated rows [%d, %d): %d entries in %.1f s" % (r0, r1, nnz_local, time.perf_counter() - t_gen))

    d_row_ptr = torch.from_numpy(row_ptr).cuda()
    d_col_ind = torch.from_numpy(col_ind).cuda()
    d_val = torch.from_numpy(val).cuda()
    A = sm.CsrMatrix(rows_local, cols_total, d_row_ptr, d_col_ind, d_val, device=local_rank)
    if args.kernel != "auto" or args.kernel_param:
        A.set_kernel({"auto": 0, "vector": 1, "stream": 2}[args.kernel], args.kernel_param)
    kernel_name, alg_bytes_local = A.describe()

    if args.x == "ones":
        x_host = np.ones(cols_total)
    else:
        x_host = np.random.default_rng(67890).random(cols_total)
    d_x = torch.from_numpy(x_host).cuda()
    d_y_full = torch.zeros(rows_total, dtype=torch.float64, device="cuda")
    d_y = d_y_full[r0:r1] if world > 1 else d_y_full
    stream = torch.cuda.current_stream()

    def spmv_only():
        A.spmv(d_x, d_y, stream=stream)

    def step():
        A.spmv(d_x, d_y, stream=stream)
        if world > 1 and not args.no_allgather:
            dist.all_gather_into_tensor(d_y_full, d_y)

    # correctness gate before any timing: an independent host computation of this block
    step()
    torch.cuda.synchronize()
    host = np.add.reduceat(val * x_host[col_ind], row_ptr[:-1]) * (np.diff(row_ptr) > 0)
    scale = np.add.reduceat(np.abs(val * x_host[col_ind]), row_ptr[:-1]) * (np.diff(row_ptr) > 0)
    got = d_y_full[r0:r1].cpu().numpy()
    worst = float((np.abs(got - host) / np.maximum(scale, 1e-300)).max())
    if not np.all(np.abs(got - host) <= 1e-9 * scale):
        raise SystemExit("rank %d: product is wrong (max normwise error %g)" % (rank, worst))
    if world > 1 and not args.no_allgather:
        chk = d_y_full.sum().item()      # every rank must hold the same gathered vector
        t = torch.tensor([chk, -chk], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if float(t[0]) != -float(t[1]):
            raise SystemExit("all-gathered y differs between ranks")
    log(rank, "correct: max |dy| / sum|a x| = %.2e over %d local rows (%s)" % (worst, rows_local, kernel_name))

    # ------------------------------------------------------------ the timed run
    for _ in range(args.warmup):
        step()
    wall, ev_ms = timed_region(torch, dist, world, args.steps, step)
    ms_per_step = wall * 1e3 / args.steps

    t_nnz = torch.tensor([nnz_local, alg_bytes_local], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t_nnz)
    nnz_total, alg_bytes_total = float(t_nnz[0]), float(t_nnz[1])
    gflops = 2.0 * nnz_total / (wall / args.steps) * 1e-9

    # dominant kernel alone (no collective in the window), HIP events on the launch stream
    for _ in range(3):
        spmv_only()
    _, k_ms = timed_region(torch, dist, world, args.steps, spmv_only)
    k_ms_per_launch = k_ms / args.steps
    achieved = alg_bytes_local / (k_ms_per_launch * 1e-3) * 1e-9          # GB/s, this rank's kernel
    roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "alg_bytes_per_launch": alg_bytes_local, "ms_per_launch": round(k_ms_per_launch, 5),
                "note": "per launch = csr kernel + its carry fix-up; traffic: see profiles/ (PMC pass)"}

    extra = {"device": dev_name, "compute_units": cus, "nnz": int(nnz_total), "rows": rows_total,
             "alg_bytes_per_step": alg_bytes_total, "x": args.x,
             "whole_job_GBps": round(alg_bytes_total / (wall / args.steps) * 1e-9, 1)}
    if world > 1:
        extra["spmv_only_GFLOPs_per_rank_max_time"] = round(2.0 * nnz_total / (k_ms_per_launch * 1e-3) * 1e-9, 1)
        extra["allgather_in_step"] = not args.no_allgather
        extra["y_bytes_gathered"] = rows_total * 8

    # ------------------------------------------------------------ TJDS beside it
    if not args.no_tjds and world == 1:
        try:
            t0 = time.perf_counter()
            coo = np.zeros(nnz_local, dtype=sm.COO_DTYPE)
            coo["row"] = np.repeat(np.arange(rows_local, dtype=np.int32), np.diff(row_ptr))
            coo["col"], coo["val"] = col_ind, val
            tj = sm.tjds_from_coo(coo, rows_local, cols_total)
            del coo
            T = sm.TjdsMatrix(tj, device=local_rank)
            tname, tbytes = T.describe()
            log(rank, "TJDS built in %.1f s: %d jagged diagonals" % (time.perf_counter() - t0, tj.num_diag))
            d_yt = torch.empty(rows_local, dtype=torch.float64, device="cuda")
            T.set_x(d_x, stream=stream)

            def tjds_step():
                T.zero_y(d_yt, stream=stream)       # the scatter needs y = 0 (main-cli.c:1008); counted in the step
                T.spmv(d_yt, stream=stream)

            tjds_step()
            torch.cuda.synchronize()
            terr = float(((d_yt - d_y).abs().cpu().numpy() / np.maximum(scale, 1e-300)).max())
            tsteps = max(5, args.steps // 10)
            _, t_ms = timed_region(torch, dist, 1, tsteps, tjds_step)
            t_ms /= tsteps
            extra["tjds"] = {"kernel": tname, "ms_per_step": round(t_ms, 4), "num_diag": tj.num_diag,
                             "GFLOPs": round(2.0 * nnz_local / (t_ms * 1e-3) * 1e-9, 1),
                             "achieved_GBps": round(tbytes / (t_ms * 1e-3) * 1e-9, 1),
                             "frac_of_hbm_peak": round(tbytes / (t_ms * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4),
                             "max_normwise_diff_vs_csr": terr, "steps": tsteps,
                             "note": "step = memset(y) + scatter kernel (fp64 atomics)"}
            T.close()
        except Exception as e:  # the TJDS leg is informational; never lose the headline line over it
            extra["tjds"] = {"error": str(e)}

    # ------------------------------------------------------------ CPU baseline
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding as ob          # the checker, used here only as the CPU baseline leg

        _, probe = ob.csr_timed(row_ptr, col_ind, val, x_host, 1)
        iters = args.cpu_iters or int(max(2, min(50, round(15000.0 / max(probe[0], 1e-3)))))
        y_cpu, ms = ob.csr_timed(row_ptr, col_ind, val, x_host, iters)
        cpu_ok = bool(np.all(np.abs(y_cpu - got) <= 1e-9 * scale))
        model = ""
        try:
            model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        cpu = {"value": round(2.0 * nnz_local / (ms.mean() * 1e-3) * 1e-9, 3), "unit": "GFLOP/s", "cores": 1,
               "kind": "port", "host_cores_total": os.cpu_count(), "host_cpu": model,
               "GBps": round(alg_bytes_local / (ms.mean() * 1e-3) * 1e-9, 2), "ms_per_product": round(float(ms.mean()), 2),
               "sample": "the full workload matrix, %d products of the serial loop (oracle restatement of "
                         "main-cli.c:410-416, gcc -O3, y reset outside the window)" % iters,
               "agrees_with_gpu": cpu_ok}

    if rank == 0:
        line = {
            "metric": "fp64 CSR SpMV GFLOP/s (2*nnz flop per product; achieved HBM GB/s in roofline)",
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl_name + ", CSR, x=%s" % args.x +
                       (", %d row blocks + RCCL all-gather of y" % world if world > 1 and not args.no_allgather else ""),
                       "format": "csr", "kernel": kernel_name, "nnz": int(nnz_total), "rows": rows_total,
                       "sharding": "row-block x%d" % world},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    A.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
