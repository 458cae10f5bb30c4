#!/usr/bin/env python3
"""bench.py -- the headline benchmark: fp64 SpMV on MI355X, CSR (and TJDS beside it).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one product y = A x over the whole (sharded) matrix, inputs resident in
HBM, CSR, x = ones like the reference (main-cli.c:368-369).  memplus.mtx itself is
1.9 MB and lives in L2, so it says nothing about HBM; the workloads are HBM-sized
matrices with memplus's shape (DESIGN.md "Workloads" has the reasoning):

  memplus_tiled   (default, the headline) memplus.mtx replicated 944x along the diagonal,
                  kron(I_944, memplus): 16.76 M rows, 119 M entries, 1.77 GB of
                  algorithmic traffic.  It is the EXACT-STRUCTURE SUBSTITUTE for the random
                  "memplus-shaped" model of SURVEY 8(d): every structural property of memplus
                  is kept exactly (row lengths, symmetry, its 165 hub rows/columns, all entries
                  within 17757 of the diagonal) and y is checkable at full size against
                  the reference's own committed memplus report.
  memplus_shaped  the random model of SURVEY 8(d) itself: memplus's row-length histogram
                  and band profile, entries beyond distance 4096 uniform over ALL 2^24
                  columns.  Always measured too (extra.survey_random_model): it is bound
                  by the chip's random-gather rate, not by HBM.
  uniform32       BASELINE config 4: 10 M x 10 M, 32 uniform entries per row.  Measured at
                  EVERY N (extra.config4): the 1 -> 8 GPU curve of BASELINE.md section 4 is
                  written on this matrix -- local products alone, products + all-gather of y,
                  and the chunked form that sends chunk c while chunk c+1 is multiplied.
  pwt_tiled       pwt.mtx x459 (extra.pwt_tiled, N = 1).

With N > 1 the headline matrix is cut into N row blocks, one process per GPU, and a
step is the local product plus the RCCL all-gather of the y blocks over xGMI
(strong scaling).

Started as plain `python bench.py --gpus N` (N > 1, no RANK / WORLD_SIZE in the environment: the driver's command shape)
the script is its own launcher: before torch or HIP are touched it starts the N ranks as child processes and relays
rank 0's line (spawn_ranks).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against HBM
(algorithmic bytes of SURVEY 8(d) / measured time per launch); `cpu_baseline` is the
reference's serial loop (the C oracle's restatement of main-cli.c:410-416) timed on
this box's host, one thread.  The driver's record keeps the SCALAR keys of `roofline` (and of
`cpu_baseline` / `config`) and drops nested objects: every figure a record needs -- the other kernels'
fractions, traffic over algorithmic bytes, config 4's t1 / tN / speed-up keys, the C layer's legs and exchange
times, the sample matrices' microseconds, what the communicator reports -- is therefore also a flat scalar in
`roofline` (flat_keys); `roofline.others` and `extra` carry the detail for a human reader.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
TOL = 1e-9              # row-normwise: |dy| <= TOL * sum_j |a_rj x_j|


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="memplus_tiled", choices=["memplus_tiled", "pwt_tiled", "memplus_shaped", "uniform32"])
    ap.add_argument("--copies", type=int, default=0, help="memplus_tiled / pwt_tiled: diagonal blocks (0 = 944 / 459 -> 16.76 M rows)")
    ap.add_argument("--rows-log2", type=int, default=24, help="memplus_shaped: total rows = 2^k")
    ap.add_argument("--rows", type=int, default=10_000_000, help="uniform32: total rows (BASELINE config 4)")
    ap.add_argument("--format", default="csr", choices=["csr", "tjds"],
                    help="storage format of the timed product (the other one is reported in extra at N = 1)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "stream", "vector", "stream-carry", "colsweep", "binned"])
    ap.add_argument("--kernel-param", type=int, default=0)
    ap.add_argument("--x", default="ones", choices=["ones", "random"])
    ap.add_argument("--no-tjds", action="store_true", help="skip the TJDS leg (extra.tjds)")
    ap.add_argument("--no-random-model", action="store_true", help="skip extra.survey_random_model")
    ap.add_argument("--no-samples", action="store_true", help="skip extra.sample_matrices (BASELINE configs 2, 3, 5)")
    ap.add_argument("--no-config4", action="store_true", help="skip extra.config4 (10 M x 32/row, every N)")
    ap.add_argument("--no-pwt-tiled", action="store_true", help="skip extra.pwt_tiled (N = 1)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="config 4, N > 1: row chunks per rank for the overlapped all-gather (0 = chosen from this run's own "
                         "measurements of 1 / 2 / 4 chunks, sharding.choose_chunks)")
    ap.add_argument("--no-eighth", action="store_true", help="config 4, N = 1: skip the 1 / 2 / 4-chunk timing of one rank's share at N = 8")
    ap.add_argument("--config4-kernel", default="auto", choices=["auto", "colsweep", "tile"],
                    help="config 4: kernel of the step timings (auto = what the library picks: the column sweep; the tile "
                         "kernel's product time is reported either way)")
    ap.add_argument("--no-c-layer", action="store_true",
                    help="skip roofline.others.config4_c_layer (the C ABI's own sharded product, smvp_sharded_spmv, on all GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (rocprofv3 --pmc child passes); fall back to profiles/")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the inner run of a --pmc child pass
    ap.add_argument("--c-layer-child", type=int, default=0, help=argparse.SUPPRESS)   # child process: the C layer alone on this many GPUs
    ap.add_argument("--c-layer-budget", type=float, default=240.0,
                    help="N > 1: wall-clock seconds the C-layer leg (a child process of rank 0) may take before it is given up")
    ap.add_argument("--launch-budget", type=float, default=1500.0,
                    help="--gpus N > 1 started WITHOUT a launcher: wall-clock seconds the N rank processes this script starts "
                         "for itself may take before they are ended")
    ap.add_argument("--cpu-iters", type=int, default=0, help="0 = sized for about 15 s")
    ap.add_argument("--no-allgather", action="store_true", help="N > 1: time the local products only")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the same matrix cut into N row blocks (default); weak = N times the matrix")
    return ap.parse_args()


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# Untimed products in front of a leg's timed region.  A leg follows seconds of host work (building and checking its matrix),
# and the first milliseconds of device work after such a pause run slower than the steady state (measured: the random-model
# leg read 0.677 ms with 3 warm-up products and 50 timed ones, 0.650 ms with 20 and 200 in the same process on the same box,
# profiles/r04_binned_measured.txt section 13; round 5: whichever of the headline's two timed regions came first behind the
# driver's --warmup 5 = 1.5 ms read 7 % slower than the other -- 0.310-0.321 against 0.291-0.294 ms): every leg runs about
# PREWARM_MS of untimed products first (prewarm); the W warm-up steps the contract names come on top, in front of the K timed steps.
PREWARM_MS = 40.0  # the headline leg: untimed products in front of its two timed regions (the W warm-up steps of the contract come on top)
WARM_SHORT = 20   # products of < 1 ms
WARM_LONG = 8     # products of a few ms (config 4)


def timed_region(torch, dist, world, steps, body):
    """barrier + sync, `steps` x body(), sync + barrier -> (wall seconds, HIP-event ms), both MAX over ranks."""
    import gc

    collecting = gc.isenabled()
    gc.disable()        # no collector pause between the two clock readings
    try:
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
    finally:
        if collecting:
            gc.enable()
    ev_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([wall, ev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(t[0]), float(t[1])
    return wall, ev_ms


def prewarm(torch, body, ms=None):
    """About `ms` (default PREWARM_MS) of untimed device work in front of a leg's timed region: three probe runs timed with a
    HIP event pair, then as many more as fill the time (at most 400)."""
    ms = PREWARM_MS if ms is None else ms
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        body()
    e1.record()
    torch.cuda.synchronize()
    each = max(e0.elapsed_time(e1) / 3.0, 1e-3)
    for _ in range(int(min(400, max(0, ms / each - 3)))):
        body()


GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_file(kind, name):
    """Plain path of a committed data fixture (tests/golden/<kind>/<name>[.gz]); .gz files are inflated once into a
    per-user temp dir.  Ranks may race here: each writes its own temp file and renames it into place."""
    import gzip
    import shutil
    import tempfile

    plain = os.path.join(GOLDEN, kind, name)
    if os.path.exists(plain):
        return plain
    cache = os.path.join(tempfile.gettempdir(), "smvp_bench_cache_%d" % os.getuid())
    os.makedirs(cache, exist_ok=True)
    out = os.path.join(cache, name)
    if not os.path.exists(out):
        fd, tmp = tempfile.mkstemp(prefix=name + ".", dir=cache)
        with gzip.open(plain + ".gz", "rb") as src, os.fdopen(fd, "wb") as dst:
            shutil.copyfileobj(src, dst)
        os.replace(tmp, out)
    return out


def report_y_lines(name):
    lines = open(golden_file("reports", name)).read().split("\n")
    return lines[lines.index("[") + 1:lines.index("]")]


def build_block(sm, sharding, workload, args, rank, world):
    """This rank's row block of the workload -> dict with host CSR arrays and a description."""
    t0 = time.perf_counter()
    if workload in ("memplus_tiled", "pwt_tiled"):
        base_name, base_report, base_copies = (("memplus.mtx", "smvp-toolbox_report_CSR_1615284663.txt", 944) if workload == "memplus_tiled"
                                               else ("pwt.mtx", "smvp-toolbox_report_CSR_1615284671.txt", 459))
        tc, m, n, coo = sm.mm_read_coo(golden_file("sample-data", base_name))
        rp, ci, v = sm.csr_from_coo(coo, m)
        total = (getattr(args, "copies", 0) or base_copies) * (world if getattr(args, "scaling", "strong") == "weak" else 1)
        copies = total - total % world if total >= world else world
        c0, c1 = copies * rank // world, copies * (rank + 1) // world
        row_ptr, col_ind, val = sharding.tile_block_diagonal(rp, ci, v, n, c0, c1)
        blk = dict(rows_total=m * copies, cols_total=n * copies, r0=m * c0, r1=m * c1,
                   bounds=np.array([m * (copies * g // world) for g in range(world + 1)], dtype=np.int64),
                   name=("memplus.mtx x%d block-diagonal (kron(I_%d, memplus)) -- the exact-structure substitute for the "
                         "SURVEY 8(d) random memplus-shaped model, which is in extra.survey_random_model" % (copies, copies))
                   if workload == "memplus_tiled" else
                   "pwt.mtx x%d block-diagonal (kron(I_%d, pwt), stored triangle only like the reference)" % (copies, copies),
                   base=(m, n, rp, ci, v, c1 - c0, base_report))
    else:
        if workload == "memplus_shaped":
            kind, seed, param = sm.SYNTH_MEMPLUS_SHAPED, 12345, 0
            rows_total = cols_total = (1 << args.rows_log2) * (world if getattr(args, "scaling", "strong") == "weak" else 1)
            name = "memplus_shaped random model (SURVEY 8(d)) rows=%d seed=%d" % (rows_total, seed)
        else:
            kind, seed, param = sm.SYNTH_UNIFORM, 2024, 32
            rows_total = cols_total = args.rows * (world if getattr(args, "scaling", "strong") == "weak" else 1)
            name = "uniform 32 entries/row rows=%d seed=%d" % (rows_total, seed)
        bounds = sharding.equal_row_bounds(rows_total, world)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        row_ptr, col_ind, val = sm.synth_csr(kind, seed, rows_total, cols_total, param, r0, r1,
                                             threads=max(1, min(16, (os.cpu_count() or 8) // max(1, world))))
        blk = dict(rows_total=rows_total, cols_total=cols_total, r0=r0, r1=r1, name=name, base=None, bounds=bounds)
    blk.update(row_ptr=row_ptr, col_ind=col_ind, val=val, nnz=int(row_ptr[-1]), rows=blk["r1"] - blk["r0"])
    log(rank, "%s: rows [%d, %d), %d entries, built in %.1f s" % (blk["name"], blk["r0"], blk["r1"], blk["nnz"],
                                                                  time.perf_counter() - t0))
    return blk


def host_check(blk, x_host, got):
    """Independent host computation of this block's y (numpy, not the oracle); returns (ok, worst, scale)."""
    row_ptr, col_ind, val = blk["row_ptr"], blk["col_ind"], blk["val"]
    nonempty = np.diff(row_ptr) > 0
    prod = val * x_host[col_ind]
    starts = np.minimum(row_ptr[:-1], max(len(prod) - 1, 0))
    host = np.add.reduceat(prod, starts) * nonempty if len(prod) else np.zeros(blk["rows"])
    scale = np.add.reduceat(np.abs(prod), starts) * nonempty if len(prod) else np.zeros(blk["rows"])
    err = np.abs(got - host)
    return bool(np.all(err <= TOL * scale)), float((err / np.maximum(scale, 1e-300)).max()), scale


def measure_csr(torch, dist, sm, sharding_mod, blk, args, world, local_rank, rank, steps, warmup, collective):
    """Upload the block, check it, time `steps` products (+ all-gather), then the kernel alone."""
    fmt = getattr(args, "format", "csr")
    if fmt == "csr":
        d_row_ptr = torch.from_numpy(blk["row_ptr"]).cuda()
        d_col_ind = torch.from_numpy(blk["col_ind"]).cuda()
        d_val = torch.from_numpy(blk["val"]).cuda()
        A = sm.CsrMatrix(blk["rows"], blk["cols_total"], d_row_ptr, d_col_ind, d_val, device=local_rank, first_row=blk["r0"])
        if args.kernel != "auto" or args.kernel_param:
            A.set_kernel({"auto": 0, "vector": 1, "stream": 2, "stream-carry": 3, "colsweep": 4, "binned": 5}[args.kernel], args.kernel_param)
    else:   # TJDS of this rank's row block, built on the GPU from the block's entries
        coo = np.zeros(blk["nnz"], dtype=sm.COO_DTYPE)
        coo["row"] = np.repeat(np.arange(blk["rows"], dtype=np.int32), np.diff(blk["row_ptr"]))
        coo["col"], coo["val"] = blk["col_ind"], blk["val"]
        d_coo = torch.from_numpy(coo.view(np.uint8)).cuda()
        del coo
        d_row_ptr = d_col_ind = d_val = None
        A = sm.TjdsMatrix(sm.tjds_from_coo_device(d_coo, blk["rows"], blk["cols_total"], blk["nnz"]), device=local_rank)
        del d_coo
    kernel_name, alg_bytes = A.describe()
    launches = A.launches() if fmt == "csr" else 1
    pi = A.plan_info()
    plan = {"plan_bytes": pi["plan_bytes"], "matrix_bytes": pi["matrix_bytes"],
            "plan_over_matrix": round(pi["plan_bytes"] / max(1.0, pi["matrix_bytes"]), 3), "plan_build_ms": round(pi["build_ms"], 1)}
    if fmt == "tjds":
        plan["value_cache"] = _value_cache(A, blk["nnz"])

    x_host = np.ones(blk["cols_total"]) if args.x == "ones" else np.random.default_rng(67890).random(blk["cols_total"])
    d_x = torch.from_numpy(x_host).cuda()
    d_y_full = torch.zeros(blk["rows_total"], dtype=torch.float64, device="cuda")
    # one rank: the product writes the full vector; several: each rank's block has its own buffer and the
    # all-gather assembles the full y on every GPU (no aliasing between send and receive buffers)
    d_y = d_y_full if not (world > 1 or dist.is_initialized()) else torch.zeros(blk["rows"], dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()
    gather = collective and (world > 1 or dist.is_initialized())

    if fmt == "tjds":
        A.set_x(d_x, stream=stream)      # the operand permutation is set-up, like main-cli.c:907-923

    def spmv_only():
        if fmt == "csr":
            A.spmv(d_x, d_y, stream=stream)
        else:
            A.spmv(d_y, stream=stream)

    bounds = blk["bounds"]

    def step():
        spmv_only()
        if gather:      # equal row blocks: one all_gather_into_tensor straight into the full y
            sharding_mod.allgather_y(dist, d_y, d_y_full, bounds)

    # correctness gate before any timing
    step()
    torch.cuda.synchronize()
    got = d_y.cpu().numpy()
    ok, worst, scale = host_check(blk, x_host, got)
    if not ok:
        raise SystemExit("rank %d: product is wrong on %s (max normwise error %g)" % (rank, blk["name"], worst))
    golden = None
    if blk["base"] is not None and args.x == "ones":
        # full-size parity against the reference's own golden vector: y must be tile(y_memplus), and y_memplus is
        # printed with "%g" in the committed report output-test/smvp-toolbox_report_CSR_1615284663.txt.  No oracle
        # here: the base product runs on the GPU too and is compared with the report's text.
        m, n, rp, ci, v, ncopies, base_report = blk["base"]
        B = sm.CsrMatrix(m, n, rp, ci, v, device=local_rank)
        d_yb = torch.empty(m, dtype=torch.float64, device="cuda")
        B.spmv(d_x[:n], d_yb, stream=stream)
        torch.cuda.synchronize()
        B.close()
        y_base = d_yb.cpu().numpy()
        want = report_y_lines(base_report)
        sc = np.add.reduceat(np.abs(v), rp[:-1])
        short = np.diff(rp) <= 32                       # summed left to right by one lane: bit-exact => same "%g" text
        text_ok = all(("%g" % y_base[i]) == want[i] for i in np.flatnonzero(short))
        num_ok = bool(np.all(np.abs(y_base - np.array([float(s) for s in want])) <= 1e-5 * sc + 1e-300))
        tiles_ok = bool(np.all(np.abs(got.reshape(ncopies, m) - y_base[None, :]) <= TOL * sc[None, :]))
        if not (text_ok and num_ok and tiles_ok):
            raise SystemExit("rank %d: y is not tile(y of the committed report %s)" % (rank, base_report))
        golden = {"y_equals_tiled_reference_memplus_y" if "1615284663" in base_report else "y_equals_tiled_reference_pwt_y": True,
                  "report_text_equal_on_rows_upto_32_entries": int(short.sum()), "rows_per_copy": int(m)}
    if gather:
        chk = float(d_y_full.sum().item())
        t = torch.tensor([chk, -chk], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if float(t[0]) != -float(t[1]):
            raise SystemExit("all-gathered y differs between ranks")
    log(rank, "correct: max |dy| / sum|a x| = %.2e over %d local rows (%s)" % (worst, blk["rows"], kernel_name))

    # the kernel alone first (HIP events on the launch stream, no collective), then the W warm-up steps and the K timed steps of
    # the contract: the first milliseconds of device work after the seconds of host work above run slower than the steady state
    # (prewarm), and W is the caller's -- the driver asks for 5 steps = 1.5 ms
    prewarm(torch, spmv_only)
    _, k_ms = timed_region(torch, dist, world, steps, spmv_only)
    for _ in range(warmup):
        step()
    wall, _ = timed_region(torch, dist, world, steps, step)

    tot = torch.tensor([blk["nnz"], alg_bytes], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tot)
    res = dict(kernel=kernel_name, launches=launches, alg_bytes_local=alg_bytes, alg_bytes_total=float(tot[1]), nnz_total=float(tot[0]),
               wall_per_step=wall / steps, kernel_ms=k_ms / steps, worst=worst, golden=golden, scale=scale, got=got, plan=plan,
               x_host=x_host, d_x=d_x, d_y=d_y, A=A, keep=(d_row_ptr, d_col_ind, d_val, d_y_full))
    return res


def _config4_block(torch, sm, rows, ranges, local_rank, threads):
    """CSR handles of the row ranges `ranges` of BASELINE config 4 (one per chunk) + what checks them on the host."""
    mats, nnz_local, alg_local, checks = [], 0, 0.0, []
    for r0, r1 in ranges:
        rp, ci, v = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, rows, rows, 32, r0, r1, threads=threads)
        A = sm.CsrMatrix(r1 - r0, rows, torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(v).cuda(),
                         device=local_rank, first_row=r0)
        n = int(rp[-1])
        nnz_local += n
        alg_local += 12.0 * n + 4.0 * (r1 - r0 + 1) + 8.0 * (r1 - r0)     # x is counted once per rank, by the caller, not once per chunk
        # x = ones (the reference's operand): y = the row sums of val, computed independently on the host
        host = np.add.reduceat(v, rp[:-1]) if n else np.zeros(r1 - r0)
        scale = np.add.reduceat(np.abs(v), rp[:-1]) if n else np.zeros(r1 - r0)
        checks.append((r0, r1, host, scale))
        mats.append(A)
        del rp, ci, v
    return mats, nnz_local, alg_local + 8.0 * rows, checks


def measure_config4(torch, dist, sm, sharding_mod, args, world, local_rank, rank, steps):
    """BASELINE config 4 (10 M x 10 M, 32 uniform entries per row, seed 2024) on `world` GPUs -> dict for extra.config4.

    Row ownership is block-cyclic (sharding.cyclic_chunk_rows): every rank holds `chunks` row chunks, each its own CSR
    handle; the all-gather of chunk c lands as one contiguous run of the full y.  Three timings, all max over ranks:
    local products only; products, then the all-gathers (nothing overlapped); each chunk's all-gather issued
    asynchronously behind its product (chunk c travels while chunk c+1 is multiplied).  The chunk count is not a
    constant: the column sweep pays for every chunk (each pulls all of x into the L2s again), so it is chosen from this
    run's own measurements (sharding.choose_chunks) unless --chunks names it.  The same keys at every N: t1_ms (the
    whole matrix on ONE GPU, measured in this run), tN_step_ms, speedup_overlapped, speedup_after.
    """
    rows = args.rows
    gather = world > 1 or dist.is_initialized()     # SMVP_FORCE_DIST rehearses the chunked path with one rank
    threads = max(1, min(64, (os.cpu_count() or 8) // max(1, world)))
    d_x = torch.ones(rows, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()

    def time_products(ex, mats, n):
        def product(c, out):
            r0, r1 = ex.ranges[c]
            if r1 > r0:
                mats[c].spmv(d_x, out, stream=stream)
        prewarm(torch, lambda: ex.step(product, overlap=False, gather=False))
        _, ev = timed_region(torch, dist, world, n, lambda: ex.step(product, overlap=False, gather=False))
        return ev / n

    # ---- how many chunks per rank
    if args.chunks > 0 or not gather:
        chunks = max(1, args.chunks) if gather else 1
        choice = {"chosen": chunks, "rule": "--chunks %d" % args.chunks if gather else "one GPU, no exchange: one chunk"}
    else:
        product_ms, gather_ms = {}, {}
        for c in (1, 2, 4):
            ex = sharding_mod.ChunkedExchange(torch, dist, rows, world, rank, c, "cuda")
            mats, _, _, _ = _config4_block(torch, sm, rows, ex.ranges, local_rank, threads)
            product_ms[c] = time_products(ex, mats, max(3, steps // 4))
            for _ in range(2):
                ex._gather(0, False)
            _, ev = timed_region(torch, dist, world, max(3, steps // 4), lambda: ex._gather(0, False))
            gather_ms[c] = ev / max(3, steps // 4)
            for A in mats:
                A.close()
            del mats, ex
            torch.cuda.empty_cache()
        choice = sharding_mod.choose_chunks(product_ms, 8.0 * rows / world, world, gather_ms)
        chunks = choice["chosen"]
        log(rank, "config 4: chunks per rank chosen from this run's measurements: %s" % json.dumps(choice))

    t0 = time.perf_counter()
    ex = sharding_mod.ChunkedExchange(torch, dist, rows, world, rank, chunks, "cuda")
    mats, nnz_local, alg_local, checks = _config4_block(torch, sm, rows, ex.ranges, local_rank, threads)
    kname = mats[0].describe()[0]
    plan = [A.plan_info() for A in mats]
    log(rank, "config 4: rows %d, %d chunk(s) per rank, %d local entries, built in %.1f s" % (rows, chunks, nnz_local,
                                                                                             time.perf_counter() - t0))

    def product(c, out):
        r0, r1 = ex.ranges[c]
        if r1 > r0:
            mats[c].spmv(d_x, out, stream=stream)

    def check():
        y_full = ex.step(product, overlap=True, gather=gather)
        torch.cuda.synchronize()
        worst = 0.0
        for c, (r0, r1, host, scale) in enumerate(checks):
            got = ex.local(c)[:r1 - r0].cpu().numpy()
            err = np.abs(got - host)
            if not np.all(err <= TOL * scale):
                raise SystemExit("rank %d: config 4 chunk %d is wrong (%s)" % (rank, c, mats[c].describe()[0]))
            worst = max(worst, float((err / np.maximum(scale, 1e-300)).max()) if len(err) else 0.0)
            if gather and not np.array_equal(y_full[r0:r1].cpu().numpy(), got):
                raise SystemExit("rank %d: the gathered y does not hold this rank's chunk %d" % (rank, c))
        return y_full, worst

    y_full, worst = check()
    if gather:   # every rank must hold the same full vector
        chk = float(y_full.sum().item())
        t = torch.tensor([chk, -chk], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if float(t[0]) != -float(t[1]):
            raise SystemExit("config 4: all-gathered y differs between ranks")

    def run(overlap, do_gather):
        if do_gather:      # (collectives: every rank the same number of calls)
            for _ in range(WARM_LONG):
                ex.step(product, overlap=overlap, gather=do_gather)
        else:
            prewarm(torch, lambda: ex.step(product, overlap=overlap, gather=do_gather))
        wall, ev = timed_region(torch, dist, world, steps, lambda: ex.step(product, overlap=overlap, gather=do_gather))
        return wall / steps * 1e3, ev / steps

    # What the library picks by itself: on this matrix the column sweep (deterministic: every row summed in ascending
    # column order, bit for bit the serial loop).  The tile kernel is timed beside it, products only.
    auto_kernel = mats[0].get_kernel()
    if args.config4_kernel == "colsweep":
        for A in mats:
            A.set_kernel(sm.CSR_KERNEL_COLSWEEP, 0)
    elif args.config4_kernel == "tile":
        for A in mats:
            A.set_kernel(sm.CSR_KERNEL_STREAM, 0)
    if args.config4_kernel != "auto":
        y_full, worst = check()
    kname = mats[0].describe()[0]
    launches = sum(A.launches() for A in mats)
    spread = mats[0].gather_spread()
    _, best_ms = run(False, False)
    y_first = ex.y_local.clone()        # this rank's chunks as the product left them
    check()
    if not torch.equal(y_first, ex.y_local):
        raise SystemExit("config 4: the product is not the same from run to run")
    tile_ms = best_ms
    if mats[0].get_kernel()[0] != sm.CSR_KERNEL_STREAM:
        saved = [A.get_kernel() for A in mats]
        for A in mats:
            A.set_kernel(sm.CSR_KERNEL_STREAM, 0)
        check()
        if not torch.equal(y_first, ex.y_local):    # 32 entries per row: both kernels sum every row in serial order
            raise SystemExit("config 4: the column sweep and the tile kernel differ")
        _, tile_ms = run(False, False)
        for A, (k, prm) in zip(mats, saved):
            A.set_kernel(k, prm)
        check()
    del y_first
    del checks
    tot = torch.tensor([nnz_local, alg_local], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tot)
    nnz, alg = float(tot[0]), float(tot[1])
    out = {"workload": "uniform 32 entries/row rows=%d seed=2024 (BASELINE config 4)" % rows, "rows": rows, "nnz": int(nnz),
           "n_gpus": world, "kernel": kname, "kernel_choice": args.config4_kernel,
           "auto_picks": {1: "vector", 2: "tile (stream)", 3: "tile (stream-carry)", 4: "column sweep", 5: "binned"}.get(auto_kernel[0]),
           "rows_per_workgroup": mats[0].get_kernel()[1], "launches_per_product": launches,
           "gather_spread_estimate": round(spread, 3), "chunks_per_rank": chunks, "chunks_chosen": chunks, "chunk_choice": choice,
           "steps": steps,
           "spmv_only_ms": round(best_ms, 4), "spmv_only_GFLOPs": round(2.0 * nnz / best_ms * 1e-6, 1),
           "max_normwise_error_vs_host": worst, "bit_identical_run_to_run": True, "bit_identical_to_tile_kernel": True,
           "x_gathers_per_second_G_per_gpu": round(nnz / best_ms * 1e-6 / world, 1),
           "tile_kernel_spmv_only_ms": round(tile_ms, 4), "tile_kernel_GFLOPs": round(2.0 * nnz / tile_ms * 1e-6, 1),
           "tile_kernel_x_gathers_per_second_G_per_gpu": round(nnz / tile_ms * 1e-6 / world, 1),
           "alg_bytes_per_product": alg,
           "plan": {"plan_bytes_local": sum(p["plan_bytes"] for p in plan), "matrix_bytes_local": sum(p["matrix_bytes"] for p in plan),
                    "plan_over_matrix": round(sum(p["plan_bytes"] for p in plan) / max(1.0, sum(p["matrix_bytes"] for p in plan)), 3),
                    "plan_build_ms": round(sum(p["build_ms"] for p in plan), 1)},
           "note": "uniform columns over an 80 MB x: with the tile kernel every x gather misses L2 and one GPU is bound by "
                   "its L2-miss gather rate (about 54 G/s, tools/gather_bench.hip), not by HBM bytes; the column-swept "
                   "kernel (AUTO's choice here; same bits as the serial loop) slides one L2-sized window over x"}
    if world == 1:
        out["frac_of_hbm_peak"] = round(alg / best_ms * 1e-6 / HBM_PEAK_GBS, 4)
        out["achieved_GBps"] = round(alg / best_ms * 1e-6, 1)
        out["tile_kernel_frac_of_hbm_peak"] = round(alg / tile_ms * 1e-6 / HBM_PEAK_GBS, 4)
    plain_ms = over_ms = best_ms
    if gather:
        plain_ms, _ = run(False, True)
        over_ms, _ = run(True, True)
        out.update(step_ms_products_then_allgather=round(plain_ms, 4), step_ms_overlapped=round(over_ms, 4),
                   step_GFLOPs_products_then_allgather=round(2.0 * nnz / plain_ms * 1e-6, 1),
                   step_GFLOPs_overlapped=round(2.0 * nnz / over_ms * 1e-6, 1), y_bytes_gathered=rows * 8,
                   exchange="block-cyclic row chunks, one all_gather_into_tensor per chunk (%s)" %
                            os.environ.get("SMVP_DIST_BACKEND", "nccl = RCCL over xGMI"))
    for A in mats:
        A.close()
    del mats, ex
    torch.cuda.empty_cache()

    # ---- the same keys at every N: the whole matrix on ONE GPU (t1_ms) against this N's step
    t1_ms = best_ms
    if world > 1:
        # rank 0 multiplies the whole matrix alone (3.8 GB + its plan fit one GPU) while the others wait at the barrier
        if rank == 0:
            ex1 = sharding_mod.ChunkedExchange(torch, dist, rows, 1, 0, 1, "cuda")
            m1, _, _, _ = _config4_block(torch, sm, rows, ex1.ranges, local_rank, max(1, min(64, os.cpu_count() or 8)))
            for _ in range(2):
                m1[0].spmv(d_x, ex1.local(0), stream=stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                m1[0].spmv(d_x, ex1.local(0), stream=stream)
            e1.record()
            torch.cuda.synchronize()
            t1_ms = e0.elapsed_time(e1) / steps
            m1[0].close()
            del m1, ex1
            torch.cuda.empty_cache()
        t = torch.tensor([t1_ms if rank == 0 else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t1_ms = float(t[0])
    out.update(t1_ms=round(t1_ms, 4), tN_step_ms=round(over_ms, 4), tN_step_after_ms=round(plain_ms, 4),
               tN_products_only_ms=round(best_ms, 4),
               speedup_overlapped=round(t1_ms / over_ms, 3), speedup_after=round(t1_ms / plain_ms, 3),
               speedup_products_only=round(t1_ms / best_ms, 3))

    # ---- N = 1: what one rank of eight would hold, cut into 1 / 2 / 4 chunks, measured here; the chunk count the model picks
    if world == 1 and not gather and not getattr(args, "no_eighth", False):
        try:
            product_ms = {}
            for c in (1, 2, 4):
                ranges = sharding_mod.cyclic_chunk_rows(rows, 8, c)[1][0]
                m8, _, _, _ = _config4_block(torch, sm, rows, ranges, local_rank, threads)
                bufs = [torch.empty(max(1, r1 - r0), dtype=torch.float64, device="cuda") for r0, r1 in ranges]

                def eighth():
                    for A, buf, (r0, r1) in zip(m8, bufs, ranges):
                        if r1 > r0:
                            A.spmv(d_x, buf, stream=stream)
                prewarm(torch, eighth)
                _, ev = timed_region(torch, dist, 1, steps, eighth)
                product_ms[c] = ev / steps
                for A in m8:
                    A.close()
                del m8, bufs
                torch.cuda.empty_cache()
            out["eighth_of_n8"] = sharding_mod.choose_chunks(product_ms, 8.0 * rows / 8, 8)
            out["eighth_of_n8"]["what"] = ("rank 0's share at N = 8 (block-cyclic, %d rows) multiplied on this one GPU as 1 / 2 / 4 chunks; "
                                           "the all-gather priced by the two link models (no second GPU here)" % (rows // 8))
            out["chunks_chosen_for_n8"] = out["eighth_of_n8"]["chosen"]
        except Exception as e:
            out["eighth_of_n8"] = {"error": str(e)}
    return out


def measure_config5(torch, dist, sm, sharding_mod, world, local_rank, rank, steps):
    """BASELINE config 5: pwt.mtx as stored (181 313 lower-triangle entries), CSR then TJDS back to back, row blocks
    balanced by entries over `world` GPUs, each product followed by the all-gather of y when world > 1 -> extra.config5_pwt.
    A 2.9 MB problem: more GPUs can only add the exchange to a 2 us product; reported as it comes out."""
    tc, m, n, coo = sm.mm_read_coo(golden_file("sample-data", "pwt.mtx"))
    rp, ci, v = sm.csr_from_coo(coo, m)
    bounds = sm.partition_rows(rp, world).astype(np.int64) if world > 1 else np.array([0, m], dtype=np.int64)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    lrp, lci, lv = sharding_mod.slice_csr(rp, ci, v, r0, r1)
    A = sm.CsrMatrix(r1 - r0, n, lrp, lci, lv, device=local_rank)
    lcoo = sm.make_coo(np.repeat(np.arange(r1 - r0), np.diff(lrp)), lci, lv)
    T = sm.TjdsMatrix(sm.tjds_from_coo(lcoo, r1 - r0, n), device=local_rank)
    d_x = torch.ones(n, dtype=torch.float64, device="cuda")
    pad = int(np.diff(bounds).max())
    y_c = torch.zeros(pad, dtype=torch.float64, device="cuda")
    y_t = torch.zeros(pad, dtype=torch.float64, device="cuda")
    y_full = torch.zeros(m, dtype=torch.float64, device="cuda")
    wire = torch.empty(world * pad, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()
    T.set_x(d_x, stream=stream)
    gather = world > 1 or dist.is_initialized()

    def csr_step():
        A.spmv(d_x, y_c, stream=stream)
        if gather:
            sharding_mod.allgather_y(dist, y_c[:r1 - r0], y_full, bounds, wire=wire)

    def tjds_step():
        T.spmv(y_t, stream=stream)
        if gather:
            sharding_mod.allgather_y(dist, y_t[:r1 - r0], y_full, bounds, wire=wire)

    want = np.array([float(s) for s in report_y_lines("smvp-toolbox_report_CSR_1615284671.txt")])
    for fn, buf in ((csr_step, y_c), (tjds_step, y_t)):
        fn()
        torch.cuda.synchronize()
        if not np.array_equal(buf[:r1 - r0].cpu().numpy(), want[r0:r1]) or (gather and not np.array_equal(y_full.cpu().numpy(), want)):
            raise SystemExit("rank %d: config 5 result differs from the reference's committed pwt report" % rank)
    out = {"workload": "pwt.mtx as stored, CSR then TJDS back to back (BASELINE config 5)", "n_gpus": world, "rows": m, "nnz": len(coo),
           "steps": steps, "y_equals_reference_report": True,
           "timing": "HIP events over %d back-to-back steps (max over ranks); the per-product device-timed figures of one GPU "
                     "are in extra.sample_matrices" % steps}
    for key, fn in (("csr", csr_step), ("tjds", tjds_step), ("csr_then_tjds", lambda: (csr_step(), tjds_step()))):
        for _ in range(5):
            fn()
        wall, ev = timed_region(torch, dist, world, steps, fn)
        out[key + "_ms_per_step"] = round(ev / steps, 6)
    out["exchange"] = ("all_gather_into_tensor of the y blocks after every product, blocks balanced by entries and padded"
                       if gather else "none (one GPU)")
    A.close()
    T.close()
    return out


def measure_pwt_tiled(torch, dist, sm, sharding_mod, local_rank, rank, steps):
    """pwt.mtx replicated 459x along the diagonal (16.76 M rows, 83 M stored entries): CSR and TJDS -> extra.pwt_tiled."""
    tc, m, n, coo = sm.mm_read_coo(golden_file("sample-data", "pwt.mtx"))
    rp, ci, v = sm.csr_from_coo(coo, m)
    copies = 459
    RP, CI, V = sharding_mod.tile_block_diagonal(rp, ci, v, n, 0, copies)
    rows, cols, nnz = m * copies, n * copies, int(RP[-1])
    A = sm.CsrMatrix(rows, cols, torch.from_numpy(RP).cuda(), torch.from_numpy(CI).cuda(), torch.from_numpy(V).cuda(), device=local_rank)
    kname, alg = A.describe()
    api = A.plan_info()
    d_x = torch.ones(cols, dtype=torch.float64, device="cuda")
    d_y = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()
    A.spmv(d_x, d_y, stream=stream)
    torch.cuda.synchronize()
    # pattern matrix, x = ones: y = tile(row lengths of pwt) exactly, which is also what the reference's committed
    # report output-test/smvp-toolbox_report_CSR_1615284671.txt prints
    want = report_y_lines("smvp-toolbox_report_CSR_1615284671.txt")
    y = d_y.cpu().numpy().reshape(copies, m)
    if not (np.array_equal(y[0], np.diff(rp).astype(np.float64)) and np.array_equal(y, np.tile(y[0], (copies, 1)))
            and all(("%g" % a) == b for a, b in zip(y[0], want))):
        raise SystemExit("pwt x%d: y is not tile(y_pwt of the committed report)" % copies)
    prewarm(torch, lambda: A.spmv(d_x, d_y, stream=stream))
    _, ms = timed_region(torch, dist, 1, steps, lambda: A.spmv(d_x, d_y, stream=stream))
    ms /= steps
    out = {"workload": "pwt.mtx x%d block-diagonal (kron(I_%d, pwt), stored triangle only like the reference)" % (copies, copies),
           "rows": rows, "nnz": nnz, "kernel": kname, "ms_per_launch": round(ms, 5), "alg_bytes_per_product": alg,
           "GFLOPs": round(2.0 * nnz / ms * 1e-6, 1),
           "achieved_GBps": round(alg / ms * 1e-6, 1), "frac_of_hbm_peak": round(alg / ms * 1e-6 / HBM_PEAK_GBS, 4),
           "y_equals_tiled_reference_pwt_y": True,
           "plan": {"plan_bytes": api["plan_bytes"], "matrix_bytes": api["matrix_bytes"],
                    "plan_over_matrix": round(api["plan_bytes"] / max(1.0, api["matrix_bytes"]), 3), "plan_build_ms": round(api["build_ms"], 1)}}
    A.close()
    coo2 = np.zeros(nnz, dtype=sm.COO_DTYPE)
    coo2["row"] = np.repeat(np.arange(rows, dtype=np.int32), np.diff(RP))
    coo2["col"], coo2["val"] = CI, V
    d_coo = torch.from_numpy(coo2.view(np.uint8)).cuda()
    del coo2, RP, CI, V
    T = sm.TjdsMatrix(sm.tjds_from_coo_device(d_coo, rows, cols, nnz), device=local_rank)
    del d_coo
    T.set_x(d_x, stream=stream)
    d_yt = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    T.spmv(d_yt, stream=stream)
    torch.cuda.synchronize()
    if not torch.equal(d_yt, d_y):
        raise SystemExit("pwt x%d: TJDS differs from CSR" % copies)
    tname, tbytes = T.describe()
    tpi = T.plan_info()
    prewarm(torch, lambda: T.spmv(d_yt, stream=stream))
    _, tms = timed_region(torch, dist, 1, steps, lambda: T.spmv(d_yt, stream=stream))
    tms /= steps
    out["tjds"] = {"kernel": tname, "ms_per_step": round(tms, 5), "alg_bytes_per_product": tbytes, "GFLOPs": round(2.0 * nnz / tms * 1e-6, 1),
                   "frac_of_hbm_peak": round(tbytes / tms * 1e-6 / HBM_PEAK_GBS, 4), "equals_csr_bit_for_bit": True,
                   "plan": {"plan_bytes": tpi["plan_bytes"], "matrix_bytes": tpi["matrix_bytes"],
                            "plan_over_matrix": round(tpi["plan_bytes"] / max(1.0, tpi["matrix_bytes"]), 3),
                            "plan_build_ms": round(tpi["build_ms"], 1), "value_cache": _value_cache(T, nnz)}}
    T.close()
    return out


def _value_cache(T, nnz):
    """The TJDS product's value cache: val lines shared by `min_tiles` tiles or more keep a tile-ordered second copy."""
    min_tiles, cached = T.get_value_cache()
    return {"min_tiles": min_tiles, "cached_share": round(cached / max(1, nnz), 4)}


def recorded_traffic(workload, kernel, alg_bytes):
    """HBM bytes per launch from the committed PMC passes (profiles/*traffic.json), if one matches this run.

    PMC counters cannot be read from inside the benchmark; tools/profile_bench.sh collects them in separate
    rocprofv3 --pmc passes over this same command and the summary is committed under profiles/.
    """
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        same_launch = abs(t.get("alg_bytes_per_launch", 0) - alg_bytes) <= 0.01 * alg_bytes
        if (t.get("workload") == workload and t.get("kernel") == kernel and same_launch
                and t.get("traffic_bytes_per_launch")):
            best = (t["traffic_bytes_per_launch"], os.path.basename(f))
    return best


def live_traffic(args, workload=None, fmt=None):
    """HBM-side bytes per PRODUCT of one workload's kernel, measured in THIS run: two child passes of this script under
    `rocprofv3 --kernel-trace --pmc` (FETCH_SIZE, then WRITE_SIZE -- they do not fit one pass), before this process
    touches the GPU.  FETCH_SIZE is doubled: gfx950 tallies 128-byte read requests at 64 B (MI355X_MICROARCH, "HBM").
    A product of several launches (the column sweep's generations) is the per-launch mean times its launches.
    Returns (bytes, description) or None when rocprofv3 is missing or a pass fails (the committed profile is used then)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    workload, fmt = workload or args.workload, fmt or args.format
    own = workload == args.workload and fmt == args.format     # the headline: its kernel flags apply
    inner = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "6", "--warmup", "2", "--workload", workload,
             "--format", fmt, "--kernel", args.kernel if own else "auto", "--kernel-param", str(args.kernel_param if own else 0),
             "--x", args.x, "--copies", str(args.copies), "--rows-log2", str(args.rows_log2), "--rows", str(args.rows)]
    env = dict(os.environ, TMPDIR="/tmp")
    vals, kernel, launches = {}, None, 1
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="smvp_pmc_", dir="/tmp")
        try:
            p = subprocess.run([rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--"] + inner,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not lines:
                return None
            roof = json.loads(lines[-1])["roofline"]
            kernel, launches = roof["kernel"], int(roof.get("launches_per_product", 1))
            # a product of several different kernels (the binned plan: "csr_binned: a + b + c", each launched once per
            # product) is the sum of their per-launch means; one kernel launched several times (the column sweep's
            # generations) its per-launch mean times its launches
            parts = [k.strip() for k in kernel.split(": ", 1)[-1].split(" + ")]
            got = {k: [] for k in parts}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != counter:
                        continue
                    for k in parts:
                        if k in row.get("Kernel_Name", ""):
                            got[k].append(float(row["Counter_Value"]))
                            break
            if not all(got.values()):
                return None
            vals[counter] = sum(sum(v) / len(v) for v in got.values()) * (launches if len(parts) == 1 else 1)
        except Exception:
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    traffic = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return traffic, ("measured in this run: rocprofv3 --kernel-trace --pmc child passes of bench.py (FETCH_SIZE %.0f KB x2 + "
                     "WRITE_SIZE %.0f KB per product = %d launch(es) of %s)" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"], launches, kernel))


def measure_c_layer(sm, rows, ngpus, steps, rank):
    """The C ABI's own sharded product -- smvp_sharded_spmv, what the command line's --gpus N and smvp_*_compute(ngpus > 1)
    run: ONE host process, one issuing thread and one RCCL rank per GPU -- on BASELINE config 4 over `ngpus` GPUs.
    Called on rank 0 only, after the torch.distributed legs (the other ranks are parked on a CPU barrier and have freed
    their matrices).  Per form: the longest GPU's event pair around the whole product, and host wall per product."""
    t0 = time.perf_counter()
    rp, ci, v = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, rows, rows, 32, 0, rows, threads=max(1, min(64, os.cpu_count() or 8)))
    host = np.add.reduceat(v, rp[:-1])
    scale = np.add.reduceat(np.abs(v), rp[:-1])
    nnz = int(rp[-1])
    out = {"workload": "uniform 32 entries/row rows=%d seed=2024 (BASELINE config 4)" % rows, "n_gpus": ngpus, "nnz": nnz, "steps": steps,
           "what": "smvp_sharded_spmv (C ABI, one process drives all GPUs; row blocks balanced by entries, each cut into row "
                   "chunks; RCCL all-gather of y per chunk)"}
    # fewer GPUs than ranks (a rehearsal on one GPU): the ranks share them and the y blocks travel by peer pushes.  Otherwise
    # SMVP_EXCHANGE_AUTO: RCCL's all-gather, peer copies and the push kernel each move one product's y when the handle is
    # created, the fastest is kept -- and this leg reports all of them, and the overlapped step under each
    virtual = ngpus > sm.device_count()
    for chunks in (1, 4):
        S = sm.ShardedMatrix("csr", ngpus, rows, rows, csr=(rp, ci, v), chunks=chunks,
                             exchange=sm.EXCHANGE_DIRECT if virtual else sm.EXCHANGE_AUTO)
        S.set_x(None)
        S.spmv(allgather=sm.GATHER_OVERLAPPED)
        S.synchronize()
        ys = [S.get_y(slot, gathered=True) for slot in sorted({0, ngpus - 1})]
        if not all(np.all(np.abs(y - host) <= TOL * scale) for y in ys) or not np.array_equal(ys[0], ys[-1]):
            raise SystemExit("C layer, %d chunk(s): the gathered y is wrong" % chunks)
        S.spmv(allgather=sm.GATHER_AFTER)
        S.synchronize()
        if not np.array_equal(S.get_y(0, gathered=True), ys[0]):
            raise SystemExit("C layer: GATHER_AFTER and GATHER_OVERLAPPED differ")
        info = S.probe_exchange(5)          # y_local now holds a real product's chunks
        chosen = info["active"]
        form = {"exchange_ms": {k: round(v_, 4) for k, v_ in info["ms"].items()}, "exchange_chosen": info["active_name"]}
        if chunks == 1:
            out["exchange"] = ("peer pushes between virtual ranks (rehearsal: %d ranks on %d GPU(s))" % (ngpus, sm.device_count())
                               if virtual else "AUTO -> %s (RCCL ncclAllGather / peer hipMemcpyAsync / push kernel, timed at creation)" % info["active_name"])
            out["exchange_chosen"], out["rccl_ranks"] = info["active_name"], info["rccl_ranks"]
            for k, v_ in info["ms"].items():
                out["exchange_%s_ms" % k] = round(v_, 4)
        by_exchange = {}
        for ex in info["available"]:
            S.set_exchange(ex)
            S.spmv(allgather=sm.GATHER_OVERLAPPED)
            S.synchronize()
            if not np.array_equal(S.get_y(ngpus - 1, gathered=True), ys[0]):
                raise SystemExit("C layer: exchange %s gives other bits" % sm.EXCHANGE_NAMES[ex])
            warm_until = time.perf_counter() + PREWARM_MS * 1e-3
            while time.perf_counter() < warm_until:
                S.spmv(allgather=sm.GATHER_OVERLAPPED)
                S.synchronize()
            ev = []
            for _ in range(steps):
                S.spmv(allgather=sm.GATHER_OVERLAPPED, timed=True)
                ev.append(S.synchronize())
            by_exchange[sm.EXCHANGE_NAMES[ex]] = round(float(np.mean(ev)), 4)
        form["overlapped_ms_by_exchange"] = by_exchange
        S.set_exchange(chosen)
        for label, mode in (("products_only", sm.GATHER_NONE), ("products_then_allgather", sm.GATHER_AFTER),
                            ("overlapped", sm.GATHER_OVERLAPPED)):
            warm_until = time.perf_counter() + PREWARM_MS * 1e-3
            while time.perf_counter() < warm_until:
                S.spmv(allgather=mode)
                S.synchronize()
            ev = []
            w0 = time.perf_counter()
            for _ in range(steps):
                S.spmv(allgather=mode, timed=True)
                ev.append(S.synchronize())
            wall = (time.perf_counter() - w0) / steps * 1e3
            form[label] = {"event_ms": round(float(np.mean(ev)), 4), "host_wall_ms": round(wall, 4),
                           "GFLOPs": round(2.0 * nnz / float(np.mean(ev)) * 1e-6, 1)}
        out["chunks_%d" % chunks] = form
        S.close()
    out["built_and_measured_in_s"] = round(time.perf_counter() - t0, 1)
    log(rank, "C layer on %d GPU(s): %s" % (ngpus, json.dumps({k: out[k] for k in out if k.startswith("chunks_")})))
    return out


def c_layer_in_child(args, ngpus, steps, rank):
    """N > 1: the C ABI's sharded product (one process driving every GPU) runs in a CHILD of rank 0, started before rank 0
    -- or any other rank: they wait on a file -- has touched a GPU, under a wall-clock budget: the layer has never run
    on more than one GPU, and a hang inside it (RCCL among the GPUs of one process) must cost this leg, not the run."""
    import signal
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--c-layer-child", str(ngpus), "--rows", str(args.rows), "--steps", str(steps)]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("TORCHELASTIC")}
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    except Exception as e:
        return {"error": "could not start the child: %s" % e}
    try:
        out, err = p.communicate(timeout=args.c_layer_budget)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)      # the process group this child was started as
        except Exception:
            pass
        try:
            p.communicate(timeout=10)
        except Exception:
            pass
        return {"error": "timeout", "budget_s": args.c_layer_budget, "n_gpus": ngpus}
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": "child exited with %s: %s" % (p.returncode, (err or out).strip().splitlines()[-1:] or "")}
    res = json.loads(lines[-1])
    res["ran_in"] = "a child process of rank 0, before any rank touched a GPU (%.1f s of a %.0f s budget)" % (time.perf_counter() - t0,
                                                                                                             args.c_layer_budget)
    log(rank, "C layer on %d GPUs (child process): %s" % (ngpus, json.dumps({k: res[k] for k in res if k.startswith("chunks_")})))
    return res


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no RANK / WORLD_SIZE in the environment -- the way
    the driver starts the scaling runs): this process becomes the launcher.  BEFORE importing torch or touching HIP in
    any way it starts the N ranks as child processes of its own -- this same script with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT set, each the leader of its own process group -- relays rank 0's stdout (the one
    JSON line), lets every rank's stderr through, waits under a wall-clock budget and returns the worst exit code.  Never
    os.exec*.  When a rank dies the others are given a short grace (they would wait for it in a collective for ever) and are
    then ended -- by the exact process groups started here.  The torch.distributed.run path stays as it was."""
    import signal
    import socket
    import subprocess

    n = args.gpus
    with socket.socket() as s:      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC")}
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SMVP_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    log(0, "--gpus %d without a launcher: starting %d rank processes (rendezvous 127.0.0.1:%d, budget %.0f s)" % (n, n, port, args.launch_budget))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0", ROLE_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True if r == 0 else None,
                                      start_new_session=True))

    def end(p):
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)     # the process group this rank was started as (its own children too)
            except Exception:
                try:
                    p.kill()
                except Exception:
                    pass

    import threading

    lines = []

    def relay():        # rank 0's stdout, line by line as it comes
        for l in procs[0].stdout:
            lines.append(l)
            sys.stdout.write(l)
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    deadline = time.time() + args.launch_budget
    grace = None
    why = ""
    while any(p.poll() is None for p in procs):
        now = time.time()
        failed = [i for i, p in enumerate(procs) if p.poll() not in (None, 0)]
        if failed and grace is None:
            grace = now + 30.0
            why = "rank %d exited with %s" % (failed[0], procs[failed[0]].returncode)
        if now > deadline or (grace is not None and now > grace):
            why = why or "the ranks did not finish within --launch-budget %.0f s" % args.launch_budget
            for p in procs:
                end(p)
            break
        time.sleep(0.1)
    for p in procs:
        try:
            p.wait(timeout=15)
        except Exception:
            end(p)
    t.join(timeout=10)
    codes = [p.returncode if p.returncode is not None else -9 for p in procs]
    rc = 0 if all(c == 0 for c in codes) else next((c for c in codes if c > 0), 1)
    if rc == 0 and not any(l.startswith("{") for l in lines):
        why, rc = "rank 0 printed no JSON line", 1
    if rc:
        log(0, "self-launched run failed (%s); exit codes by rank: %s" % (why or "non-zero exit", codes))
    return rc


def flat_keys(roof, others, extra, world, dist_info):
    """The figures a scaling record needs as FLAT SCALARS inside `roofline`: the driver's parse keeps the scalar keys of
    `roofline` / `cpu_baseline` / `config` (strings cut at 128 characters) and drops every nested object -- `roofline.others`,
    `plan`, `setup`, `extra` survive only as fragments of a truncated stdout tail (BENCH_r02 ... r04).  Same keys at every N;
    a leg that did not run leaves its keys out.  `others` stays on the line for a human reader."""
    def put(key, val, nd=4):
        if val is None:
            return
        roof[key] = round(float(val), nd) if isinstance(val, float) else val

    for key, name in (("tjds", "tjds"), ("survey_random_model", "survey_random_model"), ("config4", "config4"),
                      ("pwt_tiled_csr", "pwt_csr"), ("pwt_tiled_tjds", "pwt_tjds")):
        o = others.get(key)
        if not o or "frac" not in o:
            continue
        put("frac_" + name, o["frac"])
        put("ms_" + name, o.get("ms_per_product"), 5)
        put("traffic_over_alg_" + name, o.get("traffic_over_algorithmic"), 3)
        put("moved_frac_" + name, o.get("moved_frac_of_peak"))
    for key, name in (("tjds_two_phase", "frac_tjds_colmajor"), ("tjds_atomic", "frac_tjds_atomic")):
        if key in extra and "frac_of_hbm_peak" in extra[key]:
            put(name, extra[key]["frac_of_hbm_peak"])
    c4 = others.get("config4") or {}
    for k in ("t1_ms", "tN_step_ms", "tN_step_after_ms", "tN_products_only_ms", "speedup_overlapped", "speedup_after",
              "speedup_products_only", "chunks_chosen"):
        put("config4_" + k, c4.get(k))
    e8 = c4.get("eighth_of_n8") or {}
    for c, ms in ((e8.get("inputs") or {}).get("product_ms_by_chunks") or {}).items():
        put("config4_eighth_ms_%schunk" % c, ms)
    put("config4_eighth_chunks_chosen", e8.get("chosen"))
    cl = others.get("config4_c_layer") or {}
    if "error" in cl:
        put("config4_c_layer_error", str(cl["error"])[:120])
    for ch in (1, 4):
        f = cl.get("chunks_%d" % ch) or {}
        for form, short in (("products_only", "products_only"), ("products_then_allgather", "after"), ("overlapped", "overlapped")):
            put("config4_c_layer_%s_ms_%dchunk" % (short, ch), (f.get(form) or {}).get("event_ms"))
        for name, ms in (f.get("overlapped_ms_by_exchange") or {}).items():
            put("config4_c_layer_overlapped_ms_%dchunk_%s" % (ch, name), ms)
    for k in ("exchange_rccl_ms", "exchange_copies_ms", "exchange_direct_ms", "exchange_chosen", "rccl_ranks"):
        put("c_layer_" + k if k in ("exchange_chosen", "rccl_ranks") else k, cl.get(k))
    # t1 over the C layer's best overlapped step (any chunk count, any exchange form it ran): the >= 3.5x figure for the one-process,
    # N-GPU driver, beside config4_speedup_overlapped (one process per GPU, RCCL)
    best = [v for k, v in roof.items() if k.startswith("config4_c_layer_overlapped_ms_") and isinstance(v, float) and v > 0]
    if best and c4.get("t1_ms"):
        put("config4_c_layer_step_best_ms", min(best))
        put("config4_c_layer_speedup_best", float(c4["t1_ms"]) / min(best), 3)
    hp = others.get("headline_products_only") or {}
    put("headline_products_only_ms", hp.get("ms_per_product"), 5)
    for name, e in (others.get("sample_matrices_us_per_product") or {}).items():
        if not isinstance(e, dict):
            continue
        stem = name.replace(".mtx", "")
        for k, short in (("csr_avg_ms", "csr_us"), ("tjds_avg_ms", "tjds_us"), ("csr_loop_wall_ms_per_product", "csr_loop_wall_us"),
                         ("tjds_loop_wall_ms_per_product", "tjds_loop_wall_us")):
            put("%s_%s" % (stem, short), e.get(k), 3)
    c5 = extra.get("config5_pwt") or {}
    for k, short in (("csr_ms_per_step", "config5_csr_us"), ("tjds_ms_per_step", "config5_tjds_us"), ("csr_then_tjds_ms_per_step", "config5_both_us")):
        if k in c5:
            put(short, c5[k] * 1e3, 3)
    put("exchange", dist_info.get("exchange"))
    put("dist_backend", dist_info.get("backend"))
    put("rccl_ranks", dist_info.get("rccl_ranks"))
    put("n_gpus", world)
    put("self_launched", dist_info.get("self_launched"))
    put("prewarm_ms", PREWARM_MS)      # untimed device work in front of every leg's timed region (the W warm-up steps come on top)


def roofline_of(res, workload=None):
    achieved = res["alg_bytes_local"] / (res["kernel_ms"] * 1e-3) * 1e-9
    r = {"bound": "hbm", "kernel": res["kernel"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
         "alg_bytes_per_launch": res["alg_bytes_local"] / res.get("launches", 1),
         "ms_per_launch": round(res["kernel_ms"] / res.get("launches", 1), 5), "launches_per_product": res.get("launches", 1),
         "ms_per_product": round(res["kernel_ms"], 5), "plan": res.get("plan"),
         "note": "achieved / frac = SURVEY 8(d)'s ALGORITHMIC bytes over the measured time (moved_* = the bytes the counters saw); "
                 "HIP events on the launch stream over the timed products; one launch per product except the column "
                 "sweep's generations and the binned plan's three kernels" +
                 ("; this kernel reads the plan's 16-bit column offsets (2 B per entry) where the algorithmic count has "
                  "col_ind's 4 B, so the measured traffic can lie below the algorithmic bytes" if ", 5, " in res["kernel"] else "")}
    rec = recorded_traffic(workload, res["kernel"], res["alg_bytes_local"]) if workload else None
    if rec:
        r["traffic"] = rec[0]
        r["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)" % rec[1]
    return r


def leg_tjds(torch, dist, sm, args, blk, res, local_rank, rank, extra):
    """The TJDS product of the headline matrix beside the CSR headline (N = 1): built on the GPU from the block's entries, checked
    against the CSR result, the one-kernel form timed, then the two-phase and the atomic form -> extra.tjds, extra.tjds_two_phase,
    extra.tjds_atomic.  Informational: an exception is reported in extra.tjds, the headline line is never lost over it."""
    try:
        t0 = time.perf_counter()
        coo = np.zeros(blk["nnz"], dtype=sm.COO_DTYPE)
        coo["row"] = np.repeat(np.arange(blk["rows"], dtype=np.int32), np.diff(blk["row_ptr"]))
        coo["col"], coo["val"] = blk["col_ind"], blk["val"]
        d_coo = torch.from_numpy(coo.view(np.uint8)).cuda()
        del coo
        tj = sm.tjds_from_coo_device(d_coo, blk["rows"], blk["cols_total"], blk["nnz"])   # radix sort + scans on the GPU
        del d_coo
        torch.cuda.synchronize()
        t_conv = time.perf_counter() - t0
        T = sm.TjdsMatrix(tj, device=local_rank)
        tname, tbytes = T.describe()
        tpi = T.plan_info()
        log(rank, "TJDS built in %.1f s: %d jagged diagonals" % (time.perf_counter() - t0, tj.num_diag))
        stream = torch.cuda.current_stream()
        d_yt = torch.empty(blk["rows"], dtype=torch.float64, device="cuda")
        T.set_x(res["d_x"], stream=stream)

        def tjds_step():
            T.zero_y(d_yt, stream=stream)       # a no-op unless the atomic form is selected
            T.spmv(d_yt, stream=stream)

        tjds_step()
        torch.cuda.synchronize()
        terr = float((np.abs(d_yt.cpu().numpy() - res["got"]) / np.maximum(res["scale"], 1e-300)).max())
        if terr > TOL:
            raise RuntimeError("TJDS differs from CSR: %g" % terr)
        tsteps = max(20, args.steps // 2)      # (a handful of sub-millisecond products right behind an idle device read up to 6 % fast)
        prewarm(torch, tjds_step)
        _, t_ms = timed_region(torch, dist, 1, tsteps, tjds_step)
        t_ms /= tsteps
        tj_workload = blk["name"] + ", TJDS, x=%s" % args.x
        trec = recorded_traffic(tj_workload, tname, tbytes)
        extra["tjds"] = {"kernel": tname, "ms_per_step": round(t_ms, 4), "num_diag": tj.num_diag, "alg_bytes_per_product": tbytes,
                         "GFLOPs": round(2.0 * blk["nnz"] / (t_ms * 1e-3) * 1e-9, 1),
                         "achieved_GBps": round(tbytes / (t_ms * 1e-3) * 1e-9, 1),
                         "frac_of_hbm_peak": round(tbytes / (t_ms * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4),
                         "max_normwise_diff_vs_csr": terr, "steps": tsteps,
                         "plan": {"plan_bytes": tpi["plan_bytes"], "matrix_bytes": tpi["matrix_bytes"],
                                  "plan_over_matrix": round(tpi["plan_bytes"] / max(1.0, tpi["matrix_bytes"]), 3),
                                  "plan_build_ms": round(tpi["build_ms"], 1), "value_cache": _value_cache(T, blk["nnz"])},
                         "convert_device_ms": round(t_conv * 1e3, 1),
                         "traffic_bytes_per_product": trec[0] if trec else None,
                         "traffic_source": ("profiles/" + trec[1]) if trec else None,
                         "note": "ONE kernel per product: the entries regrouped by row at create time (val / row_ind / "
                                 "start_pos / perm untouched), every 2048-entry tile walks its piece of the jagged "
                                 "diagonals in TJDS order, products meet in LDS, one lane (or wave) per row sums them; "
                                 "no atomics, bit-reproducible.  extra.tjds_two_phase / tjds_atomic are the older forms"}
        for key, mode in (("tjds_two_phase", sm.TJDS_MODE_TWO_PHASE), ("tjds_atomic", sm.TJDS_MODE_ATOMIC)):
            T.set_mode(mode)
            tjds_step()
            _, a_ms = timed_region(torch, dist, 1, max(3, tsteps // 2), tjds_step)
            a_ms /= max(3, tsteps // 2)
            extra[key] = {"kernel": T.describe()[0], "ms_per_step": round(a_ms, 4),
                          "GFLOPs": round(2.0 * blk["nnz"] / (a_ms * 1e-3) * 1e-9, 1),
                          "frac_of_hbm_peak": round(tbytes / (a_ms * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4)}
        extra["tjds_two_phase"]["note"] = "column-major products kernel + per-row sums through the row-inverted index"
        extra["tjds_atomic"]["note"] = "memset(y) + column-major scatter with fp64 atomics"
        T.close()
        del T, tj, d_yt
    except Exception as e:  # the TJDS leg is informational; never lose the headline line over it
        extra["tjds"] = {"error": str(e)}


def leg_cpu_baseline(args, blk, res, extra):
    """`cpu_baseline`: the reference's serial loop (the oracle's restatement of main-cli.c:410-416, gcc -O3 -DNDEBUG, one thread, y
    reset outside the window) on the whole headline matrix, sized for about 15 s; beside it, for context only, the same loop on
    every core of this host (extra.cpu_all_cores_context).  Returns the cpu_baseline object."""
    cpu = None
    import oracle_binding as ob          # the checker, used here only as the CPU baseline leg

    rp, ci, v, xh = blk["row_ptr"], blk["col_ind"], blk["val"], res["x_host"]
    _, probe = ob.csr_timed(rp, ci, v, xh, 1)
    iters = args.cpu_iters or int(max(2, min(100, round(15000.0 / max(probe[0], 1e-3)))))
    y_cpu, ms = ob.csr_timed(rp, ci, v, xh, iters)
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    cpu = {"value": round(2.0 * blk["nnz"] / (ms.mean() * 1e-3) * 1e-9, 3), "unit": "GFLOP/s", "cores": 1,
           "kind": "port", "host_cores_total": os.cpu_count(), "host_cpu": model,
           "GBps": round(res["alg_bytes_local"] / (ms.mean() * 1e-3) * 1e-9, 2),
           "ms_per_product": round(float(ms.mean()), 2),
           "sample": "the full workload matrix, %d products of the serial loop (oracle restatement of "
                     "main-cli.c:410-416, gcc -O3 -DNDEBUG, y reset outside the window)" % iters,
           "agrees_with_gpu": bool(np.all(np.abs(y_cpu - res["got"]) <= TOL * res["scale"])),
           # SURVEY 8(c) asks for these two beside the row-normwise bound: element-wise relative error (large only on
           # rows whose sum cancels to ~1e-15 of its terms, whatever the order) and the infinity-norm error of y
           "max_elementwise_rel_error": float((np.abs(y_cpu - res["got"]) / np.maximum(np.abs(y_cpu), 1e-300))[y_cpu != 0].max())
           if np.any(y_cpu != 0) else 0.0,
           "inf_norm_rel_error": float(np.abs(y_cpu - res["got"]).max() / max(float(np.abs(y_cpu).max()), 1e-300)),
           "gpu_rows_bit_identical_to_serial": round(float((y_cpu == res["got"]).mean()), 4)}
    # context only, NOT the reference (which is one thread): the same serial loop on every core of this host, each
    # thread on its own run of rows (ctypes releases the GIL inside the C loop)
    try:
        from concurrent.futures import ThreadPoolExecutor

        T = min(os.cpu_count() or 1, 64)
        cuts = np.searchsorted(rp, np.linspace(0, rp[-1], T + 1)).clip(0, blk["rows"])
        cuts[0], cuts[-1] = 0, blk["rows"]
        parts = [(rp[a:b + 1] - rp[a], ci[rp[a]:rp[b]], v[rp[a]:rp[b]]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        with ThreadPoolExecutor(len(parts)) as pool:
            list(pool.map(lambda p: ob.csr_spmv(p[0], p[1], p[2], xh), parts))          # warm
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                ys = list(pool.map(lambda p: ob.csr_spmv(p[0], p[1], p[2], xh), parts))
                dt = (time.perf_counter() - t0) * 1e3
                best = dt if best is None else min(best, dt)
        extra["cpu_all_cores_context"] = {
            "threads": len(parts), "ms_per_product": round(best, 3), "GFLOPs": round(2.0 * blk["nnz"] / best * 1e-6, 2),
            "agrees_with_serial": bool(np.array_equal(np.concatenate(ys), y_cpu)),
            "note": "not the reference (it is serial): the same C loop on row blocks of equal entry count, one thread each; "
                    "includes Python's dispatch of the threads"}
    except Exception as e:
        extra["cpu_all_cores_context"] = {"error": str(e)}
    return cpu


def leg_sample_matrices(sm, args, local_rank):
    """BASELINE configs 1-3 and 5 through the reference-shaped entry points (smvp_csr_compute / smvp_tjds_compute, -n 1000): the
    in-kernel window per product, the loop wall per product, hipEvent pairs, and the serial CPU loops on this host beside them
    -> extra.sample_matrices.  Cache-resident and launch-bound (1.9 / 2.9 MB of traffic): no HBM roofline is claimed for these."""
    samples = {}
    for name in ("ibm32.mtx", "memplus.mtx", "pwt.mtx"):      # BASELINE configs 1 (on the GPU: there is no CPU path), 2/3, 5
        try:
            tc, m, n, coo = sm.mm_read_coo(golden_file("sample-data", name))
            y_c, ms_c, st_c = sm.csr_compute(coo, m, n, iters=1000, device=local_rank)
            info_c = sm.last_run_info()
            y_t, ms_t, st_t = sm.tjds_compute(coo, m, n, iters=1000, device=local_rank)
            info_t = sm.last_run_info()
            _, _, ev_c = sm.csr_compute(coo, m, n, iters=1000, device=local_rank, timing=sm.TIMING_EVENTS)
            _, _, ev_t = sm.tjds_compute(coo, m, n, iters=1000, device=local_rank, timing=sm.TIMING_EVENTS)
            e = {"rows": m, "nnz": len(coo), "iters": 1000,
                 "timing": "per product on the device: every wave stamps the constant-rate wall clock when it starts "
                           "and when its last store is acknowledged, time = max(last) - min(first); the 1000 "
                           "products run %s" % ("up to 1024 per launch of the repeating kernel (barrier between products)"
                                                 if info_c.repeat_launches else "one launch each, replayed from a hipGraph")
                           if info_c.timing == sm.TIMING_DEVICE else "hipEvent pairs",
                 "repeat_launches": info_c.repeat_launches, "graph_replays": info_c.graph_replays,
                 "csr_avg_ms": round(st_c.time_avg, 6), "csr_min_ms": round(st_c.time_min, 6),
                 "csr_GFLOPs": round(2.0 * len(coo) / st_c.time_avg * 1e-6, 2),
                 "csr_loop_wall_ms_per_product": round(info_c.wall_ms / 1000.0, 6),
                 "tjds_avg_ms": round(st_t.time_avg, 6), "tjds_min_ms": round(st_t.time_min, 6),
                 "tjds_GFLOPs": round(2.0 * len(coo) / st_t.time_avg * 1e-6, 2),
                 "tjds_loop_wall_ms_per_product": round(info_t.wall_ms / 1000.0, 6),
                 "csr_avg_ms_event_pairs": round(ev_c.time_avg, 6), "tjds_avg_ms_event_pairs": round(ev_t.time_avg, 6)}
            if not args.no_cpu_baseline:
                import oracle_binding as ob      # CPU baseline leg: the serial loops on this host, 1 thread

                rp, ci, v = ob.csr_build(coo, m)
                y_cpu, ms_cpu = ob.csr_timed(rp, ci, v, np.ones(n), 1000)
                yt_cpu, mst_cpu = ob.tjds_timed(ob.tjds_build(coo, m, n), np.ones(n), 1000)
                sc = ob.csr_spmv(rp, ci, np.abs(v), np.ones(n))
                e.update(cpu_csr_avg_ms=round(float(ms_cpu.mean()), 6), cpu_tjds_avg_ms=round(float(mst_cpu.mean()), 6),
                         csr_agrees_with_cpu=bool(np.all(np.abs(y_c - y_cpu) <= TOL * sc)),
                         tjds_agrees_with_cpu=bool(np.all(np.abs(y_t - yt_cpu) <= TOL * sc)),
                         csr_rows_bit_identical=round(float((y_c == y_cpu).mean()), 4))
            # the only numbers the reference publishes: average times in its committed reports (BASELINE.md,
            # hardware not stated) -- output-test/smvp-toolbox_report_{CSR,TJDS}_*.txt
            published = {"ibm32.mtx": (0.0004319, 0.0007779), "memplus.mtx": (0.387638, 0.549908),
                         "pwt.mtx": (0.569281, 1.1823)}[name]
            e["reference_report_csr_avg_ms"], e["reference_report_tjds_avg_ms"] = published
            # (no GPU-over-reference ratio is printed: the reference's window is a host clock around its product on
            # unknown hardware; the comparable figure here is csr_loop_wall_ms_per_product, beside it above)
            samples[name] = e
        except Exception as ex:
            samples[name] = {"error": str(ex)}
    return samples


def leg_setup_conversion(torch, sm, blk):
    """COO -> CSR of the headline matrix (main-cli.c:340-365): smvp_csr_from_coo_device on the GPU against smvp_csr_from_coo on the
    host (one thread, a 2^20-row sample scaled up) -> roofline.setup."""
    try:
        coo = np.zeros(blk["nnz"], dtype=sm.COO_DTYPE)
        coo["row"] = np.repeat(np.arange(blk["rows"], dtype=np.int32), np.diff(blk["row_ptr"]))
        coo["col"], coo["val"] = blk["col_ind"], blk["val"]
        d_coo = torch.from_numpy(coo.view(np.uint8)).cuda()
        torch.cuda.synchronize()
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            rp_d, ci_d, v_d = sm.csr_from_coo_device(d_coo, blk["rows"], blk["cols_total"], blk["nnz"])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        same = bool(torch.equal(rp_d.cpu(), torch.from_numpy(blk["row_ptr"])) and torch.equal(ci_d.cpu(), torch.from_numpy(blk["col_ind"])))
        del d_coo, rp_d, ci_d, v_d
        k = min(blk["rows"], 1 << 20)          # the first 2^20 rows on the host
        nk = int(blk["row_ptr"][k])
        t0 = time.perf_counter()
        sm.csr_from_coo(coo[:nk], k)
        host_ms = (time.perf_counter() - t0) * 1e3
        del coo
        return {
            "convert_device_ms": round(best, 1), "device_arrays_equal_input": same,
            "convert_host_ms_sample": round(host_ms, 1), "host_sample": "%d rows, %d entries, one thread" % (k, nk),
            "convert_host_ms_scaled_to_full": round(host_ms * blk["nnz"] / max(nk, 1), 1),
            "note": "COO -> CSR (main-cli.c:340-365) of the headline matrix: smvp_csr_from_coo_device (radix sort + scan on the "
                    "GPU, COO already in HBM) against smvp_csr_from_coo on the host; plan = the launch plan the product keeps "
                    "beside the format's arrays (roofline.plan)"}
    except Exception as e:
        return {"error": str(e)}


def leg_random_model(torch, dist, sm, sharding, args, local_rank, rank, extra):
    """The SURVEY 8(d) random model itself -- the workload the >= 60 % target is written on -- through whatever AUTO picks (the binned
    plan), checked, timed, its plan priced, its bit-reproducibility asserted, the tile kernel timed beside it
    -> extra.survey_random_model."""
    world = 1
    try:
        blk2 = build_block(sm, sharding, "memplus_shaped", args, rank, world)
        r2 = measure_csr(torch, dist, sm, sharding, blk2, args, world, local_rank, rank, max(20, args.steps // 2),
                         min(args.warmup, WARM_SHORT), False)
        rl = roofline_of(r2)
        far = float((np.abs(np.repeat(np.arange(blk2["rows"]), np.diff(blk2["row_ptr"])) - blk2["col_ind"]) > 4096).mean())
        extra["survey_random_model"] = {
            "workload": blk2["name"], "nnz": blk2["nnz"], "kernel": rl["kernel"], "ms_per_launch": rl["ms_per_launch"],
            "ms_per_product": rl["ms_per_product"], "launches_per_product": rl["launches_per_product"],
            "GFLOPs": round(2.0 * blk2["nnz"] / (r2["kernel_ms"] * 1e-3) * 1e-9, 1),
            "achieved_GBps": rl["achieved"], "frac_of_hbm_peak": rl["frac"],
            "share_of_entries_beyond_4096": round(far, 3), "alg_bytes_per_product": r2["alg_bytes_local"],
            "gather_spread_estimate": round(r2["A"].gather_spread(), 3)}
        # what AUTO picked, what its plan costs, that it repeats itself bit for bit, and the tile kernel beside it
        A2, rm = r2["A"], extra["survey_random_model"]
        auto_kernel = A2.get_kernel()
        rm["auto_picks"] = {1: "vector", 2: "tile (stream)", 3: "tile (stream-carry)", 4: "column sweep",
                            5: "binned (near band %d)" % auto_kernel[1]}.get(auto_kernel[0])
        pi = A2.plan_info()
        rm["plan"] = {"plan_bytes": pi["plan_bytes"], "matrix_bytes": pi["matrix_bytes"],
                      "plan_over_matrix": round(pi["plan_bytes"] / pi["matrix_bytes"], 3), "plan_build_ms": round(pi["build_ms"], 1)}
        st2 = torch.cuda.current_stream()
        A2.spmv(r2["d_x"], r2["d_y"], stream=st2)
        torch.cuda.synchronize()
        y_first = r2["d_y"].clone()
        A2.spmv(r2["d_x"], r2["d_y"], stream=st2)
        torch.cuda.synchronize()
        rm["bit_identical_run_to_run"] = bool(torch.equal(y_first, r2["d_y"]))
        if not rm["bit_identical_run_to_run"]:
            raise SystemExit("the random model's product is not the same from run to run")
        del y_first
        if auto_kernel[0] != sm.CSR_KERNEL_STREAM:
            A2.set_kernel(sm.CSR_KERNEL_STREAM, 0)
            prewarm(torch, lambda: A2.spmv(r2["d_x"], r2["d_y"], stream=st2))
            tsteps = max(5, args.steps // 8)
            _, t_ms = timed_region(torch, dist, 1, tsteps, lambda: A2.spmv(r2["d_x"], r2["d_y"], stream=st2))
            t_ms /= tsteps
            ok2, worst2, _ = host_check(blk2, r2["x_host"], r2["d_y"].cpu().numpy())
            if not ok2:
                raise SystemExit("the tile kernel is wrong on the random model (%g)" % worst2)
            rm["tile_kernel"] = A2.describe()[0]
            rm["tile_kernel_ms"] = round(t_ms, 5)
            rm["tile_kernel_frac"] = round(r2["alg_bytes_local"] / t_ms * 1e-6 / HBM_PEAK_GBS, 4)
        r2["A"].close()
    except Exception as e:
        extra["survey_random_model"] = {"error": str(e)}


def build_others(extra, blk, res, world, c_layer, live_others):
    """roofline.others: every other kernel the line reports, priced like the headline (algorithmic bytes of SURVEY 8(d) per product /
    measured time; traffic from this run's own --pmc child passes where they ran).  `extra` repeats these with more detail; the
    driver's parse drops nested objects, so flat_keys repeats the figures that matter as scalars of `roofline`."""
    def other(kernel, ms, alg, nnz, key=None, **more):
        o = {"kernel": kernel, "ms_per_product": round(ms, 5), "alg_bytes_per_product": alg,
             "achieved": round(alg / ms * 1e-6, 1), "unit": "GB/s", "frac": round(alg / ms * 1e-6 / HBM_PEAK_GBS, 4),
             "GFLOPs": round(2.0 * nnz / ms * 1e-6, 1), "traffic": None}
        lt = live_others.get(key) if key else None
        if lt:
            o["traffic"], o["traffic_over_algorithmic"], o["traffic_source"] = lt[0], round(lt[0] / alg, 3), lt[1]
            o["moved_GBps"], o["moved_frac_of_peak"] = round(lt[0] / ms * 1e-6, 1), round(lt[0] / ms * 1e-6 / HBM_PEAK_GBS, 4)
        o.update(more)
        return o

    others = {}
    t = extra.get("tjds")
    if t and "error" not in t:
        others["tjds"] = other(t["kernel"], t["ms_per_step"], t["alg_bytes_per_product"], blk["nnz"], "tjds", workload=blk["name"] + ", TJDS",
                               plan=t.get("plan"), convert_device_ms=t.get("convert_device_ms"))
        if others["tjds"]["traffic"] is None and t.get("traffic_bytes_per_product"):
            others["tjds"]["traffic"], others["tjds"]["traffic_source"] = t["traffic_bytes_per_product"], t["traffic_source"]
    c4 = extra.get("config4")
    if c4 and "error" not in c4:
        if world == 1:
            others["config4"] = other(c4["kernel"], c4["spmv_only_ms"], c4["alg_bytes_per_product"], c4["nnz"], "config4",
                                      workload=c4["workload"], launches_per_product=c4["launches_per_product"],
                                      auto_picks=c4["auto_picks"], bit_identical_run_to_run=True,
                                      tile_kernel_ms=c4["tile_kernel_spmv_only_ms"], tile_kernel_frac=c4["tile_kernel_frac_of_hbm_peak"])
        else:
            others["config4"] = {k: c4[k] for k in ("workload", "n_gpus", "kernel", "chunks_per_rank", "spmv_only_ms", "spmv_only_GFLOPs",
                                                    "step_ms_products_then_allgather", "step_ms_overlapped",
                                                    "step_GFLOPs_products_then_allgather", "step_GFLOPs_overlapped",
                                                    "tile_kernel_spmv_only_ms", "exchange") if k in c4}
            others["config4"]["note"] = ("the matrix BASELINE.md writes the >= 3.5x at 8 GPUs target on; t1_ms is the whole matrix on "
                                         "one GPU of this node, measured in this run")
        # the same keys at every N (N = 1: the step is the product, the speed-ups are 1)
        for k in ("t1_ms", "tN_step_ms", "tN_step_after_ms", "tN_products_only_ms", "speedup_overlapped", "speedup_after",
                  "speedup_products_only", "chunks_chosen", "chunk_choice", "plan", "eighth_of_n8", "chunks_chosen_for_n8"):
            if k in c4:
                others["config4"][k] = c4[k]
    if c_layer:
        others["config4_c_layer"] = c_layer
    if world > 1:   # the headline step's own product time (no exchange), so that the curve can be read both ways
        others["headline_products_only"] = {"ms_per_product": round(res["kernel_ms"], 5),
                                            "GFLOPs": round(2.0 * res["nnz_total"] / (res["kernel_ms"] * 1e-3) * 1e-9, 1),
                                            "y_bytes_gathered_per_step": blk["rows_total"] * 8,
                                            "note": "7 entries per row: 8 B of y per row over xGMI against 105 B per row from HBM, "
                                                    "so the headline step is exchange-bound at N > 1 by construction"}
    pt = extra.get("pwt_tiled")
    if pt and "error" not in pt:
        others["pwt_tiled_csr"] = other(pt["kernel"], pt["ms_per_launch"], pt["alg_bytes_per_product"], pt["nnz"], workload=pt["workload"],
                                        plan=pt.get("plan"))
        tj = pt.get("tjds")
        if tj:
            others["pwt_tiled_tjds"] = other(tj["kernel"], tj["ms_per_step"], tj["alg_bytes_per_product"], pt["nnz"],
                                             workload=pt["workload"] + ", TJDS", plan=tj.get("plan"))
    rm = extra.get("survey_random_model")
    if rm and "error" not in rm:
        others["survey_random_model"] = other(rm["kernel"], rm["ms_per_product"], rm["alg_bytes_per_product"], rm["nnz"],
                                              "survey_random_model", workload=rm["workload"],
                                              launches_per_product=rm["launches_per_product"], auto_picks=rm.get("auto_picks"),
                                              tile_kernel_ms=rm.get("tile_kernel_ms"), tile_kernel_frac=rm.get("tile_kernel_frac"),
                                              bit_identical_run_to_run=rm.get("bit_identical_run_to_run"),
                                              plan=rm.get("plan"),
                                              note="the model SURVEY 8(d) writes the >= 60 % target on: 39 % of its entries point "
                                                   "anywhere in a 134 MB x.  AUTO picks the binned plan for it (near part on the tile "
                                                   "kernel; far products through LDS-resident blocks of x into bins, then per-row sums); "
                                                   "the tile kernel alone runs it at the L2-miss gather rate (tile_kernel_*).  The "
                                                   "headline is this model's exact-structure substitute")
    sm_ = extra.get("sample_matrices")
    if sm_:
        others["sample_matrices_us_per_product"] = {
            name: {k: round(e[k] * 1e3, 3) for k in ("csr_avg_ms", "tjds_avg_ms", "csr_avg_ms_event_pairs", "tjds_avg_ms_event_pairs",
                                                      "csr_loop_wall_ms_per_product", "tjds_loop_wall_ms_per_product",
                                                      "cpu_csr_avg_ms", "cpu_tjds_avg_ms") if k in e}
            for name, e in sm_.items() if "error" not in e}
        others["sample_matrices_us_per_product"]["note"] = (
            "BASELINE configs 1-3, 5 at -n 1000, microseconds: *_avg_ms = in-kernel wall-clock stamps (what the report "
            "file prints by default), *_event_pairs = hipEvent pair around each launch, *_loop_wall = host wall of the "
            "whole 1000-product loop / 1000, cpu_* = the reference's serial loop on this host; cache-resident, no HBM claim")
    return others


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and not args.c_layer_child and not args.pmc_child:
        # no launcher around this process: it starts its own ranks (before torch or HIP are touched) and relays rank 0's line
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if world != args.gpus:
        args.gpus = world

    if args.c_layer_child:      # child of rank 0 (N > 1): nothing but the C layer, its result as one JSON line
        import smvp_toolkit_amd as sm
        print(json.dumps(measure_c_layer(sm, args.rows, args.c_layer_child, max(5, args.steps), 0)), flush=True)
        return
    # N > 1: the C-layer leg first, in a child of rank 0, while no rank holds a GPU context; the others wait on a file
    c_layer = None
    if world > 1 and not args.no_c_layer and not args.no_config4 and not args.pmc_child:
        import tempfile
        flag = os.path.join(tempfile.gettempdir(), "smvp_bench_c_layer_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
        if rank == 0:
            c_layer = c_layer_in_child(args, world, max(5, args.steps // 10), rank)
            open(flag, "w").write("done\n")
        else:
            deadline = time.time() + args.c_layer_budget + 60.0
            while not os.path.exists(flag) and time.time() < deadline:
                time.sleep(0.2)
    if args.pmc_child:      # inner run of a counter pass: the headline product only
        args.no_tjds = args.no_random_model = args.no_samples = args.no_cpu_baseline = True
        args.no_config4 = args.no_pwt_tiled = args.no_live_traffic = True
    live, live_others = None, {}
    if rank == 0 and world == 1 and not args.no_live_traffic and "RANK" not in os.environ:
        log(rank, "roofline.traffic: two rocprofv3 --pmc child passes of the headline product ...")
        live = live_traffic(args)
        log(rank, "roofline.traffic: %s" % (live[1] if live else "child passes unavailable, using the committed profile"))
        # the same for the other kernels the line carries in roofline.others (each its own pair of child passes)
        wanted = []
        if args.workload == "memplus_tiled" and args.format == "csr":
            if not args.no_tjds:
                wanted.append(("tjds", "memplus_tiled", "tjds"))
            if not args.no_random_model:
                wanted.append(("survey_random_model", "memplus_shaped", "csr"))
            if not args.no_config4:
                wanted.append(("config4", "uniform32", "csr"))
        for key, wl, fmt in wanted:
            live_others[key] = live_traffic(args, wl, fmt)
            log(rank, "roofline.others.%s.traffic: %s" % (key, live_others[key][1] if live_others[key] else "child passes unavailable"))

    import torch
    import torch.distributed as dist

    import smvp_toolkit_amd as sm
    from smvp_toolkit_amd import sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # one rank per GPU; SMVP_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the sharded path
    backend = os.environ.get("SMVP_DIST_BACKEND", "nccl")
    local_rank %= max(1, torch.cuda.device_count()) if backend != "nccl" else max(1, local_rank + 1)
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("SMVP_FORCE_DIST") == "1" and "RANK" in os.environ   # rehearse RCCL with one rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev_name, cus, hbm = sm.device_info(local_rank)

    # ------------------------------------------------------------ headline: CSR on the workload
    blk = build_block(sm, sharding, args.workload, args, rank, world)
    res = measure_csr(torch, dist, sm, sharding, blk, args, world, local_rank, rank, args.steps, args.warmup,
                      collective=not args.no_allgather)
    gflops = 2.0 * res["nnz_total"] / res["wall_per_step"] * 1e-9
    extra = {"device": dev_name, "compute_units": cus, "nnz": int(res["nnz_total"]), "rows": blk["rows_total"],
             "alg_bytes_per_step": res["alg_bytes_total"], "x": args.x,
             "whole_job_GBps": round(res["alg_bytes_total"] / res["wall_per_step"] * 1e-9, 1),
             "max_normwise_error_vs_host": res["worst"]}
    if res["golden"]:
        extra["full_size_parity"] = res["golden"]
    if world > 1:
        extra["spmv_only_ms"] = round(res["kernel_ms"], 5)
        extra["spmv_only_GFLOPs"] = round(2.0 * res["nnz_total"] / (res["kernel_ms"] * 1e-3) * 1e-9, 1)
        extra["allgather_in_step"] = not args.no_allgather
        extra["y_bytes_gathered"] = blk["rows_total"] * 8

    # ------------------------------------------------------------ TJDS beside it (same matrix)
    if not args.no_tjds and world == 1 and args.format == "csr":
        leg_tjds(torch, dist, sm, args, blk, res, local_rank, rank, extra)

    # ------------------------------------------------------------ CPU baseline (rank 0, N = 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = leg_cpu_baseline(args, blk, res, extra)

    # ------------------------------------------------------------ the reference's own sample matrices, -n 1000
    # BASELINE configs 2/3 (memplus.mtx CSR / TJDS) and 5 (pwt.mtx CSR + TJDS back to back) through the
    # reference-shaped entry points: per-iteration hipEvent windows, y cleared outside them.  Cache-resident
    # and launch-bound (1.9 / 2.9 MB of traffic): no HBM roofline is claimed for these.
    if rank == 0 and world == 1 and not args.no_samples:
        extra["sample_matrices"] = leg_sample_matrices(sm, args, local_rank)

    headline_roofline = roofline_of(res, blk["name"] + ", %s, x=%s" % (args.format.upper(), args.x))
    # set-up beside the product (the reference's user waits for main-cli.c:340-365 / :766-926, not for the timed loop):
    # COO -> CSR of the headline matrix on the GPU, and the host converter (one thread) on a slice of it
    if rank == 0 and world == 1 and args.format == "csr" and not args.pmc_child:
        headline_roofline["setup"] = leg_setup_conversion(torch, sm, blk)
    if live:
        # (a product of several different kernels -- the binned plan -- is priced whole; the column sweep's generations per launch)
        per = 1 if " + " in res["kernel"] else res.get("launches", 1)
        headline_roofline["traffic"], headline_roofline["traffic_source"] = live[0] / per, live[1]
    if headline_roofline.get("traffic"):
        # `achieved` / `frac` price SURVEY 8(d)'s ALGORITHMIC bytes; this is the same time against the bytes the counters saw move
        # (the 16-bit column offsets move fewer than the algorithmic 12 B per entry, the binned and TJDS plans more)
        per_ms = headline_roofline["ms_per_product"] if " + " in res["kernel"] else headline_roofline["ms_per_launch"]
        headline_roofline["moved_GBps"] = round(headline_roofline["traffic"] / per_ms * 1e-6, 1)
        headline_roofline["moved_frac_of_peak"] = round(headline_roofline["traffic"] / per_ms * 1e-6 / HBM_PEAK_GBS, 4)
    res["A"].close()
    del res["keep"], res["d_x"], res["d_y"]
    torch.cuda.empty_cache()

    # ------------------------------------------------------------ BASELINE config 4 at this N; pwt x459 at N = 1
    if not args.no_config4:
        try:
            extra["config4"] = measure_config4(torch, dist, sm, sharding, args, world, local_rank, rank, max(5, args.steps // 4))
        except SystemExit:
            raise
        except Exception as e:
            extra["config4"] = {"error": str(e)}
        torch.cuda.empty_cache()
    if not args.no_samples:
        try:
            extra["config5_pwt"] = measure_config5(torch, dist, sm, sharding, world, local_rank, rank, 200)
        except SystemExit:
            raise
        except Exception as e:
            extra["config5_pwt"] = {"error": str(e)}
    if world == 1 and not args.no_pwt_tiled:
        try:
            extra["pwt_tiled"] = measure_pwt_tiled(torch, dist, sm, sharding, local_rank, rank, max(20, args.steps // 2))
        except SystemExit:
            raise
        except Exception as e:
            extra["pwt_tiled"] = {"error": str(e)}
        torch.cuda.empty_cache()

    # ------------------------------------------------------------ the survey's random model, for the record
    if args.workload == "memplus_tiled" and not args.no_random_model and world == 1:
        leg_random_model(torch, dist, sm, sharding, args, local_rank, rank, extra)

    # ------------------------------------------------------------ the C ABI's own sharded product (N = 1: here; N > 1: it ran
    # first, in a child process of rank 0 -- see above)
    if world == 1 and not args.no_c_layer and not args.no_config4 and not args.pmc_child:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        try:
            c_layer = measure_c_layer(sm, args.rows, world, max(5, args.steps // 10), rank)
        except BaseException as e:   # a wrong result included: this leg reports its failure instead of ending the run
            c_layer = {"error": str(e) or type(e).__name__}
    if rank == 0 and world > 1:
        try:
            os.remove(flag)
        except Exception:
            pass

    headline_roofline["others"] = others = build_others(extra, blk, res, world, c_layer, live_others)
    # what the communicator itself reports (not WORLD_SIZE): every rank adds a one and the sum is what took part
    dist_info = {"backend": "none (one process, one GPU)", "rccl_ranks": 0, "exchange": "none (one GPU)",
                 "self_launched": os.environ.get("SMVP_BENCH_SELF_LAUNCHED") == "1"}
    if dist.is_initialized():
        one = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(one)
        took_part = int(round(float(one[0])))
        dist_info["backend"] = "nccl (RCCL)" if backend == "nccl" else backend
        dist_info["rccl_ranks"] = took_part if backend == "nccl" else 0
        dist_info["ranks_in_group"] = took_part
        dist_info["exchange"] = ("all_gather_into_tensor of the y blocks over %s, %d ranks; config 4: block-cyclic chunks, gather "
                                 "behind each chunk's product" % ("RCCL/xGMI" if backend == "nccl" else backend, took_part))
    flat_keys(headline_roofline, others, extra, world, dist_info)
    extra["dist"] = dist_info

    if rank == 0:
        # what this process leaves behind: its children (the rocprofv3 --pmc passes, the C-layer child) have been waited for;
        # BENCH_r02 / r03 counted one process at the end of the run -- not one of these (none is left)
        try:
            import psutil
            kids = psutil.Process().children(recursive=True)
            extra["child_processes_at_exit"] = [" ".join(k.cmdline())[:120] for k in kids if k.is_running()]
        except Exception:
            extra["child_processes_at_exit"] = None
        line = {
            "metric": "fp64 %s SpMV GFLOP/s (2*nnz flop per product; achieved HBM GB/s in roofline)" % args.format.upper() +
                      ("; step = local product + RCCL all-gather of y on the headline matrix (strong scaling); the curve BASELINE.md "
                       "writes the >= 3.5x at 8 GPUs target on (config 4) is roofline.others.config4, the products alone "
                       "roofline.others.headline_products_only" if world > 1 else ""),
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(res["wall_per_step"] * 1e3, 5), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": blk["name"] + ", %s, x=%s" % (args.format.upper(), args.x) +
                       (", %d row blocks + %s all-gather of y" % (world, "RCCL" if backend == "nccl" else backend)
                        if world > 1 and not args.no_allgather else ""),
                       "format": args.format, "kernel": res["kernel"], "nnz": int(res["nnz_total"]),
                       "rows": blk["rows_total"], "sharding": "row-block x%d" % world},
            "roofline": headline_roofline, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
