"""bench_legs.py -- the secondary legs of bench.py (TJDS, BASELINE configs 4 and 5, pwt x459, the SURVEY 8(d) random model, the
sample matrices at -n 1000, the C ABI's own sharded product, the live --pmc passes, the CPU baseline) and what turns their
results into the flat scalar keys of the compact line and into bench_detail.json.  bench.py decides which legs run and in
which order; nothing here prints on stdout."""
import json
import os
import sys
import time

import numpy as np

from bench_core import (HBM_PEAK_GBS, ROOT, TOL, build_block, golden_file, host_check, log, measure_csr, prewarm, recorded_traffic,
                        report_y_lines, roofline_of, timed_region, _value_cache, PREWARM_MS, WARM_LONG, WARM_SHORT)
import bench_core


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


def measure_config4(torch, dist, sm, sharding_mod, args, world, local_rank, rank, steps, left=None):
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

    sweep_budget = min(60.0, max(10.0, (left() if left else 180.0) / 3.0))
    # ---- how many chunks per rank
    if args.chunks > 0 or not gather:
        chunks = max(1, args.chunks) if gather else 1
        choice = {"chosen": chunks, "rule": "--chunks %d" % args.chunks if gather else "one GPU, no exchange: one chunk"}
    else:
        product_ms, gather_ms = {}, {}
        sweep_t0 = time.perf_counter()
        for c in (1, 2, 4):
            # the sweep has its own wall-clock share (a third of what the soft budget leaves, 60 s at most; rank 0's clock decides for
            # every rank): a slow sweep ends early and the choice is made from the chunk counts it got to
            if c > 1:
                over = torch.tensor([1.0 if time.perf_counter() - sweep_t0 > sweep_budget else 0.0], dtype=torch.float64, device="cuda")
                dist.broadcast(over, 0)
                if float(over[0]):
                    log(rank, "config 4: the chunk sweep ran over its %.0f s; choosing from %s" % (sweep_budget, sorted(product_ms)))
                    break
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



def live_traffic(args, workload=None, fmt=None, budget=300.0):
    """HBM-side bytes per PRODUCT of one workload's kernel, measured in THIS run: two child passes of this script under
    `rocprofv3 --kernel-trace --pmc` (FETCH_SIZE, then WRITE_SIZE -- they do not fit one pass), before this process
    touches the GPU.  FETCH_SIZE is doubled: gfx950 tallies 128-byte read requests at 64 B (MI355X_MICROARCH, "HBM").
    A product of several launches (the column sweep's generations) is the per-launch mean times its launches.
    Returns (bytes, description) or None when rocprofv3 is missing or a pass fails or runs over its half of `budget` seconds (the
    committed profile is used then)."""
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
    inner = [sys.executable, bench_core.BENCH_SCRIPT, "--pmc-child", "--steps", "6", "--warmup", "2", "--workload", workload,
             "--format", fmt, "--kernel", args.kernel if own else "auto", "--kernel-param", str(args.kernel_param if own else 0),
             "--x", args.x, "--copies", str(args.copies), "--rows-log2", str(args.rows_log2), "--rows", str(args.rows)]
    env = dict(os.environ, TMPDIR="/tmp")
    vals, kernel, launches = {}, None, 1
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="smvp_pmc_", dir="/tmp")
        try:
            p = subprocess.run([rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--"] + inner,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=max(15.0, budget / 2.0))
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

    cmd = [sys.executable, bench_core.BENCH_SCRIPT, "--c-layer-child", str(ngpus), "--rows", str(args.rows), "--steps", str(steps)]
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



L2_GATHER_CEILING_G = 246.0     # G random 8-byte gathers per second that HIT the L2 (a 1 MB table, 16 in flight per lane; 190 with a 4 MB table that only just fits): profiles/r06_gather_ceiling.txt


def flat_keys(roof, others, extra, world, dist_info):
    """The figures a record needs as FLAT SCALARS inside `roofline` -- the compact line carries nothing nested (bench.py
    compact_line), so the other kernels' fractions, traffic over algorithmic bytes, config 4's t1 / tN / speed-up keys, the C
    layer's legs and exchange times, the sample matrices' microseconds and what the communicator reports are put here by name.
    Same keys at every N; a leg that did not run leaves its keys out.  `others` and `extra` go to bench_detail.json."""
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
    # config 4 is bound by random 8-byte gathers that hit the L2, not by HBM bytes: the rate beside the chip's ceiling for such gathers
    # (tools/gather_ceiling.hip -- profiles/r06_gather_ceiling.txt)
    xg = (extra.get("config4") or {}).get("x_gathers_per_second_G_per_gpu")
    if xg:
        put("config4_G_gathers_per_s_per_gpu", xg, 1)
        put("config4_frac_of_l2_gather_ceiling", float(xg) / L2_GATHER_CEILING_G)
    # N > 1: what ONE chunk's all-gather took by chunks per rank, and the local products by chunks per rank (the sweep's measurements)
    inputs = (c4.get("chunk_choice") or {}).get("inputs") or {}
    for c, ms in (inputs.get("gather_ms_by_chunks") or {}).items():
        put("config4_allgather_ms_%schunk" % c, ms)
    for c, ms in (inputs.get("product_ms_by_chunks") or {}).items():
        put("config4_products_ms_%schunk" % c, ms)
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
    for name, e in extra.items():            # a leg that reported its own failure: <leg>_error on the line
        if isinstance(e, dict) and "error" in e:
            put(name + "_error", str(e["error"])[:100])
    put("exchange", dist_info.get("exchange"))
    put("dist_backend", dist_info.get("backend"))
    put("rccl_ranks", dist_info.get("rccl_ranks"))
    put("n_gpus", world)
    put("self_launched", dist_info.get("self_launched"))
    put("prewarm_ms", PREWARM_MS)      # untimed device work in front of every leg's timed region (the W warm-up steps come on top)


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
