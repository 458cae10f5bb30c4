#!/usr/bin/env python3
"""bench.py -- the headline benchmark: fp64 SpMV on MI355X, CSR (and TJDS beside it).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one product y = A x over the whole (sharded) matrix, inputs resident in HBM, CSR, x = ones like the
reference (main-cli.c:368-369).  memplus.mtx itself is 1.9 MB and lives in L2, so the headline workload is memplus.mtx
replicated 944x along the diagonal (kron(I_944, memplus): 16.76 M rows, 119 M entries, 1.77 GB of algorithmic traffic) --
the exact-structure substitute for SURVEY 8(d)'s random "memplus-shaped" model, which is measured beside it
(roofline.frac_survey_random_model), as are BASELINE config 4 (10 M x 32/row, roofline.config4_*: the 1 -> N GPU curve is
written on it), config 5 (pwt.mtx), pwt.mtx x459 and the sample matrices at -n 1000.  DESIGN.md "Workloads" has the reasoning.

With N > 1 the headline matrix is cut into N row blocks, one process per GPU, and a step is the local product plus the RCCL
all-gather of the y blocks over xGMI (strong scaling).  Started as plain `python bench.py --gpus N` (no RANK / WORLD_SIZE in
the environment) the script is its own launcher (bench_core.spawn_ranks, before torch or HIP are touched).

This file is the contract: the flags, the order of the legs, their wall-clock budgets, and the ONE compact JSON line rank 0
prints LAST on stdout (< 8000 characters at every N: the contract keys, `roofline` and `cpu_baseline` as flat scalars).
Everything else -- roofline.others, extra, plans, notes -- goes to bench_detail.json beside this script and to stderr.
bench_core.py holds the timed region and the headline measurement, bench_legs.py the secondary legs.  A leg that fails, or
that the budget no longer allows, leaves a `<leg>_error` key on the line; a leg that hangs costs that leg: at --hard-deadline a
watchdog prints the line from what has been measured and ends the process.
"""
import argparse
import json
import os
import sys
import threading
import time

import bench_core as core
from bench_core import HBM_PEAK_GBS, ROOT, log

LINE_LIMIT = 7600       # characters; the driver parsed 18.7 KB and lost 21.9 KB -- stay far below either
STRING_LIMIT = 110      # characters of any one string value on the compact line


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="memplus_tiled", choices=["memplus_tiled", "pwt_tiled", "memplus_shaped", "uniform32"])
    ap.add_argument("--copies", type=int, default=0, help="memplus_tiled / pwt_tiled: diagonal blocks (0 = 944 / 459 -> 16.76 M rows)")
    ap.add_argument("--rows-log2", type=int, default=24, help="memplus_shaped: total rows = 2^k")
    ap.add_argument("--rows", type=int, default=10_000_000, help="uniform32 / config 4: total rows")
    ap.add_argument("--format", default="csr", choices=["csr", "tjds"], help="storage format of the timed product")
    ap.add_argument("--kernel", default="auto", choices=["auto", "stream", "vector", "stream-carry", "colsweep", "binned"])
    ap.add_argument("--kernel-param", type=int, default=0)
    ap.add_argument("--x", default="ones", choices=["ones", "random"])
    for leg in ("tjds", "random-model", "samples", "config4", "pwt-tiled", "eighth", "c-layer", "cpu-baseline", "live-traffic", "allgather"):
        ap.add_argument("--no-" + leg, action="store_true")
    ap.add_argument("--chunks", type=int, default=0, help="config 4, N > 1: row chunks per rank (0 = chosen from this run's measurements)")
    ap.add_argument("--config4-kernel", default="auto", choices=["auto", "colsweep", "tile"])
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)          # the inner run of a --pmc child pass
    ap.add_argument("--c-layer-child", type=int, default=0, help=argparse.SUPPRESS)      # child process: the C layer alone on this many GPUs
    ap.add_argument("--c-layer-budget", type=float, default=200.0, help="N > 1: seconds the C-layer child of rank 0 may take")
    ap.add_argument("--budget", type=float, default=400.0,
                    help="soft wall-clock budget (s): a secondary leg is not started when less than its estimate is left")
    ap.add_argument("--hard-deadline", type=float, default=540.0,
                    help="seconds after which the watchdog prints the line from what has been measured and ends the process")
    ap.add_argument("--launch-budget", type=float, default=0.0, help="--gpus N without a launcher: seconds for the ranks (0 = hard deadline + 60)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the full record goes")
    ap.add_argument("--cpu-iters", type=int, default=0, help="0 = sized for about 15 s")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the same matrix cut into N row blocks (default); weak = N times the matrix")
    a = ap.parse_args()
    a.launch_budget = a.launch_budget or a.hard_deadline + 60.0
    return a


def scalars(d, keep=()):
    """The scalar entries of a dict, strings cut at STRING_LIMIT, floats at 6 significant digits where they are long."""
    out = {}
    for k, v in (d or {}).items():
        if isinstance(v, (dict, list, tuple)) and k not in keep:
            continue
        if isinstance(v, str) and len(v) > STRING_LIMIT:
            v = v[:STRING_LIMIT - 3] + "..."
        if isinstance(v, float) and (v != v or v in (float("inf"), float("-inf"))):
            v = None        # (NaN / Infinity are not JSON: a strict parser would lose the whole line over one of them)
        if isinstance(v, float) and v == v and abs(v) not in (0.0, float("inf")) and len(repr(v)) > 12:
            v = float("%.6g" % v)
        out[k] = v
    return out


# what leaves the line first when it is too long (prefixes of roofline keys, least important first)
DROP_ORDER = ("leg_seconds", "ibm32_", "pwt_csr_loop", "pwt_tjds_loop", "memplus_csr_loop", "memplus_tjds_loop", "config5_", "pwt_", "memplus_",
              "config4_c_layer_overlapped_ms_", "config4_c_layer_after", "config4_c_layer_products", "moved_frac_", "ms_", "traffic_over_alg_",
              "config4_eighth_", "exchange_", "note", "traffic_source")
PROTECTED = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "ms_per_launch", "ms_per_product", "launches_per_product",
             "alg_bytes_per_launch")


def compact_line(state):
    """The one stdout line: contract keys, `roofline` / `cpu_baseline` / `config` as flat scalars, nothing nested below them."""
    line = scalars(state["head"])
    line["config"] = scalars(state["config"])
    roof = scalars(state["roofline"])
    for leg, err in state["errors"].items():
        roof[leg + "_error"] = str(err)[:STRING_LIMIT]
    if state["leg_seconds"]:
        roof["leg_seconds"] = ", ".join("%s %.0f" % kv for kv in state["leg_seconds"].items())[:300]
    roof["wall_s"] = round(time.time() - state["t0"], 1)
    if state.get("watchdog"):
        roof["watchdog"] = state["watchdog"]
    line["roofline"] = roof
    line["cpu_baseline"] = scalars(state["cpu"]) if state["cpu"] else None
    line["detail"] = os.path.basename(state["detail_path"]) if state.get("detail_written") else None
    text = json.dumps(line, allow_nan=False)
    for prefix in DROP_ORDER:
        if len(text) <= LINE_LIMIT:
            break
        for k in [k for k in roof if k.startswith(prefix) and k not in PROTECTED]:
            del roof[k]
        roof["dropped_for_length"] = True
        text = json.dumps(line, allow_nan=False)
    return text


DYING = threading.Event()      # set by the watchdog: it is ending this process with an exit code of its own


class Emitter:
    """Prints the compact line exactly once -- from main() when the legs are through, or from the watchdog at the hard deadline."""

    def __init__(self, state, rank, hard):
        self.state, self.rank, self.lock, self.done = state, rank, threading.Lock(), False
        t = threading.Thread(target=self._watch, args=(hard,), daemon=True)
        t.start()

    def emit(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0 and self.state.get("head"):
                write_detail(self.state)
                self.state["out"].write(compact_line(self.state) + "\n")
                self.state["out"].flush()

    def _watch(self, hard):
        while time.time() - self.state["t0"] < hard:
            time.sleep(0.25)
            if self.done:
                return
        DYING.set()     # (from here on an exception in the main thread -- its peers are leaving too -- must not decide the exit code)
        self.state["watchdog"] = "hard deadline %.0f s passed in leg '%s'" % (hard, self.state.get("leg", "?"))
        self.state["errors"].setdefault(self.state.get("leg", "run"), "did not finish before the hard deadline")
        log(self.rank, self.state["watchdog"] + ": printing the line from what has been measured")
        had_head = bool(self.state.get("head"))
        self.emit()
        sys.stderr.flush()
        os._exit(0 if had_head or self.rank else 3)     # every rank leaves at the same deadline


def write_detail(state):
    """bench_detail.json: the compact line's figures with everything the line leaves out (others, extra, plans, notes)."""
    detail = {"head": state.get("head"), "config": state.get("config"), "roofline": state.get("roofline"), "cpu_baseline": state.get("cpu"),
              "others": state.get("others"), "extra": state.get("extra"), "errors": state.get("errors"), "leg_seconds": state.get("leg_seconds")}
    try:
        with open(state["detail_path"], "w") as f:
            json.dump(detail, f, indent=1, default=str)
        state["detail_written"] = True
    except Exception as e:
        log(0, "could not write %s: %s" % (state["detail_path"], e))
    for k in ("others", "extra"):
        for name, v in (state.get(k) or {}).items():
            log(0, "detail.%s.%s: %s" % (k, name, json.dumps(v, default=str)[:1500]))


def main():
    args = parse()
    rank, world, local_rank = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    child = bool(args.c_layer_child or args.pmc_child)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and not child:
        sys.exit(core.spawn_ranks(args, sys.argv[1:]))     # no launcher around this process: it starts its own ranks
    args.gpus = world
    import bench_legs as legs

    if args.c_layer_child:      # child of rank 0 (N > 1): nothing but the C layer, its result as one JSON line
        import smvp_toolkit_amd as sm
        print(json.dumps(legs.measure_c_layer(sm, args.rows, args.c_layer_child, max(5, args.steps), 0)), flush=True)
        return
    # stdout carries the ONE line and nothing else: whatever the libraries print there (RCCL's version banner ...) goes to stderr
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    state = {"t0": time.time(), "errors": {}, "leg_seconds": {}, "extra": {}, "others": {}, "cpu": None, "roofline": {}, "config": {},
             "detail_path": args.detail, "leg": "start", "out": line_out}
    emitter = None if child else Emitter(state, rank, args.hard_deadline)
    left = lambda: args.budget - (time.time() - state["t0"])
    if args.pmc_child:          # inner run of a counter pass: the headline product only
        for k in ("tjds", "random_model", "samples", "cpu_baseline", "config4", "pwt_tiled", "live_traffic", "c_layer"):
            setattr(args, "no_" + k, True)

    # ---- roofline.traffic measured in this run: rocprofv3 --pmc child passes, before this process touches the GPU (N = 1)
    live, live_others = None, {}
    if rank == 0 and world == 1 and not args.no_live_traffic and "RANK" not in os.environ:
        state["leg"], t_leg = "live_traffic", time.time()
        live = legs.live_traffic(args, budget=min(60.0, left()))
        log(rank, "roofline.traffic: %s" % (live[1] if live else "child passes unavailable, using the committed profile"))
        if args.workload == "memplus_tiled" and args.format == "csr":
            for key, wl, fmt, skip in (("tjds", "memplus_tiled", "tjds", args.no_tjds), ("survey_random_model", "memplus_shaped", "csr", args.no_random_model),
                                       ("config4", "uniform32", "csr", args.no_config4)):
                if not skip and time.time() - t_leg < 150.0:        # all passes together: 150 s at most
                    live_others[key] = legs.live_traffic(args, wl, fmt, budget=min(60.0, 150.0 - (time.time() - t_leg)))
        state["leg_seconds"]["live_traffic"] = time.time() - t_leg

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
    if world > 1 or (os.environ.get("SMVP_FORCE_DIST") == "1" and "RANK" in os.environ):     # (the latter: RCCL with one rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))
    dev_name, cus, _ = sm.device_info(local_rank)

    # ---- the headline: CSR (or --format tjds) on the workload; K timed steps between barriers, max over ranks
    state["leg"], t_leg = "headline", time.time()
    blk = core.build_block(sm, sharding, args.workload, args, rank, world)
    res = core.measure_csr(torch, dist, sm, sharding, blk, args, world, local_rank, rank, args.steps, args.warmup, collective=not args.no_allgather)
    short = ("memplus.mtx x%d block-diagonal (substitute for the SURVEY 8(d) random model; that model: roofline.frac_survey_random_model)"
             % (blk["rows_total"] // blk["base"][0]) if args.workload == "memplus_tiled" else blk["name"])
    roof = core.roofline_of(res, short + ", %s, x=%s" % (args.format.upper(), args.x))
    if live:      # (the binned plan's product of several kernels is priced whole; the column sweep's generations per launch)
        per = 1 if " + " in res["kernel"] else res.get("launches", 1)
        roof["traffic"], roof["traffic_source"] = live[0] / per, live[1]
    if roof.get("traffic"):
        per_ms = roof["ms_per_product"] if " + " in res["kernel"] else roof["ms_per_launch"]
        roof["moved_GBps"] = round(roof["traffic"] / per_ms * 1e-6, 1)
        roof["moved_frac_of_peak"] = round(roof["traffic"] / per_ms * 1e-6 / HBM_PEAK_GBS, 4)
    extra = state["extra"]
    extra.update(device=dev_name, compute_units=cus, nnz=int(res["nnz_total"]), rows=blk["rows_total"], alg_bytes_per_step=res["alg_bytes_total"],
                 x=args.x, whole_job_GBps=round(res["alg_bytes_total"] / res["wall_per_step"] * 1e-9, 1), max_normwise_error_vs_host=res["worst"])
    if res["golden"]:
        extra["full_size_parity"] = res["golden"]
        roof["y_equals_tiled_reference_y"] = True
    state["config"] = {"workload": short + ", %s, x=%s" % (args.format.upper(), args.x), "format": args.format, "kernel": res["kernel"],
                       "nnz": int(res["nnz_total"]), "rows": blk["rows_total"], "sharding": "row-block x%d" % world,
                       "exchange": ("%s all-gather of y" % ("RCCL" if backend == "nccl" else backend)) if world > 1 and not args.no_allgather else "none"}
    state["roofline"] = roof
    state["head"] = {"metric": "fp64 %s SpMV GFLOP/s (2*nnz flop per product; achieved HBM GB/s in roofline)" % args.format.upper(),
                     "value": round(2.0 * res["nnz_total"] / res["wall_per_step"] * 1e-9, 2), "unit": "GFLOP/s", "n_gpus": world,
                     "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(res["wall_per_step"] * 1e3, 5),
                     "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    state["leg_seconds"]["headline"] = time.time() - t_leg

    def leg(name, estimate, fn, when=True):
        """One secondary leg under the soft budget; every rank takes rank 0's decision.  A wrong result (SystemExit) ends the run."""
        if not when:
            return
        go = torch.tensor([1.0 if left() >= estimate else 0.0], dtype=torch.float64, device="cuda")
        if dist.is_initialized():
            dist.broadcast(go, 0)
        if float(go[0]) == 0.0:
            state["errors"][name] = "skipped: %.0f s of the --budget %.0f s left, the leg is estimated at %.0f s" % (left(), args.budget, estimate)
            return
        state["leg"], t0 = name, time.time()
        try:
            fn()
        except SystemExit:
            raise
        except Exception as e:
            state["errors"][name] = "%s: %s" % (type(e).__name__, e)
        state["leg_seconds"][name] = time.time() - t0
        torch.cuda.empty_cache()

    n1, lead = world == 1, rank == 0 and world == 1
    leg("tjds", 15, lambda: legs.leg_tjds(torch, dist, sm, args, blk, res, local_rank, rank, extra), n1 and not args.no_tjds and args.format == "csr")
    leg("cpu_baseline", 30, lambda: state.update(cpu=legs.leg_cpu_baseline(args, blk, res, extra)), lead and not args.no_cpu_baseline)
    leg("sample_matrices", 15, lambda: extra.update(sample_matrices=legs.leg_sample_matrices(sm, args, local_rank)), lead and not args.no_samples)
    leg("setup", 10, lambda: roof.update(setup=legs.leg_setup_conversion(torch, sm, blk)), lead and args.format == "csr" and not args.pmc_child)
    res["A"].close()
    del res["keep"], res["d_x"], res["d_y"]
    torch.cuda.empty_cache()
    leg("config4", 45 if n1 else 90, lambda: extra.update(config4=legs.measure_config4(torch, dist, sm, sharding, args, world, local_rank, rank,
                                                                                      max(5, args.steps // 4), left)), not args.no_config4)
    leg("config5_pwt", 10, lambda: extra.update(config5_pwt=legs.measure_config5(torch, dist, sm, sharding, world, local_rank, rank, 200)), not args.no_samples)
    leg("pwt_tiled", 20, lambda: extra.update(pwt_tiled=legs.measure_pwt_tiled(torch, dist, sm, sharding, local_rank, rank, max(20, args.steps // 2))),
        n1 and not args.no_pwt_tiled)
    leg("survey_random_model", 30, lambda: legs.leg_random_model(torch, dist, sm, sharding, args, local_rank, rank, extra),
        n1 and args.workload == "memplus_tiled" and not args.no_random_model)

    # what the communicator itself reports (not WORLD_SIZE): every rank adds a one and the sum is what took part
    dist_info = {"backend": "none (one process, one GPU)", "rccl_ranks": 0, "exchange": "none (one GPU)",
                 "self_launched": os.environ.get("SMVP_BENCH_SELF_LAUNCHED") == "1"}
    if dist.is_initialized():
        one = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(one)
        took_part = int(round(float(one[0])))
        dist_info.update(backend="nccl (RCCL)" if backend == "nccl" else backend, rccl_ranks=took_part if backend == "nccl" else 0, ranks_in_group=took_part,
                         exchange="all_gather_into_tensor of y over %s, %d ranks" % ("RCCL/xGMI" if backend == "nccl" else backend, took_part))
        dist.destroy_process_group()
    extra["dist"] = dist_info
    if rank != 0:           # the other ranks are done: rank 0 alone runs the C-layer child (one process driving every GPU) and prints
        emitter.done = True
        return

    # ---- the C ABI's own sharded product (smvp_sharded_spmv): N = 1 in this process; N > 1 in a child of rank 0, now that the
    # other ranks have left their GPUs, under its own wall-clock budget
    c_layer = None
    if not args.no_c_layer and not args.no_config4 and not args.pmc_child:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        state["leg"], t_leg = "config4_c_layer", time.time()
        if left() < 30:
            c_layer = {"error": "skipped: %.0f s of the --budget left" % left()}
        elif world == 1:
            try:
                c_layer = legs.measure_c_layer(sm, args.rows, 1, max(5, args.steps // 10), rank)
            except BaseException as e:       # a wrong result included: this leg reports its failure instead of ending the run
                c_layer = {"error": str(e) or type(e).__name__}
        else:
            args.c_layer_budget = max(20.0, min(args.c_layer_budget, args.hard_deadline - (time.time() - state["t0"]) - 25.0))
            c_layer = legs.c_layer_in_child(args, world, max(5, args.steps // 10), rank)
        state["leg_seconds"]["config4_c_layer"] = time.time() - t_leg
    state["leg"] = "line"
    state["others"] = others = legs.build_others(extra, blk, res, world, c_layer, live_others)
    legs.flat_keys(roof, others, extra, world, dist_info)
    try:
        import psutil
        extra["child_processes_at_exit"] = [" ".join(k.cmdline())[:120] for k in psutil.Process().children(recursive=True) if k.is_running()]
    except Exception:
        extra["child_processes_at_exit"] = None
    if emitter:
        emitter.emit()
    else:       # a --pmc child pass: the parent reads roofline.kernel / launches_per_product off this line
        state["detail_written"] = False
        line_out.write(compact_line(state) + "\n")
        line_out.flush()


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        if DYING.is_set():      # the watchdog has fired (the peers are going away under this thread's collectives): its exit code counts
            time.sleep(30.0)
        raise
