#!/usr/bin/env python3
"""One-GPU run of the C layer's sharded product (smvp_sharded_spmv, what `--gpus N` uses) for a timeline:

    cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -o t -- python3 tools/sharded_timeline.py
    python3 tools/sharded_timeline.py --summarize OUT > profiles/r03_sharded_overlap_timeline.txt

A config-4-shaped matrix (uniform 32 entries per row) as ONE row block cut into 4 chunks, 1-rank RCCL communicator.
Per product: 4 chunk products on the compute stream; SMVP_GATHER_OVERLAPPED puts the all-gather of chunk c on the
communication stream right behind the event of chunk c's product.  With one rank RCCL's all-gather is a device-to-device
copy of the chunk (a __amd_rocclr_copyBuffer kernel in the kernel trace, or an entry of the memory-copy trace).
"""
import argparse, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smvp-toolkit_amd", "python"))


def run(rows, chunks):
    import smvp_toolkit_amd as sm
    rp, ci, v = sm.synth_csr(sm.SYNTH_UNIFORM, 2024, rows, rows, 32, 0, rows, threads=16)
    S = sm.ShardedMatrix("csr", 1, rows, rows, csr=(rp, ci, v), chunks=chunks)
    S.set_x(None)
    for mode in (sm.GATHER_OVERLAPPED, sm.GATHER_AFTER):
        for _ in range(4):
            S.spmv(allgather=mode)
            ms = S.synchronize()
        print("mode %d: last product %.4f ms" % (mode, ms), flush=True)
    S.close()


def summarize(out):
    ev = []
    for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))))
    for f in glob.glob(os.path.join(out, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "?"), r.get("Stream_Id", "?")))
    ev.sort()
    prod = [i for i, e in enumerate(ev) if e[2] == "kernel" and ("csr_colsweep" in e[3] or "csr_stream_owner" in e[3])]
    place = [i for i, e in enumerate(ev) if e[2] == "kernel" and "place_gathered" in e[3]]
    print("# %d events; %d product launches, %d place_gathered launches" % (len(ev), len(prod), len(place)))
    # the products of the run: 4 overlapped, then 4 gather-after; show the last of each mode
    per_mode = len(place) // 2
    for label, pi in (("SMVP_GATHER_OVERLAPPED", place[per_mode - 1]), ("SMVP_GATHER_AFTER", place[-1])):
        lo = place[place.index(pi) - 1] + 1 if place.index(pi) > 0 else 0
        t0 = ev[lo][0]
        print("\n== %s: one product (times in us from its first launch)" % label)
        print("%10s %10s  %-6s %-8s %s" % ("start", "end", "kind", "stream", "what"))
        last_prod_end, first_copy_start = None, None
        for e in ev[lo:pi + 1]:
            print("%10.1f %10.1f  %-6s %-8s %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, e[2], e[4], e[3]))
            if e[2] == "kernel" and ("csr_colsweep" in e[3] or "csr_stream_owner" in e[3]):
                last_prod_end = e[1]
            if (e[2] == "copy" or "copyBuffer" in e[3]) and first_copy_start is None:   # the 1-rank all-gather
                first_copy_start = e[0]
        if first_copy_start is not None and last_prod_end is not None:
            print("-> first gather starts %.1f us %s the last chunk product ends" % (
                abs(last_prod_end - first_copy_start) / 1e3, "BEFORE" if first_copy_start < last_prod_end else "after"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=5_000_000)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--summarize", default=None)
    a = ap.parse_args()
    if a.summarize:
        summarize(a.summarize)
    else:
        run(a.rows, a.chunks)
