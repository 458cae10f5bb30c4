#!/bin/bash
# [PAT=substring] [BENCH_ARGS=..] tools/profile_bench.sh TAG  -- run on the GPU box (through gpurun): per-kernel times and HBM traffic of bench.py's
# headline workload.  Writes gpurun_out/profile_TAG/{kernel_stats.csv, pmc_summary.txt, traffic.json, bench.json}.
#   1. rocprofv3 --kernel-trace --stats on `bench.py --no-random-model --no-tjds --no-cpu-baseline` (bench.json = that process's line)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC, SQ ...) on the same command, fewer steps
# FETCH_SIZE is doubled (gfx950 tallies 128-B reads at 64 B, MI355X_MICROARCH.md "HBM"); WRITE_SIZE is exact.
set -u
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p "$OUT"
ARGS="--no-live-traffic --no-c-layer --no-random-model --no-tjds --no-cpu-baseline --no-samples --no-config4 --no-pwt-tiled ${BENCH_ARGS:-}"
PAT=${PAT:-csr_stream_owner<8, 5}   # kernel-name substring the PMC summary is taken over; several, separated by '|', for a product of
                                    # several kernels (the binned plan): their per-launch means are added
# ONE process gives both the kernel trace and the bench line (bench.json = the stdout of the traced run): the profile's average
# duration and the line's ms_per_launch can be reconciled without a run-to-run or box-to-box caveat
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 $R/bench.py --steps 100 --warmup 10 --detail "$OUT/bench_detail.json" $ARGS > "$OUT/bench.json" 2> "$OUT/trace.log" || { echo "kernel-trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
cp "$OUT/trace/t_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
cd $R
bash tools/pmc_passes.sh gpurun_out/profile_$TAG/pmc -- python3 $R/bench.py --steps 10 --warmup 2 $ARGS > /dev/null
: > "$OUT/pmc_summary.txt"
IFS='|' read -ra PATS <<< "$PAT"
for P1 in "${PATS[@]}"; do
  echo "== $P1" >> "$OUT/pmc_summary.txt"
  python3 tools/pmc_summary.py "$OUT/pmc" "$P1" >> "$OUT/pmc_summary.txt"
done
python3 - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
vals = {}
for line in open(out + "/pmc_summary.txt"):
    m = re.match(r"(\S+)\s+mean\s+([\d.]+)", line)
    if m:
        vals[m.group(1)] = vals.get(m.group(1), 0.0) + float(m.group(2))     # summed over the product's kernels
bench = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
fetch, write = vals.get("FETCH_SIZE"), vals.get("WRITE_SIZE")
t = {"workload": bench["config"]["workload"], "kernel": bench["roofline"]["kernel"],
     "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
     "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0 if fetch and write else None,
     "alg_bytes_per_launch": bench["roofline"]["alg_bytes_per_launch"] * (bench["roofline"].get("launches_per_product", 1) if " + " in bench["roofline"]["kernel"] else 1),
     "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE as read",
     "counters": vals}
if t["traffic_bytes_per_launch"]:
    t["traffic_over_algorithmic"] = t["traffic_bytes_per_launch"] / t["alg_bytes_per_launch"]
json.dump(t, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps({k: t[k] for k in ("traffic_bytes_per_launch", "alg_bytes_per_launch", "traffic_over_algorithmic") if k in t}))
PY
# keep the summaries, drop the bulky per-dispatch traces (gpurun copies back at most 64 MiB)
rm -rf "$OUT/trace" "$OUT"/pmc/*_kernel_trace.csv "$OUT"/pmc/*_counter_collection.csv "$OUT"/pmc/*_agent_info.csv
cut -c1-150 "$OUT/kernel_stats.csv" | head -6
cat "$OUT/bench.json" | cut -c1-700
