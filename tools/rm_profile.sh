#!/bin/bash
# tools/rm_profile.sh TAG [bench args] -- on the GPU box: the SURVEY 8(d) random model through bench.py under
# rocprofv3 --kernel-trace --stats; prints the bench line's roofline and the per-kernel averages of the product's kernels.
# Writes gpurun_out/r04/rm_TAG.{json,log,kernel_stats.csv}.
set -u
TAG=${1:-x}; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04
mkdir -p "$OUT"
ARGS="--workload memplus_shaped --no-tjds --no-samples --no-config4 --no-pwt-tiled --no-cpu-baseline --no-c-layer --no-live-traffic --steps 50 --warmup 5 $*"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rm_trace_$TAG" -o t -- python3 $R/bench.py $ARGS > "$OUT/rm_$TAG.json" 2> "$OUT/rm_$TAG.log" || { echo "bench failed (a debug build gives wrong results: expected then)"; tail -3 "$OUT/rm_$TAG.log"; }
cp "$OUT/rm_trace_$TAG/t_kernel_stats.csv" "$OUT/rm_$TAG.kernel_stats.csv"
rm -rf "$OUT/rm_trace_$TAG"
python3 - "$OUT/rm_$TAG.json" "$OUT/rm_$TAG.kernel_stats.csv" <<'PY'
import csv, json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
r = json.loads(lines[-1])["roofline"] if lines else {"ms_per_product": 0, "frac": 0, "kernel": "(no line)"}
print("product: %.4f ms  frac %.4f  kernel %s" % (r["ms_per_product"], r["frac"], r["kernel"]))
for row in csv.DictReader(open(sys.argv[2])):
    n = row["Name"]
    if "csr_stream_owner" in n or "csr_binned" in n or "csr_colsweep" in n or "csr_near" in n:
        print("  %-60s calls %4s  avg %8.1f us  min %8.1f  max %8.1f" % (n.split("(")[0][-60:], row["Calls"], float(row["AverageNs"]) / 1e3,
                                                                    float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
