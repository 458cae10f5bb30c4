#!/bin/bash
# tools/exp_binned_prof.sh TAG [exp_binned.py args] -- exp_binned.py under rocprofv3 --kernel-trace --stats; prints the product's kernels
set -u
TAG=${1:-x}; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/xb_trace_$TAG" -o t -- python3 $R/tools/exp_binned.py "$@" > "$OUT/xb_$TAG.log" 2>&1 || { echo "failed"; tail -20 "$OUT/xb_$TAG.log"; exit 1; }
grep -A1 "^csr_\|error vs" "$OUT/xb_$TAG.log"
cp "$OUT/xb_trace_$TAG/t_kernel_stats.csv" "$OUT/xb_$TAG.kernel_stats.csv"; rm -rf "$OUT/xb_trace_$TAG"
python3 - "$OUT/xb_$TAG.kernel_stats.csv" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if "csr_stream_owner" in n or "csr_binned" in n or "csr_colsweep" in n or "csr_near" in n:
        short = n[n.find("csr_"):].split("(")[0]
        print("  %-50s calls %4s  avg %8.1f us  min %8.1f  max %8.1f" % (short[:50], row["Calls"], float(row["AverageNs"]) / 1e3,
                                                                    float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
