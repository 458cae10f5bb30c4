#!/bin/bash
# tools/profile_eighth.sh TAG [LOCAL_ROWS] -- on the GPU box: one rank's share of BASELINE config 4 at N = 8 (the first 1.25 M rows
# of the 10 M x 10 M x 32 matrix, all 10 M columns) through the column sweep as AUTO plans it: rocprofv3 --kernel-trace --stats and
# separate --pmc passes (FETCH_SIZE; WRITE_SIZE; TCC hit / miss).  VERDICT r04 item 4 asked for exactly this block profiled alone.
# Writes gpurun_out/profile_TAG/{kernel_stats.csv, pmc_summary.txt, run.txt}
set -u
TAG=${1:-r05_config4_eighth}; L=${2:-1250000}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p "$OUT"
CMD="python3 $R/tools/exp_colsweep.py --workload uniform --local-rows $L --rb 0 --g 0"
$CMD > "$OUT/run.txt" 2>&1 || { echo "run failed"; tail -5 "$OUT/run.txt"; exit 1; }
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.log" 2>&1 || echo "kernel-trace pass failed"
cp "$OUT/trace/t_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
: > "$OUT/pmc_summary.txt"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rm -rf "$OUT/p$i"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -o p -- $CMD > "$OUT/p$i.log" 2>&1 || echo "pass $i ($grp) failed"
  python3 - "$OUT/p$i" >> "$OUT/pmc_summary.txt" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "csr_colsweep" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("%-32s mean per launch %16.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
  rm -rf "$OUT/p$i"
done
rm -rf "$OUT/trace"
grep -h "colsweep" "$OUT/run.txt" "$OUT/kernel_stats.csv" | cut -c1-200
cat "$OUT/pmc_summary.txt"
