#!/bin/bash
# tools/profile_cli_samples.sh TAG [extra CLI flags] -- on the GPU box: rocprofv3 --kernel-trace --stats of the C host
# program on the reference's own inputs (BASELINE configs 2/3/5): smvp-toolkit-cli -c -t -n 1000 on memplus.mtx and pwt.mtx.
# Puts kernel time (rocprof) beside the per-product window the program itself reports.
set -u
TAG=${1:-r02}; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/cli_$TAG
mkdir -p "$OUT/reports"
for m in ibm32 memplus pwt; do
  [ -f /tmp/$m.mtx ] || { [ -f $R/tests/golden/sample-data/$m.mtx ] && cp $R/tests/golden/sample-data/$m.mtx /tmp/$m.mtx || gunzip -c $R/tests/golden/sample-data/$m.mtx.gz > /tmp/$m.mtx; }
done
cd /tmp; export TMPDIR=/tmp
for m in ibm32 memplus pwt; do
  $R/smvp-toolkit_amd/bin/smvp-toolkit-cli -c -t -n 1000 -d "$OUT/reports" "$@" /tmp/$m.mtx > "$OUT/${m}_plain.log" 2>&1 || echo "plain run of $m failed"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$m" -o t -- \
      $R/smvp-toolkit_amd/bin/smvp-toolkit-cli -c -t -n 1000 -d "$OUT/reports" "$@" /tmp/$m.mtx > "$OUT/${m}_rocprof.log" 2>&1 || echo "rocprof run of $m failed"
  cp "$OUT/trace_$m/"*kernel_stats.csv "$OUT/${m}_kernel_stats.csv" 2>/dev/null
  rm -rf "$OUT/trace_$m"
  echo "== $m"; grep -h -E "Average|avg|ms" "$OUT/${m}_plain.log" | head -12
  cut -c1-160 "$OUT/${m}_kernel_stats.csv" | head -8
done
grep -h "Time" "$OUT"/reports/*.txt | head -40
rm -rf "$OUT/reports"
