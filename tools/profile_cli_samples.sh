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
wall() {  # wall milliseconds of a command, process start to exit (the report files are written by then)
  local s e; s=$(date +%s%N); "$@" > /dev/null 2>&1; e=$(date +%s%N); echo $(( (e - s) / 1000000 ))
}
for m in ibm32 memplus pwt; do
  # end to end, what the reference's user waits for (its own set-up is main-cli.c:340-365 and the O(nnz * N) loop at :894-904):
  # process start -> both reports written; the first run of a process on a fresh box also pays for HIP start-up
  echo "== $m: end-to-end wall of smvp-toolkit-cli -c -t -n N -d <dir> $m.mtx (ms; three runs each)"
  for n in 1 1000; do
    echo "   -n $n: $(wall $R/smvp-toolkit_amd/bin/smvp-toolkit-cli -c -t -n $n -d "$OUT/reports" "$@" /tmp/$m.mtx) $(wall $R/smvp-toolkit_amd/bin/smvp-toolkit-cli -c -t -n $n -d "$OUT/reports" "$@" /tmp/$m.mtx) $(wall $R/smvp-toolkit_amd/bin/smvp-toolkit-cli -c -t -n $n -d "$OUT/reports" "$@" /tmp/$m.mtx)"
  done
  echo "   -n 1000 --device-convert: $(wall $R/smvp-toolkit_amd/bin/smvp-toolkit-cli -c -t -n 1000 --device-convert -d "$OUT/reports" "$@" /tmp/$m.mtx)"
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
