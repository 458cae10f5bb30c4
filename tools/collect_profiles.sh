#!/bin/bash
# tools/collect_profiles.sh ROUND -- here, after gpurun has merged gpurun_out/: copy the summaries of tools/profile_round.sh into profiles/
RN=${1:-r06}
cd "$(dirname "$0")/.."
for d in gpurun_out/profile_${RN}_*; do
  t=$(basename "$d" | sed 's/^profile_//')
  for f in kernel_stats.csv pmc_summary.txt traffic.json; do [ -f "$d/$f" ] && cp "$d/$f" "profiles/${t}_$f"; done
  [ -f "$d/bench.json" ] && tail -1 "$d/bench.json" > "profiles/${t}_line.json"
done
ls profiles | grep "^${RN}_"
