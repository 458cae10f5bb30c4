#!/bin/bash
# dev: headline CSR (and TJDS) rate vs matrix size -- memplus replicated k times (k * 1.87 MB of algorithmic traffic)
R=$GRAFT_REPO_ROOT
echo "copies rows nnz alg_MB csr_ms csr_GFLOPs csr_frac_of_8TBs tjds_ms tjds_frac"
for k in 1 8 30 59 118 236 472 944 1888 2832; do
  python3 $R/bench.py --copies $k --steps 100 --warmup 20 --no-random-model --no-samples --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
t = j['extra'].get('tjds', {})
print($k, j['config']['rows'], j['config']['nnz'], round(j['roofline']['alg_bytes_per_launch'] / 1e6, 1), j['roofline']['ms_per_launch'], j['value'], j['roofline']['frac'], t.get('ms_per_step'), t.get('frac_of_hbm_peak'))"
done
