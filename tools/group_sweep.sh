#!/bin/bash
# dev: sweep SMVP_TILE_GROUP: time (both workloads) and FETCH_SIZE on the tiled matrix
R=$GRAFT_REPO_ROOT
for g in 1 8 32 128 512 2048; do
  echo "== group $g"
  SMVP_TILE_GROUP=$g python3 $R/tools/exp_tiled.py --matrix memplus.mtx --variants stream:1024 --steps 30 2>&1 | grep csr_
  SMVP_TILE_GROUP=$g python3 $R/tools/exp_csr.py --kind memplus --rows-log2 24 --variants stream:1024 2>&1 | grep csr_
  (cd /tmp && SMVP_TILE_GROUP=$g rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/gs/g$g -o p -- python3 $R/tools/exp_tiled.py --matrix memplus.mtx --variants stream:1024 --steps 3 > /dev/null 2>&1)
  python3 $R/tools/pmc_summary.py $R/gpurun_out/gs/g$g csr_stream_
done
