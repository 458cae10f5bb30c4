#!/bin/bash
# tools/pmc_tjds.sh TAG VARIANT KERNEL_SUBSTRING [WORKLOAD] -- PMC passes over one TJDS variant of tools/exp_tjds.py
set -u
TAG=$1; VAR=$2; PAT=$3; WL=${4:-memplus_tiled}
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_passes.sh gpurun_out/pmc_$TAG -- python3 $R/tools/exp_tjds.py --workloads $WL --variants $VAR --steps 2 > /dev/null
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$TAG "$PAT" > $R/gpurun_out/pmc_$TAG/summary.txt
rm -f $R/gpurun_out/pmc_$TAG/*_kernel_trace.csv $R/gpurun_out/pmc_$TAG/*_counter_collection.csv $R/gpurun_out/pmc_$TAG/*_agent_info.csv
cat $R/gpurun_out/pmc_$TAG/summary.txt
