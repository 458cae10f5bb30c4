#!/bin/bash
# tools/profile_round.sh ROUND -- on the GPU box: the round's profiles of every workload the bench line carries, each through
# tools/profile_bench.sh (ONE process gives the kernel trace and the bench line; --pmc passes separately):
#   headline CSR, headline TJDS, pwt x459 CSR / TJDS, BASELINE config 4 (column sweep), the SURVEY 8(d) random model (binned plan)
# Writes gpurun_out/profile_<ROUND>_*/ ; copy the summaries into profiles/ (tools/collect_profiles.sh).
set -u
RN=${1:-r06}
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_bench.sh ${RN}_bench
PAT="csr_stream_owner<8, 4" BENCH_ARGS="--format tjds" bash tools/profile_bench.sh ${RN}_bench_tjds
BENCH_ARGS="--workload pwt_tiled" bash tools/profile_bench.sh ${RN}_pwt_tiled
PAT="csr_stream_owner<8, 4" BENCH_ARGS="--workload pwt_tiled --format tjds" bash tools/profile_bench.sh ${RN}_pwt_tiled_tjds
PAT="csr_colsweep" BENCH_ARGS="--workload uniform32 --steps 40" bash tools/profile_bench.sh ${RN}_config4
PAT="csr_near_window|csr_binned_far_products|csr_binned_far_sums" BENCH_ARGS="--workload memplus_shaped" bash tools/profile_bench.sh ${RN}_random_model
