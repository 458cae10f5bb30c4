#!/bin/bash
# tools/pmc_passes.sh OUTDIR -- CMD...   : one rocprofv3 --pmc pass per counter group (separate runs, kernel-trace only)
# usage on the GPU box:  bash tools/pmc_passes.sh gpurun_out/pmc_x -- python3 tools/exp_tiled.py --variants stream:2048 --steps 5
set -u
OUT=$1; shift; shift
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$GRAFT_REPO_ROOT/$OUT" -o "p$i" -- "$@" > "$GRAFT_REPO_ROOT/$OUT/p$i.log" 2>&1 || echo "pass $i ($grp) failed rc=$?"
done
ls "$GRAFT_REPO_ROOT/$OUT"
