#!/bin/bash
# tools/dist_rehearsal.sh -- on the one-GPU box: bench.py's multi-rank code paths end to end (NOT a scaling measurement):
#   1. one rank through RCCL (SMVP_FORCE_DIST=1): communicator, chunked exchange, C-layer leg in-process
#   2. two ranks sharing the card over gloo: the C-layer leg in a child process of rank 0 (virtual ranks), the chunk choice from
#      measured products and gathers, the speed-up keys
# Writes gpurun_out/r04/dist_rehearsal.txt
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04; mkdir -p "$OUT"
F=$OUT/dist_rehearsal.txt
: > "$F"
show() {
python3 - "$1" >> "$F" <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not lines:
    print("NO JSON LINE"); sys.exit(0)
j = json.loads(lines[-1])
print(json.dumps({k: j[k] for k in ("metric", "value", "n_gpus", "ms_per_step", "scaling")} | {"config": j["config"]["workload"][-80:]}))
for k in ("config4", "config4_c_layer", "headline_products_only"):
    if k in j["roofline"]["others"]:
        print("roofline.others.%s: %s" % (k, json.dumps(j["roofline"]["others"][k])))
PY
}
cd $R
echo '$ SMVP_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 10 --warmup 2 --no-random-model --no-samples --no-cpu-baseline --no-tjds --no-pwt-tiled --chunks 4   (one rank, backend nccl = RCCL)' >> "$F"
SMVP_FORCE_DIST=1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 10 --warmup 2 --no-random-model --no-samples --no-cpu-baseline --no-tjds --no-pwt-tiled --chunks 4 > "$OUT/reh1.out" 2> "$OUT/reh1.err" || { echo "rehearsal 1 failed" >> "$F"; tail -5 "$OUT/reh1.err" >> "$F"; }
show "$OUT/reh1.out"
echo >> "$F"
echo '$ SMVP_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 5 --warmup 1 --copies 64 --rows 2000000 --no-cpu-baseline   (two ranks sharing the card, gloo; the C-layer leg in a child of rank 0 with two virtual ranks)' >> "$F"
SMVP_DIST_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 5 --warmup 1 --copies 64 --rows 2000000 --no-cpu-baseline > "$OUT/reh2.out" 2> "$OUT/reh2.err" || { echo "rehearsal 2 failed" >> "$F"; tail -8 "$OUT/reh2.err" >> "$F"; }
show "$OUT/reh2.out"
cat "$F" | cut -c1-400
