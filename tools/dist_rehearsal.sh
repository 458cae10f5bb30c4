#!/bin/bash
# tools/dist_rehearsal.sh -- on the one-GPU box: bench.py's multi-rank code paths end to end (NOT a scaling measurement):
#   1. one rank through RCCL (SMVP_FORCE_DIST=1, torch.distributed.run): communicator, chunked exchange, C-layer leg in-process
#   2. the DRIVER'S command shape: `python3 bench.py --gpus 2` with NO launcher around it -- bench.py starts its two ranks itself
#      (here over gloo, sharing the card); the C-layer leg in a child process of rank 0 (two virtual ranks), the chunk choice from
#      measured products and gathers, the flat roofline.* keys a scaling record needs
#   3. the same with four ranks (gloo), smaller matrices
#   4. what is left running afterwards (BENCH_r02 ... r04 counted one process at the end of the run)
# Writes gpurun_out/r05/dist_rehearsal.txt
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05; mkdir -p "$OUT"
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
r = j["roofline"]
print("roofline (flat scalars, what the driver's parse keeps): " + json.dumps({k: v for k, v in r.items() if not isinstance(v, (dict, list)) and k not in ("note", "traffic_source")}))
for k in ("config4", "config4_c_layer", "headline_products_only"):
    if k in r["others"]:
        print("roofline.others.%s: %s" % (k, json.dumps(r["others"][k])))
print("extra.dist: %s; child_processes_at_exit: %s" % (json.dumps(j["extra"].get("dist")), j["extra"].get("child_processes_at_exit")))
PY
}
census() {
  echo "processes of this user after the run ($1):" >> "$F"
  ps -u "$(id -u)" -o pid,ppid,etime,cmd --no-headers | grep -v -E "ps -u|dist_rehearsal|grep|bash -o pipefail|sleep" | cut -c1-160 >> "$F"
}
cd $R
echo '$ SMVP_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 10 --warmup 2 --no-random-model --no-samples --no-cpu-baseline --no-tjds --no-pwt-tiled --chunks 4   (one rank, backend nccl = RCCL)' >> "$F"
SMVP_FORCE_DIST=1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 10 --warmup 2 --no-random-model --no-samples --no-cpu-baseline --no-tjds --no-pwt-tiled --chunks 4 > "$OUT/reh1.out" 2> "$OUT/reh1.err" || { echo "rehearsal 1 failed" >> "$F"; tail -5 "$OUT/reh1.err" >> "$F"; }
show "$OUT/reh1.out"
echo >> "$F"
echo '$ SMVP_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 --warmup 1 --rows 2000000   (NO launcher: bench.py starts its own two ranks; they share the card over gloo; the C-layer leg in a child of rank 0 with two virtual ranks)' >> "$F"
SMVP_DIST_BACKEND=gloo timeout -k 10 700 python3 bench.py --gpus 2 --steps 5 --warmup 1 --rows 2000000 > "$OUT/reh2.out" 2> "$OUT/reh2.err"; echo "exit code $?" >> "$F"
grep -h "without a launcher\|self-launched" "$OUT/reh2.err" >> "$F"
show "$OUT/reh2.out"
census "gloo 2"
echo >> "$F"
echo '$ SMVP_DIST_BACKEND=gloo python3 bench.py --gpus 4 --steps 5 --warmup 1 --copies 64 --rows 1000000 --no-cpu-baseline   (four self-launched ranks on the one card)' >> "$F"
SMVP_DIST_BACKEND=gloo timeout -k 10 700 python3 bench.py --gpus 4 --steps 5 --warmup 1 --copies 64 --rows 1000000 --no-cpu-baseline > "$OUT/reh3.out" 2> "$OUT/reh3.err"; echo "exit code $?" >> "$F"
show "$OUT/reh3.out"
census "gloo 4"
echo >> "$F"
echo '$ python3 bench.py --steps 20 --warmup 5   (the default run, then the census: what BENCH_rNN.run.procs_at_end could be counting)' >> "$F"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > "$OUT/reh4.out" 2> "$OUT/reh4.err"; echo "exit code $?" >> "$F"
census "default run"
cat "$F" | cut -c1-600
