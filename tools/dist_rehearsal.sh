#!/bin/bash
# tools/dist_rehearsal.sh -- on the one-GPU box: the command shape the driver uses for the scaling runs, `python3 bench.py --gpus N
# --steps 20 --warmup 5` with NO launcher around it, end to end (NOT a scaling measurement: the ranks share one card over gloo).
#   1. --gpus 6 at reduced size (six ranks is what this pool's process guard lets one job put on a card; the driver's N = 8
#      differs in nothing but the number: same self-launch, same legs, the C-layer child with N virtual ranks)
#   2. --gpus 2 at BASELINE config 4's FULL size (10 M rows): per-rank set-up here is 4x what a rank of eight builds, so the
#      per-leg seconds bound the real run's from above (the exchange over gloo through host memory does not: ignore tN)
#   3. --gpus 4 with a hard deadline that falls into the legs: the watchdog prints the line from what has been measured, every rank
#      exits 0, nothing is left running
#   4. one rank through RCCL (SMVP_FORCE_DIST=1 under torch.distributed.run): communicator + chunked exchange
# Each: exit code, wall seconds, the line's length, its n_gpus / config4_* / exchange_* / leg_seconds keys, and a process census.
# Writes gpurun_out/r06/dist_rehearsal.txt
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06; mkdir -p "$OUT"
F=$OUT/dist_rehearsal.txt
: > "$F"
show() {   # $1 = stdout file, $2 = wall seconds
python3 - "$1" "$2" >> "$F" <<'PY'
import json, sys
lines = open(sys.argv[1]).read().splitlines()
print("stdout: %d line(s); wall %s s" % (len(lines), sys.argv[2]))
if not lines or not lines[-1].startswith("{"):
    print("NO JSON LINE"); sys.exit(0)
j = json.loads(lines[-1])
r = j["roofline"]
print("line: %d characters; nested objects below roofline/config/cpu_baseline: %s" % (
    len(lines[-1]), [k for o in (r, j["config"], j["cpu_baseline"] or {}) for k, v in o.items() if isinstance(v, (dict, list))]))
print(json.dumps({k: j[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling")} | {"config.workload": j["config"]["workload"][:60]}))
keys = [k for k in r if k.startswith(("config4_t", "config4_speedup", "config4_chunks", "config4_c_layer_step", "config4_c_layer_speedup",
                                      "exchange", "c_layer_", "rccl_ranks", "dist_backend", "self_launched", "n_gpus", "frac", "headline_products",
                                      "leg_seconds", "wall_s", "watchdog")) or k.endswith("_error")]
print(json.dumps({k: r[k] for k in keys}))
PY
}
census() {
  echo "processes of this user after the run ($1):" >> "$F"
  ps -u "$(id -u)" -o pid,ppid,etime,cmd --no-headers | grep -v -E "ps -u|dist_rehearsal|grep|bash -o pipefail|sleep|cut -c" | cut -c1-160 >> "$F"
  echo "(end of census)" >> "$F"
}
run() {    # $1 = tag, rest = command
  local tag=$1; shift
  echo "\$ $*" >> "$F"
  local t0=$(date +%s.%N)
  "$@" > "$OUT/reh_$tag.out" 2> "$OUT/reh_$tag.err"
  local rc=$?
  local t1=$(date +%s.%N)
  echo "exit code $rc" >> "$F"
  grep -h "without a launcher\|self-launched\|hard deadline\|skipped" "$OUT/reh_$tag.err" | cut -c1-220 >> "$F"
  show "$OUT/reh_$tag.out" "$(python3 -c "print(round($t1 - $t0, 1))")"
  census "$tag"
  echo >> "$F"
}
cd $R
export SMVP_DIST_BACKEND=gloo
run gloo6 timeout -k 10 900 python3 bench.py --gpus 6 --steps 20 --warmup 5 --rows 1200000 --copies 96
run gloo2_full timeout -k 10 900 python3 bench.py --gpus 2 --steps 20 --warmup 5
run gloo4_deadline timeout -k 10 300 python3 bench.py --gpus 4 --steps 20 --warmup 5 --rows 2000000 --copies 96 --hard-deadline 40
unset SMVP_DIST_BACKEND
SMVP_FORCE_DIST=1 run rccl1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 10 --warmup 2 --no-random-model --no-samples --no-cpu-baseline --no-tjds --no-pwt-tiled --chunks 4
cut -c1-400 "$F"
