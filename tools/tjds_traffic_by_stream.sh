#!/bin/bash
# tools/tjds_traffic_by_stream.sh -- on the GPU box: what each stream of the one-kernel TJDS product moves.
# Builds diagnostic libraries with ONE stream of csr_stream_owner<8, 4> taken out at a time (-DSMVP_TJDS_NEUTRALISE=mask: wrong
# results, that is the point), runs the product on memplus x944 under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
# --kernel-trace only) and prints the per-product means; the difference to the full kernel is that stream's traffic.
# Then the normal library with the value cache's threshold at 0 (off), 1 (everything), 2 (default), 4.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tjds_by_stream
mkdir -p "$OUT" /tmp/smvp_diag
cd /tmp; export TMPDIR=/tmp
PKG=$R/smvp-toolkit_amd
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$R/include -I$PKG/csrc"
pass() {  # pass LIB LABEL VARIANT
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/smvp_diag/p
    SMVP_LIB_PATH=$1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/smvp_diag/p -o p -- \
        python3 $R/tools/exp_tjds.py --workloads memplus_tiled --variants $3 --steps 2 > /tmp/smvp_diag/run_$ctr.log 2>&1 || echo "pass failed: $2 $ctr"
    python3 - "$2" $ctr <<'PY'
import csv, glob, sys
vals = []
for f in glob.glob("/tmp/smvp_diag/p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "csr_stream_owner<8, 4" in row["Kernel_Name"] and row["Counter_Name"] == sys.argv[2]:
            vals.append(float(row["Counter_Value"]))
scale = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0    # gfx950 tallies 128-byte read requests at 64 B (MI355X_MICROARCH, HBM)
print("%-44s %-10s %9.1f MB per product (n=%d)" % (sys.argv[1], sys.argv[2], scale * sum(vals) / max(len(vals), 1) / 1024.0, len(vals)), flush=True)
PY
  done
  grep -h "^gather" /tmp/smvp_diag/run_WRITE_SIZE.log | cut -c1-150
}
for mask in 0 1 2 4 8 3 7 15; do
  mkdir -p /tmp/smvp_diag/b$mask
  /opt/rocm/bin/hipcc $FLAGS -DSMVP_TJDS_NEUTRALISE=$mask -c $PKG/csrc/smvp_kernels.hip -o /tmp/smvp_diag/b$mask/smvp_kernels.o || exit 1
  OBJ=$(ls $PKG/build/*.o | grep -v smvp_kernels.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/smvp_diag/b$mask/libsmvp_amd.so $OBJ /tmp/smvp_diag/b$mask/smvp_kernels.o -lpthread -ldl || exit 1
  pass /tmp/smvp_diag/b$mask/libsmvp_amd.so "neutralised mask $mask (1 x_perm, 2 val in place, 4 val cache, 8 overflow)" gather:half:0:2
done 2>&1 | tee "$OUT/by_stream.txt"
for cache in 0 1 2 4; do
  pass $PKG/lib/libsmvp_amd.so "normal build, value cache threshold $cache" gather:half:0:$cache
done 2>&1 | tee "$OUT/by_cache.txt"
