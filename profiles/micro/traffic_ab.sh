#!/bin/bash
# HBM traffic of the exact-GS dataflow kernel for experiment builds of the library: bash profiles/micro/traffic_ab.sh "wide narrow" [shape]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in $1; do
  export LSF_LIB_PATH=$PWD/build/exp/liblsf_$L.so
  for C in FETCH_SIZE WRITE_SIZE; do
    OUT=gpurun_out/traffic_ab/${L}_$C; rm -rf $OUT; mkdir -p $OUT
    LSF_GS_SKEW_W=${2:-2x2} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 32 --warmup 32 --mode gs --no-cpu-baseline --no-secondary > /dev/null 2> $OUT.log
  done
  python3 - $L <<'PY'
import csv, glob, sys
L = sys.argv[1]
tot = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/traffic_ab/{L}_{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_reinit_gs_persist" in r["Kernel_Name"] and r["Counter_Name"] == C:
                tot[C] = tot.get(C, 0.0) + float(r["Counter_Value"])
f, w = tot.get("FETCH_SIZE", 0) * 1024 / 64, tot.get("WRITE_SIZE", 0) * 1024 / 64  # KiB over 64 sweeps -> bytes per sweep
alg = 24.0 * 510 ** 3
print(f"{L}: per sweep fetch (x2 gfx950 correction) {2 * f / 1e9:.2f} GB, write {w / 1e9:.2f} GB, total {(2 * f + w) / 1e9:.2f} GB = {(2 * f + w) / alg:.2f} x algorithmic")
PY
done
