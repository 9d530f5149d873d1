#!/bin/bash
# Run ON THE GPU BOX: bash profiles/micro/sq_jacobi.sh  -- SQ counters of the fp64 Jacobi sweep kernels, per block shape
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sqj
rm -rf "$OUT"; mkdir -p "$OUT"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
P3="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"
for SH in ${SHAPES:-0 4x1 1x4}; do
  LSF_JAC_SH=$SH rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d "$OUT/p1_$SH" -- python3 bench.py --steps 8 --warmup 8 --mode jacobi --no-cpu-baseline --no-secondary > /dev/null 2> "$OUT/p1_$SH.log"
  LSF_JAC_SH=$SH rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d "$OUT/p2_$SH" -- python3 bench.py --steps 8 --warmup 8 --mode jacobi --no-cpu-baseline --no-secondary > /dev/null 2> "$OUT/p2_$SH.log"
  LSF_JAC_SH=$SH rocprofv3 --pmc $P3 --kernel-trace --output-format csv -d "$OUT/p3_$SH" -- python3 bench.py --steps 8 --warmup 8 --mode jacobi --no-cpu-baseline --no-secondary > /dev/null 2> "$OUT/p3_$SH.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "p1_*"))):
    if not os.path.isdir(d): continue
    sh = os.path.basename(d)[3:]
    tot, n = {}, 0
    for p in ("p1", "p2", "p3"):
        for f in glob.glob(os.path.join(out, f"{p}_{sh}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_reinit_jacobi" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if tot.get("SQ_WAVES"):
        tot["per_wave"] = {k: round(v / tot["SQ_WAVES"], 1) for k, v in tot.items() if k != "SQ_WAVES"}
    res[sh] = tot
json.dump(res, open(os.path.join(out, "sq_jacobi.json"), "w"), indent=1)
for sh, t in res.items():
    print(sh, json.dumps(t.get("per_wave")))
PY
