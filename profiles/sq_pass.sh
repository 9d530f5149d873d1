#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash profiles/sq_pass.sh r01
# SQ issue/stall counters of the two sweep kernels (own PMC passes, --kernel-trace only), at the STEADY STATE of the
# headline: 64 sweeps after 64 (round 1 profiled 2-sweep launches, i.e. fill and drain).
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
for M in ${SQ_MODES:-gs jacobi}; do
  rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d "$OUT/p1_$M" -- python3 bench.py --steps ${SQ_STEPS:-64} --warmup ${SQ_STEPS:-64} --mode $M --no-cpu-baseline --no-secondary > /dev/null 2> "$OUT/p1_$M.log"
  rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d "$OUT/p2_$M" -- python3 bench.py --steps ${SQ_STEPS:-64} --warmup ${SQ_STEPS:-64} --mode $M --no-cpu-baseline --no-secondary > /dev/null 2> "$OUT/p2_$M.log"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for mode, kern in (("gs", "k_reinit_gs_"), ("jacobi", "k_reinit_jacobi")):
    tot = {}
    for p in ("p1", "p2"):
        for f in glob.glob(os.path.join(out, f"{p}_{mode}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if tot.get("SQ_WAVES"):
        tot["per_wave"] = {k: round(v / tot["SQ_WAVES"], 1) for k, v in tot.items() if k != "SQ_WAVES"}
    res[kern] = tot
json.dump(res, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
json.dump(res, open(os.path.join(out, "sq_counters.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
