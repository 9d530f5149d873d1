#!/bin/bash
# One lane per cell vs three lanes per cell: pure work term (every tile of a sweep in one launch, results wrong:
# experiment builds only, -DLSF_EXPERIMENTS), dataflow time, per-tile wait/work, and the SQ counters of the work term.
# Run ON THE GPU BOX: bash profiles/micro/cell_probe.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-secondary"
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(d["ms_per_step"],4), "ms/step; launches/sweep", d["roofline"]["launches_per_sweep"], "avg launch us", round(d["roofline"]["avg_launch_us"],1))'
for L in ${LIBS:-cu8 cu4 cu16}; do
  export LSF_LIB_PATH=$PWD/build/exp/liblsf_$L.so
  for W in ${SHAPES:-2x2 c1x4}; do
    for A in fast strict; do
      echo -n "$L $W $A nodeps(skew): "; LSF_GS_SKEW_W=$W LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew $B --arith $A 2>/dev/null | python3 -c "$J"
      echo -n "$L $W $A dataflow:     "; LSF_GS_SKEW_W=$W $B --arith $A 2>/dev/null | python3 -c "$J"
    done
    echo -n "$L $W fast tiles: "; LSF_GS_SKEW_W=$W LSF_TRACE_TILES=1 $B 2>&1 | grep "dataflow batch" | tail -1
  done
done
export LSF_LIB_PATH=$PWD/build/exp/liblsf_${SQLIB:-cu8}.so
OUT=gpurun_out/sq_cell; rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
P2="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"
P3="SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"
for W in ${SHAPES:-2x2 c1x4}; do
  for P in 1 2 3; do
    eval PM=\$P$P
    LSF_GS_SKEW_W=$W LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew rocprofv3 --pmc $PM --kernel-trace --output-format csv -d $OUT/p${P}_$W -- python3 bench.py --steps 2 --warmup 0 --mode gs --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/p${P}_$W.log
  done
done
python3 - $OUT ${SHAPES:-2x2 c1x4} <<'PY'
import csv, glob, json, os, sys
out=sys.argv[1]
for V in sys.argv[2:]:
    tot={}
    for p in ("p1","p2","p3"):
        for f in glob.glob(os.path.join(out,f"{p}_{V}","**","*counter_collection.csv"),recursive=True):
            for r in csv.DictReader(open(f)):
                if "gs_skew" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]]=tot.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
    w=tot.get("SQ_WAVES",1)
    print(V, json.dumps({k:round(v/w,1) for k,v in tot.items()}), "waves",w)
PY
