#!/bin/bash
# Round 5, VERDICT r4 item 1 (a): what a smaller LDS image of the one-lane-per-cell tile would buy, BEFORE building it.
# Experiment builds (-DLSF_EXPERIMENTS) whose bundle / halo rows are laid out at the pitch a ring-buffered image would have
# (rows overlap: the field comes out wrong, the instruction stream and the LDS traffic are those of the product tile) and that
# are compiled for 3 / 4 wavefronts per SIMD:   x = product layout (22 / 18 entries, 2 tiles per CU)
#   r3 = pitch 14 / 10, 3 tiles per CU     r4 = pitch 10 / 6, 4 tiles per CU
# "nodeps" = every tile of a sweep in one launch, dependencies ignored (the work term); "dataflow" = the product launch.
# Run ON THE GPU BOX: bash profiles/micro/ring_probe.sh > gpurun_out/ring_probe.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-secondary"
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(d["ms_per_step"],4), "ms/step; kernel", d["roofline"].get("kernel"), "avg launch us", round(d["roofline"]["avg_launch_us"],1))'
for L in ${LIBS:-x r3 r4}; do
  export LSF_LIB_PATH=$PWD/build/exp/liblsf_$L.so
  for W in ${SHAPES:-c1x4}; do
    for A in fast strict; do
      echo -n "$L $W $A nodeps(skew): "; LSF_GS_SKEW_W=$W LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew timeout -k 10 120 $B --arith $A 2>/dev/null | python3 -c "$J"
      echo -n "$L $W $A dataflow:     "; LSF_GS_SKEW_W=$W timeout -k 10 120 $B --arith $A 2>/dev/null | python3 -c "$J"
    done
    echo -n "$L $W fast phases nodeps: "; LSF_GS_SKEW_W=$W LSF_TRACE_TILES=1 LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew timeout -k 10 120 $B 2>&1 | grep "tile phases" | tail -n 1
  done
done
