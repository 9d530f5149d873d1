#!/bin/bash
# per-tile phase times (experiment build): bash profiles/micro/cell_phases.sh
export LSF_LIB_PATH=$PWD/build/exp/liblsf_${LIB:-cu8}.so
B="python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-secondary"
for W in ${SHAPES:-2x2 c1x4}; do
  for A in ${ARITH:-fast strict}; do
  echo "== $W $A nodeps"; LSF_GS_SKEW_W=$W LSF_TRACE_TILES=1 LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew $B --arith $A 2>&1 | grep "tile phases" | tail -1
  echo "== $W $A dataflow"; LSF_GS_SKEW_W=$W LSF_TRACE_TILES=1 $B --arith $A 2>&1 | grep "\[lsf\]" | tail -2
  done
done
