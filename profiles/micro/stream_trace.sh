#!/bin/bash
# per-tile statistics of the dataflow launch (LSF_TRACE_TILES): continuation counts and take+wait / work+publish times
# usage: stream_trace.sh [size=512] [sweeps=64] [shapes="default c1x4"] [ariths="fast strict"]   (LSF_LIB_PATH selects an experiment build)
N=${1:-512}; K=${2:-64}
for A in ${4:-fast strict}; do
  for S in ${3:-default c1x4}; do
    for V in "LSF_GS_STREAM=0" "LSF_GS_STREAM=1" "LSF_GS_STREAM=1 LSF_GS_CONT=0"; do
      echo "== $N^3 $A $S $V"
      SH=""; [ "$S" != default ] && SH="LSF_GS_SKEW_W=$S"
      env $V $SH LSF_TRACE_TILES=1 python3 bench.py --size $N --steps $K --warmup $K --arith $A --no-cpu-baseline --no-secondary 2>&1 | grep -E "^\[lsf\]|ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/ms_per_step \1/' | tail -4
    done
  done
done
