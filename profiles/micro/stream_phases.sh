#!/bin/bash
# tile phase times (experiment build build/exp/liblsf_x.so: -DLSF_EXPERIMENTS) of the dataflow launches, strict c1x4 / fast 2x2 and c1x4
# usage: stream_phases.sh [size=512] [sweeps=32]
N=${1:-512}; K=${2:-32}
export LSF_LIB_PATH=$PWD/build/exp/liblsf_x.so
for CFG in "strict c1x4" "fast c1x4" "fast 2x2"; do
  set -- $CFG
  for V in "LSF_GS_STREAM=0" "LSF_GS_STREAM=1" "LSF_GS_STREAM=1 LSF_GS_CONT=0"; do
    echo "== $N^3 $1 $2 $V"
    env $V LSF_GS_SKEW_W=$2 LSF_TRACE_TILES=1 python3 bench.py --size $N --steps $K --warmup $K --arith $1 --no-cpu-baseline --no-secondary 2>&1 | grep -E "^\[lsf\]|ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/ms_per_step \1/' | tail -4
  done
done
