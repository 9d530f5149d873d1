#!/bin/bash
# the dataflow launch with fewer resident blocks than the device holds (LSF_GS_BLOCKS): is a sweep bound by slots or by its chain?
# usage: stream_blocks.sh size shape arith "blocks..." [cont=0]
N=$1; S=$2; A=$3
for B in $4; do
  echo -n "$N^3 $A $S cont=${5:-0} blocks=$B: "
  LSF_GS_STREAM=1 LSF_GS_BLOCKS=$B LSF_GS_CONT=${5:-0} LSF_GS_SKEW_W=$S python3 bench.py --size $N --steps 64 --warmup 64 --arith $A --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4), 'ms/step')"
done
