#!/bin/bash
# A/B of the exact-GS tile shapes: three lanes per cell (2x2) against one lane per cell (c1x4 ...), FAST and STRICT
N=${1:-512}
for W in 2x2 c1x4 c1x2 c1x3; do
  for A in fast strict; do
    echo -n "size $N W $W $A: "
    LSF_GS_SKEW_W=$W python3 bench.py --size $N --arith $A --steps 32 --warmup 16 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4), 'ms', '%.3g'%d['value'], d['roofline']['kernel'], round(d['roofline']['frac'],4))"
  done
done
