#!/bin/bash
# exact-GS headline, three sweep counts, both arithmetics: bash profiles/micro/gs_quick.sh [shape]
W=${1:-2x2}
for A in fast strict; do
for S in "--steps 20 --warmup 5" "--steps 32 --warmup 16" "--steps 64 --warmup 64"; do
  echo -n "$W $A $S: "; LSF_GS_SKEW_W=$W python3 bench.py $S --arith $A --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4), 'ms', '%.3g'%d['value'], round(d['roofline']['frac'],4))"
done; done
