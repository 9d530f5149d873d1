#!/bin/bash
# one box: tile shapes of the exact ordering at a latency-bound size (default 256^3), library from LSF_LIB_PATH or the product
for W in ${SHAPES:-2x2 4x2 2x4 4x1 2x1 1x2 c1x2 c1x4}; do
  LSF_GS_SKEW_W=$W python3 bench.py --size ${N:-256} --steps 64 --warmup 64 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$W', 'ms/step %.4f' % d['ms_per_step'], 'kernel ms/sweep %.4f' % (d['roofline']['avg_launch_us'] / 64e3))"
done
