#!/bin/bash
# Run ON THE GPU BOX: A/B of the hand-off finer than a tile (LSF_GS_HALF = 0 / 1, lsf_skew.hpp HALF) on ONE box, alternating, the
# protocol of early_flag_probe.sh (kernel ms per sweep from the library's events, 64 sweeps after 64; 16 after 16 at 256^3 too --
# what bench.py's `sizes` entry times).  VERDICT r5 item 3: keep only if 256^3 improves >= 8 % in both arithmetics and 128^3 >= 25 %.
# LIB=path overrides the library (builds with other LSF_HALF_* constants).
[ -n "$LIB" ] && export LSF_LIB_PATH=$LIB
for N in ${SIZES:-64 128 256}; do
  for A in fast strict; do
    for REP in 1 2; do
      for H in 0 1; do
        export LSF_GS_HALF=$H
        python3 bench.py --size $N --steps ${STEPS:-64} --warmup ${WARM:-64} --arith $A --no-cpu-baseline --no-secondary --no-sizes 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('N=$N $A half=$H: %.4f ms per step, kernel %.4f ms per sweep (%s)' % (d['ms_per_step'], r['avg_launch_us'] * r['launches_per_sweep'] / 1e3, r['kernel']))"
      done
    done
  done
done
