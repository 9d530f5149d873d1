#!/bin/bash
# Run ON THE GPU BOX: lower bound of what a hand-off finer than a tile could reach (VERDICT r4 item 2 b).  Experiment build
# (make -C levelsetfortran_amd/csrc OUT=../../build/exp/liblsf_ef.so EXTRA=-DLSF_EXPERIMENTS); LSF_PROBE_EARLY_FLAG = t: every tile of the dataflow launch raises its flag in front of
# marching step t, before anything of it is stored -- the fields come out wrong on purpose, only the launch's time is read.
export LSF_LIB_PATH=$PWD/build/exp/liblsf_ef.so
for N in ${SIZES:-128 256 512}; do
  for A in fast strict; do
    for T in 0 12 8 4 1; do
      if [ $T = 0 ]; then unset LSF_PROBE_EARLY_FLAG; else export LSF_PROBE_EARLY_FLAG=$T; fi
      python3 bench.py --size $N --steps 64 --warmup 64 --arith $A --no-cpu-baseline --no-secondary --no-sizes 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('N=$N $A flag in front of step $T: %.4f ms per step, kernel %.4f ms per sweep (%s)' % (d['ms_per_step'], r['avg_launch_us'] * r['launches_per_sweep'] / 1e3, r['kernel']))"
    done
  done
done
