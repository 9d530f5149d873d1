#!/bin/bash
# Run ON THE GPU BOX: A/B of the XCD-aware block numbering of k_reinit_jacobi_strict_sh (round 6) against the library
# built from the commit before it (build/exp/liblsf_old.so), alternating on one box; kernel ms per sweep from the library's events.
J='import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]
print("%.4f ms per step, kernel %.4f ms per sweep (%s)" % (d["ms_per_step"], r["avg_launch_us"] * r["launches_per_sweep"] / 1e3, r["kernel"]))'
for N in ${SIZES:-512 256}; do
  for REP in 1 2; do
    for L in old new; do
      if [ $L = old ]; then export LSF_LIB_PATH=$PWD/build/exp/liblsf_old.so; else unset LSF_LIB_PATH; fi
      echo -n "N=$N strict jacobi $L: "; python3 bench.py --size $N --mode jacobi --arith strict --steps 32 --warmup 16 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$J"
    done
  done
done
