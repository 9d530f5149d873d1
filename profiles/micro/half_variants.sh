#!/bin/bash
# Run ON THE GPU BOX: the HALF hand-off (lsf_skew.hpp) in several builds against LSF_GS_HALF=0 of the product library, one box,
# kernel ms per sweep (64 after 64).  bash profiles/micro/half_variants.sh "pub9 pub13 probe1 probe2"  (build/exp/liblsf_<name>.so)
J='import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]
print("%.4f" % (r["avg_launch_us"] * r["launches_per_sweep"] / 1e3), end=" ")'
B="python3 bench.py --steps 64 --warmup 64 --no-cpu-baseline --no-secondary --no-sizes"
for N in ${SIZES:-64 128 256}; do
  for A in fast strict; do
    echo -n "N=$N $A: off "; LSF_GS_HALF=0 $B --size $N --arith $A 2>/dev/null | python3 -c "$J"
    echo -n " product "; LSF_GS_HALF=1 $B --size $N --arith $A 2>/dev/null | python3 -c "$J"
    for V in $1; do
      echo -n " $V "; LSF_GS_HALF=1 LSF_LIB_PATH=$PWD/build/exp/liblsf_$V.so $B --size $N --arith $A 2>/dev/null | python3 -c "$J"
    done
    echo -n " off "; LSF_GS_HALF=0 $B --size $N --arith $A 2>/dev/null | python3 -c "$J"
    echo
  done
done
