#!/bin/bash
# three lanes per cell: 16-byte LOADS on deep tiles with the 8-byte stores kept (experiment build wl: -DLSF_SKEW_WIDE=2 -DLSF_SKEW_WIDE_ST=0)
# against the product loader (x), A/B/A/B on one box.  bash profiles/micro/wl_ab.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(d["ms_per_step"],4), "ms/step; kernel ms per sweep", round(d["roofline"]["avg_launch_us"]*d["roofline"]["launches_per_sweep"]/1000,4))'
for R in 1 2; do
  for L in x wl; do
    export LSF_LIB_PATH=$PWD/build/exp/liblsf_$L.so
    for CFG in "512 fast 64 64" "512 fast 20 5" "512 strict 16 8" "256 fast 64 64" "384 fast 32 32"; do
      set -- $CFG
      echo -n "$L N=$1 $2 $3/$4 2x2: "; LSF_GS_SKEW_W=2x2 timeout -k 10 300 python3 bench.py --size $1 --steps $3 --warmup $4 --arith $2 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$J"
    done
  done
done
