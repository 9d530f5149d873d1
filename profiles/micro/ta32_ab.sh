#!/bin/bash
# tiles of 32 against 16 marching steps (three lanes per cell, 2 x 2 wavefronts) on small grids: ms per sweep, 16 after 16 and 64 after 64
# Run ON THE GPU BOX: bash profiles/micro/ta32_ab.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(d["ms_per_step"],4), "ms/step; kernel", d["roofline"].get("kernel"), "ms per sweep in the kernel", round(d["roofline"]["avg_launch_us"]*d["roofline"]["launches_per_sweep"]/1000,4))'
for N in ${SIZES:-64 128 256 320 384 512}; do
  for A in fast strict; do
    for TA in 16 32; do
      for KW in "16 16" "64 64"; do
        set -- $KW
        echo -n "N=$N $A TA=$TA steps $1 warmup $2: "; LSF_GS_SKEW_W=2x2 LSF_GS_SKEW_TA=$TA timeout -k 10 300 python3 bench.py --size $N --steps $1 --warmup $2 --arith $A --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "$J"
      done
    done
  done
done
