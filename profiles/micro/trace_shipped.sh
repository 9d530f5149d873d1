#!/bin/bash
# Run ON THE GPU BOX: the reference's default run (cube40.stl as shipped: 62^3) through the full drop-in with LSF_TRACE=1 -- where its
# 0.8-0.9 s go, per ABI call.  EXTRA="LSF_MINMAX_BAND_MAX=100" etc. adds environment switches to a third run.
W=/tmp/shipped; rm -rf $W; mkdir -p $W; cd $W
python3 - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import stl_io
s = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests/golden/surfaces.npz"))
stl_io.stl_write("cube40.stl", s["cube40_surfX"], s["cube40_surfElem"])
PY
for A in strict fast "strict $EXTRA"; do
set -- $A
T0=$(date +%s.%N)
bash -c "ulimit -s unlimited; env ${2:-LSF_NOP=1} LSF_TRACE=1 LSF_ARITH=$1 $GRAFT_REPO_ROOT/build/dropin/set3d_hip.exec cube40.stl > out.txt 2> err.txt"
T1=$(date +%s.%N)
echo "$A: wall $(python3 -c "print(round($T1-$T0,3))") s"; grep "<-" err.txt | head -30; grep -i "Total Run" out.txt | head -3
done
