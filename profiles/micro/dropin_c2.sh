#!/bin/bash
# config 2 through the drop-in to convergence, resident = 2 vs 0 (wall clock of the whole program)
set -u
cd "$GRAFT_REPO_ROOT"
W=/tmp/c2run; rm -rf $W; mkdir -p $W; cd $W
python3 - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import stl_io
s = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests/golden/surfaces.npz"))
stl_io.stl_write("cube40.stl", s["cube40_surfX"], s["cube40_surfElem"])
PY
# one discarded run first (see dropin_c3.sh)
bash -c "ulimit -s unlimited; LSF_DX=0.008565310492505354 LSF_REINIT_ITER=3 LSF_MINMAX_ITER=0 LSF_REINIT2_ITER=0 $GRAFT_REPO_ROOT/build/dropin/set3d_hip.exec cube40.stl > /dev/null 2>&1"
for R in 2 0; do
  T0=$(date +%s.%N)
  bash -c "ulimit -s unlimited; LSF_TRACE=1 LSF_ARITH=${ARITH:-strict} LSF_RESIDENT=$R LSF_DX=0.008565310492505354 LSF_MINMAX_ITER=0 LSF_REINIT2_ITER=0 $GRAFT_REPO_ROOT/build/dropin/set3d_hip.exec cube40.stl > out_$R.txt 2> err_$R.txt"
  T1=$(date +%s.%N)
  echo "arith=${ARITH:-strict} resident=$R wall $(python3 -c "print(round($T1-$T0,2))") s" >> $GRAFT_REPO_ROOT/gpurun_out/c2_dropin.txt
  grep -E "Grid Size|Run Time|steady|Asymptotic" out_$R.txt >> $GRAFT_REPO_ROOT/gpurun_out/c2_dropin.txt
  grep -E "^\[lsf\] <-" err_$R.txt >> $GRAFT_REPO_ROOT/gpurun_out/c2_dropin.txt
  ls -la *.vti >> $GRAFT_REPO_ROOT/gpurun_out/c2_dropin.txt
done
