#!/bin/bash
# wall clock of the drop-in executables on the GPU box: configuration 2 (dropin_c2.sh, STRICT and FAST) and cube40 as shipped through the
# full drop-in and through the seams alone.  bash profiles/micro/dropin_times.sh > gpurun_out/dropin_times.txt
cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/c2_dropin.txt
bash profiles/micro/dropin_c2.sh > /dev/null 2>&1
ARITH=fast bash profiles/micro/dropin_c2.sh > /dev/null 2>&1
grep -E "wall|lsf_reinit|lsf_phi0|write_vti" gpurun_out/c2_dropin.txt
W=/tmp/seams; rm -rf $W; mkdir -p $W; cd $W
python3 - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import stl_io
s = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests/golden/surfaces.npz"))
stl_io.stl_write("cube40.stl", s["cube40_surfX"], s["cube40_surfElem"])
PY
for E in set3d_hip set3d_hip_seams set3d_hip set3d_hip_seams; do
  T0=$(date +%s.%N)
  bash -c "ulimit -s unlimited; LSF_ARITH=strict $GRAFT_REPO_ROOT/build/dropin/$E.exec cube40.stl > out_$E.txt 2>&1"
  T1=$(date +%s.%N)
  echo "$E cube40 as shipped (62^3: 2155 sweeps, 406 min/max iterations, second reinit): wall $(python3 -c "print(round($T1-$T0,2))") s; $(grep -c Iteration out_$E.txt) iteration lines"
done
