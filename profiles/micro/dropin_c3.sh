#!/bin/bash
# BASELINE config 3 (twoCube10, cubic 512^3, SWEEPS reinit sweeps + 200 min/max iterations) through the reference's Fortran
# host: wall clock of the whole program and the time spent inside the library (LSF_TRACE).  bash profiles/micro/dropin_c3.sh [sweeps]
set -u
SW=${1:-128}
cd "$GRAFT_REPO_ROOT"
W=/tmp/c3run; rm -rf $W; mkdir -p $W; cd $W
python3 - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import stl_io
s = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests/golden/surfaces.npz"))
stl_io.stl_write("twoCube10.stl", s["twocube10_surfX"], s["twocube10_surfElem"])
PY
cat > c3.nml <<NML
&lsf_inputs
  dx = 0.024514811031664963
  dd_lo = 10, 234, 234
  dd_hi = 10, 235, 235
  reinit_iter = $((SW - 1))
  minmax_iter = 200
  reinit2_iter = 0
  arith = '${ARITH:-strict}'
/
NML
OUT=$GRAFT_REPO_ROOT/gpurun_out/c3_dropin.txt
# one discarded run first: the first process that allocates 8 GB of device memory on a fresh box pays for it (snapshot and
# narrowBand 30 ms instead of < 1 ms, reinit + 150 ms), later processes do not
bash -c "ulimit -s unlimited; $GRAFT_REPO_ROOT/build/dropin/set3d_hip.exec twoCube10.stl c3.nml > /dev/null 2>&1"
T0=$(date +%s.%N)
bash -c "ulimit -s unlimited; LSF_TRACE=1 $GRAFT_REPO_ROOT/build/dropin/set3d_hip.exec twoCube10.stl c3.nml > out.txt 2> err.txt"
T1=$(date +%s.%N)
echo "config 3, $SW sweeps, arith ${ARITH:-strict}: wall $(python3 -c "print(round($T1-$T0,2))") s" >> $OUT
grep -E "Grid Size|Run Time" out.txt >> $OUT
grep -E "^\[lsf\] <-" err.txt | grep -v _device >> $OUT
python3 - <<'PY' >> $OUT
import re
t = sum(float(m) for m in re.findall(r"^\[lsf\] <- lsf_(?!.*_device)\w+ \(([\d.]+) ms\)", open("err.txt").read(), re.M))
print(f"time inside the library: {t/1000:.2f} s")
PY
ls -la *.vti >> $OUT
