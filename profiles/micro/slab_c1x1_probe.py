"""one-lane single-wavefront tiles (c1x1) across slabs of one device: the soak case that timed out (154 x 186 x 239 points, FAST, 12 sweeps,
2 slabs).  python3 profiles/micro/slab_c1x1_probe.py [arith=fast] [slabs=2] [shape=c1x1]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import levelsetfortran_amd as L
from levelsetfortran_amd import fields
arith = sys.argv[1] if len(sys.argv) > 1 else "fast"
slabs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
os.environ["LSF_GS_SKEW_W"] = sys.argv[3] if len(sys.argv) > 3 else "c1x1"
npts = (154, 186, 239); sweeps = 12
phi0, dx = fields.two_sphere_phi0(npts)
n = tuple(v - 1 for v in npts); h = fields.reinit_step(dx)
want = phi0.copy(order="F")
r1 = L.reinit(want, None, None, *n, sweeps - 1, dx, h, tol=0.0, order="gs", arith=arith)
got = phi0.copy(order="F")
t0 = time.time()
try:
    r = L.reinit_multi(got, *n, sweeps - 1, dx, h, [0] * slabs, tol=0.0, arith=arith, order="gs")
    print(arith, slabs, os.environ["LSF_GS_SKEW_W"], "equal" if np.array_equal(got, want) and r.rms == r1.rms else "DIFFERS", f"{time.time() - t0:.2f} s")
except Exception as e:  # noqa: BLE001
    print(arith, slabs, os.environ["LSF_GS_SKEW_W"], "ERROR", repr(e), f"{time.time() - t0:.2f} s")
