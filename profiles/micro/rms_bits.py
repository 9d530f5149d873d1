"""RMS traces of the exact ordering, printed as hex, for a set of grids / shapes / arithmetics: run once per library build (LSF_LIB_PATH) and diff the
outputs -- a change of the reduction's instructions must not change a bit.  python3 profiles/micro/rms_bits.py"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import levelsetfortran_amd as L
from levelsetfortran_amd import fields
for npts in ((40, 33, 27), (70, 21, 45), (96, 96, 96), (130, 75, 101)):
    phi0, dx = fields.two_sphere_phi0(npts); n = tuple(v - 1 for v in npts); h = fields.reinit_step(dx)
    for arith in ("fast", "strict"):
        for shape in (None, "c1x4", "1x1", "4x2", "c1x2"):
            os.environ.pop("LSF_GS_SKEW_W", None)
            if shape:
                os.environ["LSF_GS_SKEW_W"] = shape
            a = phi0.copy(order="F")
            r = L.reinit(a, None, None, *n, 11, dx, h, tol=0.0, order="gs", arith=arith)
            print(npts, arith, shape, " ".join(float(x).hex() for x in r.rms))
