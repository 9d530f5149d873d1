import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("LSF_GS_TIMEOUT_TICKS", "100000000")  # 1 s
import numpy as np, torch
import levelsetfortran_amd as L
from levelsetfortran_amd import fields
for npts, nd, arith, iters in [((71, 83, 97), 2, "strict", 12), ((71, 83, 97), 3, "strict", 12), ((96, 80, 72), 4, "fast", 20), ((71, 83, 97), 4, "strict", 12), ((128, 128, 128), 2, "fast", 70)]:
    phi0, dx = fields.two_sphere_phi0(npts)
    n = tuple(v - 1 for v in npts); h = fields.reinit_step(dx)
    want = phi0.copy(order="F")
    r1 = L.reinit(want, None, None, *n, iters, dx, h, tol=0.0, order="gs", arith=arith)
    got = phi0.copy(order="F")
    t = time.time()
    try:
        r = L.reinit_multi(got, *n, iters, dx, h, [0] * nd, tol=0.0, arith=arith, order="gs")
    except Exception as e:
        print(npts, nd, arith, "FAILED", e, flush=True); continue
    print(npts, nd, arith, "count", r.count, r1.count, "equal", np.array_equal(got, want), "maxdiff", float(np.abs(got - want).max()),
          "rms equal", r.rms == r1.rms, "%.2fs" % (time.time() - t), flush=True)
