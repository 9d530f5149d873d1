#!/usr/bin/env python3
"""BASELINE config 1 at its literal size (cube40.stl, dx = 2/42 -> nx = ny = nz = 63: a 64^3 grid) from the reference itself.

SURVEY.md section 8b: ceiling(2 / (2/42)) + 21 = 63, "on a rounding knife-edge" -- asserted below for the extents the
STL really has (REAL*4 promoted to fp64).  The reference's main program has dx hard-coded (set3d.f90:140), so, as for
config 2 (make_golden_c2.py): phi0 comes from the pinned oracle's restatement of set3d.f90:196-268 (bit-identical to the
reference's phi0 at the shipped size), the sweeps are run by the reference's OWN `reinit` (amdflang build in oracle/_ref,
called through ctypes) until its 1e-5 stop, narrowBand + the min/max flow to its 1e-7 stop by the pinned oracle.
~3 CPU-minutes, build container only.

  python tests/golden/make_golden_c1.py        -> tests/golden/cube40_64.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from make_golden import ref_reinit, sha  # noqa: E402

import oracle_lib  # noqa: E402
import stl_io  # noqa: E402

DX = 2.0 / 42.0

s = np.load(os.path.join(HERE, "surfaces.npz"))
X, E = s["cube40_surfX"].astype(np.float64), s["cube40_surfElem"]
n, xLo, mn, mx = stl_io.grid_from_surface(X, dx=DX, dd=10)
assert tuple(n) == (63, 63, 63), n  # 64^3 points
nx, ny, nz = n
phi0 = oracle_lib.phi0(nx, ny, nz, DX, xLo, mn, mx, X, E)
ext = mx - mn
h = 0.1 * (DX / np.sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]))  # set3d.f90:301-305
f, tr = ref_reinit(phi0, nx, ny, nz, 10000, DX, h)  # set3d.f90:298, 308: the reference's own cap; stops at RMS < 1e-5
assert f is not None and 100 < len(tr) < 10000
sweeps = len(tr) + 1  # the stop sweep prints the steady-state line instead of its RMS (subs.f90:915-918)
# the same run by the pinned oracle: same field, same count (the oracle is what the GPU tests compare against elsewhere)
g = phi0.copy(order="F")
_, n_or, tr_or = oracle_lib.reinit(g, nx, ny, nz, 10000, DX, h, tol=1.0e-5)
assert n_or == sweeps and np.array_equal(g, f) and np.array_equal(np.asarray(tr_or[: len(tr)]), tr), (n_or, sweeps)
# narrowBand + min/max flow (set3d.f90:360, 386-462) by the pinned oracle
nb, sb = oracle_lib.narrowband(nx, ny, nz, DX, f)
h1 = 0.01 * (DX / np.sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]))  # set3d.f90:390-392
mm = f.copy(order="F")
nb2, sb2 = nb.copy(order="F"), sb.copy(order="F")
_, it_mm, tr_mm = oracle_lib.minmax(mm, nb2, sb2, nx, ny, nz, 10000, DX, h1, tol=1.0e-7)
np.savez_compressed(os.path.join(HERE, "cube40_64.npz"), dx=DX, h=h, h1=h1, nx=nx, xLo=xLo, xMin=mn, xMax=mx, phi0=phi0, sweeps=sweeps, rms=tr,
                    rms_stop=tr_or[sweeps - 1], phi_re=f, phi_re_sha=sha(f), nb_count=int(nb.sum()), sb_count=int(sb.sum()), mm_iters=it_mm,
                    rms_mm=np.asarray(tr_mm), phi_mm_sha=sha(mm), phi_mm_sample=np.ascontiguousarray(mm[::3, ::3, ::3]),
                    nb_sha=sha(nb2), sb_sha=sha(sb2))
print("done: sweeps", sweeps, "first/last printed RMS", tr[0], tr[-1], "min/max iterations", it_mm)
