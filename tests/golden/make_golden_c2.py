#!/usr/bin/env python3
"""BASELINE config 2 fixture (cube40.stl, 256^3 fp64, reinit only) from the reference itself.

phi0 at 256^3 comes from the oracle's restatement of set3d.f90:196-268 (bit-identical to the reference's phi0 at
the shipped size, tests/test_oracle_golden.py::test_phi0_from_surfaces; the reference has dx hard-coded, so its
main program cannot produce this grid); the 8 sweeps (one cycle of the raster directions) are run by the
reference's own `reinit` (amdflang build in oracle/_ref, called through ctypes).  The field is 134 MB: only its
SHA-256, a strided sample and the printed RMS values are kept.  ~8 CPU-minutes, build container only.
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

DX = 2.0 / 233.5  # ceiling(2/dx) = 234 -> nx = 234 + 1 + 2*10 = 255 (256^3 points), away from the ceiling knife-edge
SWEEPS = 8

s = np.load(os.path.join(HERE, "surfaces.npz"))
X, E = s["cube40_surfX"].astype(np.float64), s["cube40_surfElem"]
n, xLo, mn, mx = stl_io.grid_from_surface(X, dx=DX, dd=10)
assert tuple(n) == (255, 255, 255), n
phi0 = oracle_lib.phi0(n[0], n[1], n[2], DX, xLo, mn, mx, X, E)
ext = mx - mn
h = 0.1 * (DX / np.sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]))  # set3d.f90:301-305
f, tr = ref_reinit(phi0, n[0], n[1], n[2], SWEEPS - 1, DX, h)
assert len(tr) == SWEEPS
np.savez_compressed(os.path.join(HERE, "cube40_256.npz"), dx=DX, h=h, nx=n[0], sweeps=SWEEPS, phi0_sha=sha(phi0),
                    phi0_sample=np.ascontiguousarray(phi0[::8, ::8, ::8]), sha=sha(f),
                    sample=np.ascontiguousarray(f[::8, ::8, ::8]), rms=tr)
print("done", tr)
