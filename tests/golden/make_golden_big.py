#!/usr/bin/env python3
"""BASELINE-size fixtures from the reference itself: `reinit` (subs.f90:717, amdflang build in
oracle/_ref) called through ctypes on the synthetic two-sphere phi0 (levelsetfortran_amd.fields) at
256^3 (8 sweeps: one full cycle of the raster directions) and 512^3 (2 sweeps).  The fields are too
large to commit (134 MB / 1.07 GB): the fixture keeps the SHA-256 of the full field, a strided
sample and the printed RMS values.  ~12 CPU-minutes; run once in the build container:

    python tests/golden/make_golden_big.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import ref_reinit, sha  # noqa: E402

from levelsetfortran_amd import fields  # noqa: E402

out = {}
for N, sweeps in ((256, 8), (512, 2)):
    phi0, dx = fields.two_sphere_phi0((N, N, N))
    h = fields.reinit_step(dx)
    f, tr = ref_reinit(phi0, N - 1, N - 1, N - 1, sweeps - 1, dx, h)
    assert len(tr) == sweeps
    out[f"n{N}_sweeps"] = sweeps
    out[f"n{N}_dx"] = dx
    out[f"n{N}_h"] = h
    out[f"n{N}_sha"] = sha(f)
    out[f"n{N}_sample"] = np.ascontiguousarray(f[::16, ::16, ::16])
    out[f"n{N}_rms"] = tr
    print(N, sweeps, tr, flush=True)
    del f, phi0
np.savez_compressed(os.path.join(HERE, "synth_big.npz"), **out)
