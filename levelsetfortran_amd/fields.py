"""Deterministic synthetic initial fields and step sizes for the BASELINE configurations.

Formulas from SURVEY.md section 8d (C4/C5): domain [-1.5,1.5]^3, dx = 3/(N-1),
phi0 = d / sqrt(d^2 + dx^2) with d the signed distance to one sphere (c=0, R=1) or to the union of
two spheres (c=(-+0.6,0,0), R=0.5); h = 0.1*dx/sqrt(12).  The smeared sign mirrors what the
reference's own initialisation produces (phiSign with gM=1, set3d.f90:260-264).  Pure numpy: these
are inputs, not part of the hot path.
"""
from __future__ import annotations

import numpy as np


def grid_axes(npts, lo=-1.5, hi=1.5):
    """npts = (Nx,Ny,Nz) points; uniform spacing dx taken from the x axis."""
    nxp, nyp, nzp = npts
    dx = (hi - lo) / (nxp - 1)
    return lo + dx * np.arange(nxp), lo + dx * np.arange(nyp), lo + dx * np.arange(nzp), dx


def sphere_phi0(npts, centers=((0.0, 0.0, 0.0),), radius=1.0, lo=-1.5, hi=1.5):
    """Fortran-ordered (Nx,Ny,Nz) float64 field and dx.  Several centres -> union (min distance)."""
    x, y, z, dx = grid_axes(npts, lo, hi)
    d = None
    for c in centers:
        r = np.sqrt((x[:, None, None] - c[0]) ** 2 + (y[None, :, None] - c[1]) ** 2 + (z[None, None, :] - c[2]) ** 2)
        dd = r - radius
        d = dd if d is None else np.minimum(d, dd)
    phi = d / np.sqrt(d * d + dx * dx)
    return np.asfortranarray(phi, dtype=np.float64), float(dx)


def two_sphere_phi0(npts, lo=-1.5, hi=1.5):
    return sphere_phi0(npts, centers=((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)), radius=0.5, lo=lo, hi=hi)


def reinit_step(dx: float, extent=(2.0, 2.0, 2.0), cfl: float = 0.1) -> float:
    """h = CFL * dx / |bbox diagonal| (set3d.f90:301-305); extent (2,2,2) is cube40's bounding box."""
    return cfl * dx / float(np.sqrt(extent[0] ** 2 + extent[1] ** 2 + extent[2] ** 2))
