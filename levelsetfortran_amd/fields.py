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


def sphere_phi0(npts, centers=((0.0, 0.0, 0.0),), radius=1.0, lo=-1.5, hi=1.5, ranges=None):
    """Fortran-ordered (Nx,Ny,Nz) float64 field and dx.  Several centres -> union (min distance).
    ranges = ((i0,i1),(j0,j1),(k0,k1)) returns only that index block of the global field."""
    x, y, z, dx = grid_axes(npts, lo, hi)
    if ranges is not None:
        x, y, z = x[ranges[0][0]:ranges[0][1]], y[ranges[1][0]:ranges[1][1]], z[ranges[2][0]:ranges[2][1]]
    d = None
    for c in centers:
        r = np.sqrt((x[:, None, None] - c[0]) ** 2 + (y[None, :, None] - c[1]) ** 2 + (z[None, None, :] - c[2]) ** 2)
        dd = r - radius
        d = dd if d is None else np.minimum(d, dd)
    phi = d / np.sqrt(d * d + dx * dx)
    return np.asfortranarray(phi, dtype=np.float64), float(dx)


def two_sphere_phi0(npts, lo=-1.5, hi=1.5, ranges=None):
    return sphere_phi0(npts, centers=((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)), radius=0.5, lo=lo, hi=hi, ranges=ranges)


def reinit_step(dx: float, extent=(2.0, 2.0, 2.0), cfl: float = 0.1) -> float:
    """h = CFL * dx / |bbox diagonal| (set3d.f90:301-305); extent (2,2,2) is cube40's bounding box."""
    return cfl * dx / float(np.sqrt(extent[0] ** 2 + extent[1] ** 2 + extent[2] ** 2))


def two_sphere_phi0_device(npts, device, lo=-1.5, hi=1.5, ranges=None):
    """Same field as two_sphere_phi0, generated directly in HBM with torch (for grids whose numpy temporaries
    would not fit in host memory, e.g. 1024^3).  Returns (1-D float64 CUDA tensor, i fastest; dx).
    ranges = ((i0,i1),(j0,j1),(k0,k1)) returns only that index block of the global field."""
    import torch

    nxp, nyp, nzp = npts
    dx = (hi - lo) / (nxp - 1)
    rng = ranges if ranges is not None else ((0, nxp), (0, nyp), (0, nzp))
    ax = [lo + dx * torch.arange(r[0], r[1], dtype=torch.float64, device=device) for r in rng]
    d = None
    for c in ((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)):
        r2 = (ax[2][:, None, None] - c[2]) ** 2 + (ax[1][None, :, None] - c[1]) ** 2 + (ax[0][None, None, :] - c[0]) ** 2
        r2.sqrt_().sub_(0.5)
        d = r2 if d is None else torch.minimum(d, r2)
        del r2
    phi = d / torch.sqrt(d * d + dx * dx)
    return phi.reshape(-1), float(dx)
