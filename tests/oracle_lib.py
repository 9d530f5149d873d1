"""ctypes wrapper of oracle/liblsf_oracle.so -- test infrastructure (the checker).

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int, c_int32

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# LSF_ORACLE_OMP=1: the OpenMP build of the same source (oracle/Makefile `omp`; long fixtures only)
ORACLE_SO = os.path.join(ORACLE_DIR, "liblsf_oracle_omp.so" if os.environ.get("LSF_ORACLE_OMP") else "liblsf_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_subs.so")

GS_LEX, GS_HYPER, JACOBI = 0, 1, 2
BC_CLOSED, BC_LITERAL = 0, 1

_dp, _ip = POINTER(c_double), POINTER(c_int32)
_lib = None


def build():
    """(Re)build the C restatement with gcc; cheap, idempotent."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
            if os.environ.get("LSF_ORACLE_OMP"):
                subprocess.run(["make", "-s", "-C", ORACLE_DIR, "omp"], check=True)
        L = ctypes.CDLL(ORACLE_SO)
        L.lsf_oracle_weno.restype = c_double
        L.lsf_oracle_weno.argtypes = [c_int] * 6 + [c_double, _dp]
        L.lsf_oracle_phisign.restype = c_double
        L.lsf_oracle_phisign.argtypes = [c_double] * 3
        L.lsf_oracle_bc.restype = None
        L.lsf_oracle_bc.argtypes = [_dp, c_int, c_int, c_int, c_double, c_int]
        L.lsf_oracle_reinit.restype = c_int
        L.lsf_oracle_reinit.argtypes = [_dp, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_int, c_int,
                                        c_int, POINTER(c_int), _dp, c_int]
        L.lsf_oracle_narrowband.restype = None
        L.lsf_oracle_narrowband.argtypes = [c_int, c_int, c_int, c_double, _dp, _ip, _ip]
        L.lsf_oracle_minmax.restype = c_int
        L.lsf_oracle_minmax.argtypes = [_dp, _ip, _ip, c_int, c_int, c_int, c_int, c_double, c_double, c_double,
                                        c_int, POINTER(c_int), _dp, c_int]
        L.lsf_oracle_phi0.restype = None
        L.lsf_oracle_phi0.argtypes = [_dp, c_int, c_int, c_int, c_double, _dp, _dp, _dp, _dp, c_int, _ip, c_int]
        _lib = L
    return _lib


def _d(a):
    assert a.dtype == np.float64 and (a.flags.f_contiguous or a.ndim == 1)
    return a.ctypes.data_as(_dp)


def _i(a):
    assert a.dtype == np.int32 and (a.flags.f_contiguous or a.ndim == 1)
    return a.ctypes.data_as(_ip)


def reinit(phi, nx, ny, nz, iter, dx, h, tol=1e-5, order=GS_LEX, bc=BC_CLOSED, first_raster=0):
    """In place on phi (Fortran-ordered).  Returns (rc, sweeps, rms_trace)."""
    done = c_int(0)
    tr = np.zeros(iter + 1)
    rc = lib().lsf_oracle_reinit(_d(phi), nx, ny, nz, iter, dx, h, tol, order, bc, first_raster, ctypes.byref(done),
                                 _d(tr), iter + 1)
    return rc, done.value, tr[: done.value]


def narrowband(nx, ny, nz, dx, phi):
    nb = np.zeros(phi.shape, dtype=np.int32, order="F")
    sb = np.zeros(phi.shape, dtype=np.int32, order="F")
    lib().lsf_oracle_narrowband(nx, ny, nz, dx, _d(phi), _i(nb), _i(sb))
    return nb, sb


def minmax(phi, nb, sb, nx, ny, nz, iter, dx, h1, tol=1e-7, order=GS_LEX):
    done = c_int(0)
    tr = np.zeros(max(iter, 1))
    rc = lib().lsf_oracle_minmax(_d(phi), _i(nb), _i(sb), nx, ny, nz, iter, dx, h1, tol, order, ctypes.byref(done),
                                 _d(tr), max(iter, 1))
    return rc, done.value, tr[: done.value]


def bc(phi, nx, ny, nz, dx, kind=BC_CLOSED):
    lib().lsf_oracle_bc(_d(phi), nx, ny, nz, dx, kind)


def weno(i, j, k, nx, ny, nz, dx, phi):
    return lib().lsf_oracle_weno(i, j, k, nx, ny, nz, dx, _d(phi))


def phi0(nx, ny, nz, dx, xLo, minX, maxX, surfX, surfElem):
    phi = np.ones((nx + 1, ny + 1, nz + 1), order="F")
    sX = np.asfortranarray(surfX, dtype=np.float64)
    sE = np.asfortranarray(surfElem, dtype=np.int32)
    a3 = lambda v: np.ascontiguousarray(v, dtype=np.float64)
    lo, mn, mx = a3(xLo), a3(minX), a3(maxX)
    lib().lsf_oracle_phi0(_d(phi), nx, ny, nz, dx, _d(lo), _d(mn), _d(mx), _d(sX), sX.shape[0], _i(sE), sE.shape[0])
    return phi


def advect(phi, phiSB, nx, ny, nz, dx, xLo, surfX, iters=1000):
    """set3d.f90:464-501: order-8 gradients on the stencil band, then the node advection.  Returns surfXX."""
    L = lib()
    vp = ctypes.c_void_p
    L.lsf_oracle_firstderiv8.restype = None
    L.lsf_oracle_firstderiv8.argtypes = [vp, vp, c_int, c_int, c_int, c_double, vp]
    L.lsf_oracle_advect.restype = None
    L.lsf_oracle_advect.argtypes = [vp, vp, c_int, c_int, c_int, c_double, vp, vp, c_int, c_int]
    grad = np.zeros(phi.shape + (3,), order="F")
    sb = np.asfortranarray(phiSB, dtype=np.int32)
    lo = np.ascontiguousarray(xLo, dtype=np.float64)
    XX = np.array(surfX, dtype=np.float64, order="F", copy=True)
    L.lsf_oracle_firstderiv8(phi.ctypes.data, sb.ctypes.data, nx, ny, nz, dx, grad.ctypes.data)
    L.lsf_oracle_advect(phi.ctypes.data, grad.ctypes.data, nx, ny, nz, dx, lo.ctypes.data, XX.ctypes.data, XX.shape[0], iters)
    return XX
