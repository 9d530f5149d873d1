"""Host-side mirror of the reference's interface for the hot path.

Same names, argument meaning and error behaviour as the Fortran module procedures and the loop
they replace (citations are into /root/reference):

  reinit(phi, gradPhi, gradPhiMag, nx, ny, nz, iter, dx, h)     subs.f90:717-931
  narrowBand(nx, ny, nz, dx, phi, phiNB, phiSB)                 subs.f90:178-207
  minmaxFlow(phi, phiNB, phiSB, nx, ny, nz, iter, dx, h1)       set3d.f90:394-462 (hoisted)

Fields are updated IN PLACE like the INTENT(INOUT) dummies of the reference.  A field is either
  * a numpy float64 array, Fortran-ordered with shape (nx+1, ny+1, nz+1) (or 1-D of that size):
    the host seam -- the library copies it to HBM and back (lsf_reinit / lsf_minmax), or
  * a torch CUDA float64 tensor, C-contiguous with shape (nz+1, ny+1, nx+1) (or 1-D): the same
    bytes already resident in HBM (lsf_*_device); nothing crosses PCIe.

All arithmetic happens in liblsf_hip.so (hand-written HIP for gfx950).  This module contains no
numerical code and no fallback.
"""
from __future__ import annotations

import ctypes
import sys
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import LSF_ARITH_FAST, LSF_ARITH_STRICT, LSF_ORDER_GS, LSF_ORDER_JACOBI, LsfError, LsfNaNError

__all__ = ["reinit", "narrowBand", "minmaxFlow", "phi0Init", "advectNodes", "SweepReport", "mode_word", "LsfError", "LsfNaNError", "peer_selftest"]

REINIT_TOL = 1.0e-5  # subs.f90:915
MINMAX_TOL = 1.0e-7  # set3d.f90:448


def mode_word(order: str = "gs", arith: str = "fast") -> int:
    """Build the `mode` argument of include/lsf.h from readable names."""
    o = {"gs": LSF_ORDER_GS, "jacobi": LSF_ORDER_JACOBI}[order]
    a = {"fast": LSF_ARITH_FAST, "strict": LSF_ARITH_STRICT}[arith]
    return o | a


@dataclass
class SweepReport:
    """What the reference prints while iterating (subs.f90:916,923 / set3d.f90:449,456)."""

    count: int  # sweeps / iterations executed
    rms: List[float] = field(default_factory=list)  # RMS change after each of them
    converged: bool = False  # last RMS < tol -> the "steady state" line

    def lines(self, first_index: int, steady_msg: str) -> List[str]:
        """The stdout lines of the reference for this run (list-directed formatting aside)."""
        out = []
        n_print = self.count - 1 if self.converged else self.count
        for s in range(n_print):
            out.append(f"  Iteration:  {s + first_index}   RMS Error:  {self.rms[s]!r}")
        if self.converged:
            out.append(steady_msg)
        return out


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _npoints(nx: int, ny: int, nz: int) -> int:
    return (nx + 1) * (ny + 1) * (nz + 1)


def _host_ptr(a: np.ndarray, dtype, nx, ny, nz, name: str) -> int:
    if not isinstance(a, np.ndarray) or a.dtype != dtype:
        raise TypeError(f"{name} must be a numpy array of {np.dtype(dtype).name}")
    if a.ndim == 3:
        if a.shape != (nx + 1, ny + 1, nz + 1) or not a.flags.f_contiguous:
            raise ValueError(f"{name} must be Fortran-ordered with shape (nx+1, ny+1, nz+1) = {(nx+1, ny+1, nz+1)}")
    elif a.ndim != 1 or a.size != _npoints(nx, ny, nz) or not a.flags.c_contiguous:
        raise ValueError(f"{name} must have (nx+1)(ny+1)(nz+1) contiguous elements")
    if not a.flags.writeable:
        raise ValueError(f"{name} must be writeable (it is INTENT(INOUT) in the reference)")
    return a.ctypes.data


def _dev_ptr(t, torch_dtype, nx, ny, nz, name: str) -> int:
    import torch

    if not t.is_cuda or t.dtype != torch_dtype:
        raise TypeError(f"{name} must be a CUDA tensor of {torch_dtype}")
    if t.dim() == 3:
        if tuple(t.shape) != (nz + 1, ny + 1, nx + 1):
            raise ValueError(f"{name} must have shape (nz+1, ny+1, nx+1) (i is the unit-stride axis)")
    elif t.dim() != 1 or t.numel() != _npoints(nx, ny, nz):
        raise ValueError(f"{name} must have (nx+1)(ny+1)(nz+1) elements")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t.data_ptr()


def _stream_and_device(t):
    import torch

    lib = _lib.load()
    _lib.check(lib.lsf_set_device(t.device.index or 0))
    return torch.cuda.current_stream(t.device).cuda_stream


def reinit(phi, gradPhi=None, gradPhiMag=None, nx: int = 0, ny: int = 0, nz: int = 0, iter: int = 0,
           dx: float = 0.0, h: float = 0.0, *, tol: float = REINIT_TOL, order: str = "gs", arith: str = "fast",
           first_raster: int = 0, phiS=None, echo: bool = False) -> SweepReport:
    """SUBROUTINE reinit (subs.f90:717-931) on the GPU; `phi` is updated in place.

    gradPhi / gradPhiMag are accepted for signature parity and left untouched: the reference never
    reads what reinit stores there (set3d.f90:372-375 zeroes them).
    Runs at most iter+1 sweeps (subs.f90:735).  Raises LsfNaNError where the reference STOPs.
    echo=True prints the reference's per-sweep lines.
    A float32 field (numpy or torch) selects the single-precision path of BASELINE configuration 5, which
    exists for order="jacobi", arith="fast" only (the reference is fp64; see include/lsf.h).
    """
    lib = _lib.load()
    cap = int(iter) + 1
    trace = np.zeros(max(cap, 1), dtype=np.float64)
    done = ctypes.c_int(0)
    mode = mode_word(order, arith)
    if _is_torch(phi):
        import torch

        f32 = phi.dtype == torch.float32
        tdt = torch.float32 if f32 else torch.float64
        p = _dev_ptr(phi, tdt, nx, ny, nz, "phi")
        ps = _dev_ptr(phiS, tdt, nx, ny, nz, "phiS") if phiS is not None else None
        st = _stream_and_device(phi)
        if f32:  # single precision: Jacobi ordering, FAST arithmetic only (include/lsf.h)
            if first_raster != 0:
                raise ValueError("first_raster has no meaning for the Jacobi ordering")
            rc = lib.lsf_reinit_f32_device(p, ps, nx, ny, nz, int(iter), float(dx), float(h), float(tol), mode,
                                           ctypes.byref(done), trace.ctypes.data, cap, st)
        else:
            rc = lib.lsf_reinit_device(p, ps, nx, ny, nz, int(iter), float(dx), float(h), float(tol), mode,
                                       int(first_raster), ctypes.byref(done), trace.ctypes.data, cap, st)
    else:
        if first_raster != 0 or phiS is not None:
            raise ValueError("first_raster / phiS are only available on the device seam")
        if isinstance(phi, np.ndarray) and phi.dtype == np.float32:
            p = _host_ptr(phi, np.float32, nx, ny, nz, "phi")
            rc = lib.lsf_reinit_f32(p, nx, ny, nz, int(iter), float(dx), float(h), float(tol), mode,
                                    ctypes.byref(done), trace.ctypes.data, cap)
        else:
            p = _host_ptr(phi, np.float64, nx, ny, nz, "phi")
            rc = lib.lsf_reinit(p, nx, ny, nz, int(iter), float(dx), float(h), float(tol), mode, ctypes.byref(done),
                                trace.ctypes.data, cap)
    n = done.value
    rep = SweepReport(n, [float(v) for v in trace[:n]], bool(n and trace[n - 1] < tol))
    if echo:
        for ln in rep.lines(0, "  Distance function time integration has reached steady state "):
            print(ln)
        print()
        sys.stdout.flush()
    _lib.check(rc)
    return rep


def narrowBand(nx: int, ny: int, nz: int, dx: float, phi, phiNB, phiSB) -> None:
    """SUBROUTINE narrowBand (subs.f90:178-207): phiNB = |phi| < 4.1 dx, phiSB = |phi| < 8.1 dx."""
    lib = _lib.load()
    if _is_torch(phi):
        import torch

        st = _stream_and_device(phi)
        rc = lib.lsf_narrowband_device(_dev_ptr(phi, torch.float64, nx, ny, nz, "phi"),
                                       _dev_ptr(phiNB, torch.int32, nx, ny, nz, "phiNB"),
                                       _dev_ptr(phiSB, torch.int32, nx, ny, nz, "phiSB"), nx, ny, nz, float(dx), st)
    else:
        rc = lib.lsf_narrowband(_host_ptr(phi, np.float64, nx, ny, nz, "phi"),
                                _host_ptr(phiNB, np.int32, nx, ny, nz, "phiNB"),
                                _host_ptr(phiSB, np.int32, nx, ny, nz, "phiSB"), nx, ny, nz, float(dx))
    _lib.check(rc)


def minmaxFlow(phi, phiNB, phiSB, nx: int, ny: int, nz: int, iter: int, dx: float, h1: float, *,
               tol: float = MINMAX_TOL, order: str = "gs", echo: bool = False) -> SweepReport:
    """The min/max-flow loop of the main program (set3d.f90:394-462) as one call.

    phi, phiNB, phiSB are updated in place; the masks come back as the host would hold them after
    the loop (refreshed only on the non-exit path, set3d.f90:448-460).
    """
    lib = _lib.load()
    cap = max(int(iter), 1)
    trace = np.zeros(cap, dtype=np.float64)
    done = ctypes.c_int(0)
    mode = mode_word(order, "strict")  # min/max has a single (exact) arithmetic
    if _is_torch(phi):
        import torch

        st = _stream_and_device(phi)
        rc = lib.lsf_minmax_device(_dev_ptr(phi, torch.float64, nx, ny, nz, "phi"),
                                   _dev_ptr(phiNB, torch.int32, nx, ny, nz, "phiNB"),
                                   _dev_ptr(phiSB, torch.int32, nx, ny, nz, "phiSB"), nx, ny, nz, int(iter),
                                   float(dx), float(h1), float(tol), mode, ctypes.byref(done), trace.ctypes.data,
                                   cap, st)
    else:
        rc = lib.lsf_minmax(_host_ptr(phi, np.float64, nx, ny, nz, "phi"),
                            _host_ptr(phiNB, np.int32, nx, ny, nz, "phiNB"),
                            _host_ptr(phiSB, np.int32, nx, ny, nz, "phiSB"), nx, ny, nz, int(iter), float(dx),
                            float(h1), float(tol), mode, ctypes.byref(done), trace.ctypes.data, cap)
    n = done.value
    rep = SweepReport(n, [float(v) for v in trace[:n]], bool(n and trace[n - 1] < tol))
    if echo:
        for ln in rep.lines(1, "  Min/max time integration has reached steady state "):
            print(ln)
        print()
        sys.stdout.flush()
    _lib.check(rc)
    return rep


def phi0Init(phi, nx: int, ny: int, nz: int, dx: float, xLo, minX, maxX, surfX, surfElem) -> None:
    """Inside/outside initialisation of the main program (set3d.f90:196-268) as one call.

    phi (numpy F-ordered or torch CUDA, see module docstring) receives sgn in (-1,1) within 3 cells of the
    surface bounding box and 1.0 elsewhere.  surfX: (nSurfNode,3) float64, surfElem: (nSurfElem,3) int32,
    1-based, as stlRead returns them (subs.f90:17-121); xLo/minX/maxX as the host computes them
    (set3d.f90:94-157).  Bit-identical to the reference.
    """
    lib = _lib.load()
    sX = np.asfortranarray(surfX, dtype=np.float64)
    sE = np.asfortranarray(surfElem, dtype=np.int32)
    lo, mn, mx = (np.ascontiguousarray(v, dtype=np.float64) for v in (xLo, minX, maxX))
    args = (nx, ny, nz, float(dx), lo.ctypes.data, mn.ctypes.data, mx.ctypes.data, sX.ctypes.data, sX.shape[0],
            sE.ctypes.data, sE.shape[0])
    if _is_torch(phi):
        import torch

        st = _stream_and_device(phi)
        rc = lib.lsf_phi0_device(_dev_ptr(phi, torch.float64, nx, ny, nz, "phi"), *args, st)
    else:
        rc = lib.lsf_phi0(_host_ptr(phi, np.float64, nx, ny, nz, "phi"), *args)
    _lib.check(rc)


def advectNodes(phi, phiSB, nx: int, ny: int, nz: int, dx: float, xLo, surfXX, iter: int = 1000) -> None:
    """Order-8 gradients on the stencil band + surface-node advection, set3d.f90:470-501, as one call.

    surfXX: (nSurfNode,3) float64 numpy array (Fortran-ordered), the nodes on entry and the advected nodes on
    return (this is what the host writes to the .s3d file, set3d.f90:606-608).  phi / phiSB as for minmaxFlow.
    Bit-identical to the reference (including subs.f90:346's repeated j+1 neighbour).
    """
    lib = _lib.load()
    if not (isinstance(surfXX, np.ndarray) and surfXX.dtype == np.float64 and surfXX.ndim == 2
            and surfXX.shape[1] == 3 and surfXX.flags.f_contiguous and surfXX.flags.writeable):
        raise ValueError("surfXX must be a writeable Fortran-ordered float64 array of shape (nSurfNode, 3)")
    lo = np.ascontiguousarray(xLo, dtype=np.float64)
    if _is_torch(phi):
        import torch

        st = _stream_and_device(phi)
        rc = lib.lsf_advect_nodes_device(_dev_ptr(phi, torch.float64, nx, ny, nz, "phi"),
                                         _dev_ptr(phiSB, torch.int32, nx, ny, nz, "phiSB"), nx, ny, nz, float(dx),
                                         lo.ctypes.data, surfXX.ctypes.data, surfXX.shape[0], int(iter), st)
    else:
        rc = lib.lsf_advect_nodes(_host_ptr(phi, np.float64, nx, ny, nz, "phi"),
                                  _host_ptr(phiSB, np.int32, nx, ny, nz, "phiSB"), nx, ny, nz, float(dx),
                                  lo.ctypes.data, surfXX.ctypes.data, surfXX.shape[0], int(iter))
    _lib.check(rc)


TRANSPORTS = {"peer": _lib.LSF_TRANSPORT_PEER, "rccl": _lib.LSF_TRANSPORT_RCCL, "mock": _lib.LSF_TRANSPORT_MOCK}


def reinit_multi(phi, nx: int, ny: int, nz: int, iter: int, dx: float, h: float, devices, *, dims=None,
                 tol: float = REINIT_TOL, arith: str = "fast", check_every: int = 8, transport: str = "peer",
                 order: str = "jacobi") -> SweepReport:
    """reinit on every device of `devices` from ONE process (include/lsf.h: lsf_reinit_multi; the call site
    set3d.f90:308 for a host that wants all the GPUs of the node).  phi: Fortran-ordered numpy array, float64 or
    float32, updated in place.  order="jacobi": blocks with ghost layers, bit-identical to reinit(..., order="jacobi")
    on one device.  order="gs" (float64): the reference's in-place ordering (subs.f90:743-852) over z slabs, one per
    device, bit-identical to reinit(..., order="gs") and so, with arith="strict", to the reference; dims, check_every
    and transport do not apply.  A device may be named more than once (several blocks / slabs share it).
    check_every: sweeps between two looks at the RMS (the stop sweep, the field and the trace do not depend on it);
    transport: "peer" (peer copies), "rccl" (ncclSend / ncclRecv, a distinct device per block) or "mock" (test aid)."""
    lib = _lib.load()
    if transport not in TRANSPORTS:
        raise ValueError(f"transport must be one of {sorted(TRANSPORTS)}, not {transport!r}")
    if order not in ("jacobi", "gs"):
        raise ValueError(f"order must be 'jacobi' or 'gs', not {order!r}")
    if order == "gs" and (check_every != 8 or transport != "peer"):
        raise ValueError("order='gs' (the reference's ordering over z slabs) takes no check_every or transport")
    old_ce, old_tr = ctypes.c_int(8), ctypes.c_int(_lib.LSF_TRANSPORT_PEER)
    lib.lsf_multi_defaults_get(ctypes.byref(old_ce), ctypes.byref(old_tr))  # this thread's: put back afterwards
    _lib.check(lib.lsf_multi_defaults(int(check_every), TRANSPORTS[transport]))
    cap = int(iter) + 1
    trace = np.zeros(max(cap, 1), dtype=np.float64)
    done = ctypes.c_int(0)
    mode = mode_word(order, arith)
    devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
    dm = (ctypes.c_int * 3)(*[int(d) for d in dims]) if dims is not None else None
    f32 = isinstance(phi, np.ndarray) and phi.dtype == np.float32
    p = _host_ptr(phi, np.float32 if f32 else np.float64, nx, ny, nz, "phi")
    fn = lib.lsf_reinit_multi_f32 if f32 else lib.lsf_reinit_multi
    try:
        rc = fn(p, nx, ny, nz, int(iter), float(dx), float(h), float(tol), mode, devs, len(devices), dm, ctypes.byref(done),
                trace.ctypes.data, cap)
    finally:
        lib.lsf_multi_defaults(old_ce.value, old_tr.value)
    n = done.value
    rep = SweepReport(n, [float(v) for v in trace[:n]], bool(n and trace[n - 1] < tol))
    _lib.check(rc)
    return rep


def peer_selftest(dev_a: int, dev_b: int) -> int:
    """include/lsf.h: lsf_peer_selftest -- the litmus test of the device-to-device hand-offs the exact ordering across z slabs
    relies on.  Returns 0, or raises LsfError whose message names the violated assumption."""
    lib = _lib.load()
    bad = ctypes.c_int(0)
    _lib.check(lib.lsf_peer_selftest(int(dev_a), int(dev_b), ctypes.byref(bad)))
    return bad.value
