"""The cost of ONE hand-off between dependent tiles of the dataflow launch, in isolation: grids that are one tile wide in two of the three
tile directions, so that a sweep is a single chain of tiles; kernel time per sweep / tiles per sweep.  LSF_GS_MARCH=x keeps the kernel's axes
the caller's.  python3 profiles/micro/hop_probe.py [sweeps=16]"""
import ctypes, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import levelsetfortran_amd as L
from levelsetfortran_amd import _lib, fields
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = _lib.load()
os.environ["LSF_GS_MARCH"] = "x"
for arith in ("fast", "strict"):
    for name, npts in (("x chain", (2050, 11, 9)), ("y chain", (17, 1282, 9)), ("z chain", (17, 11, 1026)), ("x chain, c1x4", (2050, 17, 17))):
        if "c1x4" in name:
            os.environ["LSF_GS_SKEW_W"] = "c1x4"
        else:
            os.environ.pop("LSF_GS_SKEW_W", None)
        phi_h, dx = fields.two_sphere_phi0(npts)
        n = tuple(v - 1 for v in npts); h = fields.reinit_step(dx)
        phi = torch.from_numpy(np.ascontiguousarray(phi_h.ravel(order="F"))).cuda()
        L.reinit(phi.clone(), None, None, *n, K - 1, dx, h, tol=0.0, order="gs", arith=arith)
        lib.lsf_profile(1)
        L.reinit(phi, None, None, *n, K - 1, dx, h, tol=0.0, order="gs", arith=arith)
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(); nl, s = ctypes.c_longlong(), ctypes.c_int()
        lib.lsf_profile_get(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(nl), ctypes.byref(s))
        lib.lsf_profile(0)
        by, wy, wz = (16, 1, 4) if "c1x4" in name else (5, 2, 2)
        nTj, nTk = -(-(n[1] - 1) // (by * wy)), -(-(n[2] - 1) // (4 * wz))
        nM = (n[0] - 2 + by * wy * nTj - 1 + 4 * wz * nTk - 1) // 16 + 1
        planes = nM + nTj + nTk - 2
        print(f"{arith:6s} {name:14s} {npts}: tiles {nTj} x {nTk} across, {planes} hyperplanes: kernel {a.value / K:.3f} ms per sweep, K = {K}", flush=True)
