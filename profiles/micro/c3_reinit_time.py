"""Time of the 128-sweep exact-ordering reinit on the twoCube10 512^3 field of BASELINE configuration 3 (device seam),
next to the same call on the bench's two-sphere field: is the drop-in's lsf_reinit time a property of the data?"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import levelsetfortran_amd as lsf
import stl_io
from levelsetfortran_amd import fields
g = np.load('tests/golden/twocube10_512cubed_s128.npz')
s = np.load('tests/golden/surfaces.npz')
X, E = s["twocube10_surfX"].astype(np.float64), s["twocube10_surfElem"]
dx = float(g["dx"]); h = float(g["h"])
n, xLo, mn, mx = stl_io.grid_from_surface_pads(X, dx, g["pad_lo"], g["pad_hi"])
nx, ny, nz = n
phi = torch.ones((nx + 1) * (ny + 1) * (nz + 1), dtype=torch.float64, device="cuda")
lsf.phi0Init(phi, nx, ny, nz, dx, xLo, mn, mx, X, E)
two, dx2 = fields.two_sphere_phi0_device((512, 512, 512), torch.device('cuda', 0))
for name, f, d, hh in (("two-sphere", two, dx2, fields.reinit_step(dx2)), ("twoCube10", phi, dx, h), ("two-sphere", two.clone(), dx2, fields.reinit_step(dx2))):
    for arith in ("fast", "strict"):
        a = f.clone()
        lsf.reinit(a, None, None, 511, 511, 511, 3, d, hh, tol=0.0, arith=arith); torch.cuda.synchronize()
        a = f.clone(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        rep = lsf.reinit(a, None, None, 511, 511, 511, 127, d, hh, tol=1e-5, arith=arith); torch.cuda.synchronize()
        print(f"{name} {arith}: 128 sweeps {1e3 * (time.perf_counter() - t0):.1f} ms, last RMS {rep.rms[-1]:.3e}", flush=True)
