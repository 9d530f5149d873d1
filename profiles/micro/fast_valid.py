"""Up to which sweep does the FAST arithmetic stay inside north_star's 1e-10 RMS of the reference's field?

STRICT on the GPU IS the reference's field (SHA-equal on every fixture, tests/test_gpu_config3.py), so the two
arithmetics are marched side by side in chunks (exact ordering, the sweep counter carried through `first_raster`, the sign
field fixed at phi0 like subs.f90:731) and compared after every chunk.  Run on the GPU box:

    python3 profiles/micro/fast_valid.py > profiles/r03_fast_valid.json

bench.py copies the result into its line as "fast_valid_sweeps".
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import levelsetfortran_amd as lsf  # noqa: E402
import stl_io  # noqa: E402
from levelsetfortran_amd import fields  # noqa: E402

TOL = 1.0e-10


def march(name, phi0, nx, ny, nz, dx, h, total, chunk, tol_stop):
    a, b = phi0.clone(), phi0.clone()
    done, first_bad, worst, stopped = 0, None, 0.0, None
    trail = []
    while done < total:
        k = min(chunk, total - done)
        ra = lsf.reinit(a, None, None, nx, ny, nz, k - 1, dx, h, tol=tol_stop, arith="strict", first_raster=done % 8, phiS=phi0)
        rb = lsf.reinit(b, None, None, nx, ny, nz, k - 1, dx, h, tol=tol_stop, arith="fast", first_raster=done % 8, phiS=phi0)
        assert ra.count == rb.count, (name, done, ra.count, rb.count)  # the same stop sweep
        done += ra.count
        d = b - a
        rms = float(torch.sqrt(torch.mean(d * d)))
        worst = max(worst, rms)
        trail.append((done, rms))
        if rms > TOL and first_bad is None:
            first_bad = done
        if ra.count < k:  # the reference's stop test fired (subs.f90:915)
            stopped = done
            break
    last_ok = max([s for s, r in trail if r <= TOL and (first_bad is None or s < first_bad)], default=0)
    signs = int(((a < 0) != (b < 0)).sum())
    return {"sweeps_run": done, "stopped_at": stopped, "fast_within_1e-10_rms_through_sweep": last_ok,
            "first_checkpoint_outside": first_bad, "chunk": chunk, "rms_at_the_end": trail[-1][1], "largest_rms_seen": worst,
            "cells_of_other_sign_at_the_end": signs,
            "rms_by_checkpoint": [[s, float(f"{r:.3e}")] for s, r in trail]}


out = {"tolerance": TOL, "what": "RMS over all points of (FAST - STRICT), exact Gauss-Seidel ordering, STRICT == the reference bit for bit"}
G = "tests/golden"
surf = np.load(os.path.join(G, "surfaces.npz"))

# BASELINE configuration 2: cube40 at 256^3, to the reference's stop (sweep 3 299)
g = np.load(os.path.join(G, "cube40_256_converged.npz"))
X, E = surf["cube40_surfX"].astype(np.float64), surf["cube40_surfElem"]
dx, h = float(g["dx"]), float(g["h"])
n, xLo, mn, mx = stl_io.grid_from_surface(X, dx=dx, dd=10)
nx, ny, nz = n
phi0 = torch.ones((nx + 1) * (ny + 1) * (nz + 1), dtype=torch.float64, device="cuda")
lsf.phi0Init(phi0, nx, ny, nz, dx, xLo, mn, mx, X, E)
out["config 2: cube40.stl, 256^3, reinit to convergence"] = march("c2", phi0, nx, ny, nz, dx, h, 10001, 128, 1.0e-5)

# BASELINE configuration 3: twoCube10 on the cubic 512^3 grid, the configuration's 128 sweeps
g = np.load(os.path.join(G, "twocube10_512cubed_s128.npz"))
X, E = surf["twocube10_surfX"].astype(np.float64), surf["twocube10_surfElem"]
dx, h = float(g["dx"]), float(g["h"])
n, xLo, mn, mx = stl_io.grid_from_surface_pads(X, dx, g["pad_lo"], g["pad_hi"])
nx, ny, nz = n
phi0 = torch.ones((nx + 1) * (ny + 1) * (nz + 1), dtype=torch.float64, device="cuda")
lsf.phi0Init(phi0, nx, ny, nz, dx, xLo, mn, mx, X, E)
out["config 3: twoCube10.stl, 512^3, 128 sweeps"] = march("c3", phi0, nx, ny, nz, dx, h, 128, 32, 0.0)

# the field bench.py times: two spheres at 512^3, 2 048 sweeps
p0, dx = fields.two_sphere_phi0_device((512, 512, 512), torch.device("cuda"))
out["bench.py: two-sphere phi0, 512^3"] = march("bench", p0, 511, 511, 511, dx, fields.reinit_step(dx), 2048, 256, 0.0)
print(json.dumps(out, indent=1))
