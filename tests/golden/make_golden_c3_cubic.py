#!/usr/bin/env python3
"""BASELINE config 3 at its STATED size: twoCube10.stl on a cubic 512^3 grid, reinit + 200 min/max iterations.

dx = 12/489.5 gives 512 points along x with the host's 10 pad cells; the 12 x 1 x 1 bounding box needs PER-AXIS
padding to become cubic (host edit E4b, INTEGRATION.md: LSF_DD_Y_LO/HI, LSF_DD_Z_LO/HI): 42 + 234 + 235 = 511 cells
in y and z.  twoCube10 diverges in the reference as shipped (NaN near sweep 265 at this resolution, SURVEY.md
section 0), so the comparison is at FIXED sweep counts: phi0 from the oracle's restatement of set3d.f90:196-268
(pinned bit for bit to the reference's phi0 on both sample surfaces), SWEEPS sweeps by the reference's OWN `reinit`
(oracle/_ref, ctypes), narrowBand + 200 min/max iterations by the oracle (pinned bit for bit to the reference's
406-iteration run).  SHA-256 + strided samples only.

  python tests/golden/make_golden_c3_cubic.py 16     # ~15 CPU-minutes  -> twocube10_512cubed_s16.npz
  python tests/golden/make_golden_c3_cubic.py 128    # ~2 CPU-hours     -> twocube10_512cubed_s128.npz

Build container only (needs oracle/_ref and ~12 GB of memory).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from make_golden import ref_reinit, sha  # noqa: E402

import oracle_lib  # noqa: E402
import stl_io  # noqa: E402

DX = 12.0 / 489.5
MM_ITERS = 200
PAD_LO, PAD_HI = (10, 234, 234), (10, 235, 235)


def main():
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    s = np.load(os.path.join(HERE, "surfaces.npz"))
    X, E = s["twocube10_surfX"].astype(np.float64), s["twocube10_surfElem"]
    n, xLo, mn, mx = stl_io.grid_from_surface_pads(X, DX, PAD_LO, PAD_HI)
    assert tuple(n) == (511, 511, 511), n
    t0 = time.time()
    phi0 = oracle_lib.phi0(n[0], n[1], n[2], DX, xLo, mn, mx, X, E)
    print("phi0", time.time() - t0, flush=True)
    ext = mx - mn
    dxx = DX / np.sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2])  # set3d.f90:301
    h, h1 = 0.1 * dxx, 0.01 * dxx
    f, tr = ref_reinit(phi0, n[0], n[1], n[2], sweeps - 1, DX, h)
    print("reinit", time.time() - t0, len(tr), flush=True)
    assert f is not None and len(tr) == sweeps and not np.isnan(tr).any()
    nb, sb = oracle_lib.narrowband(n[0], n[1], n[2], DX, f)
    nb0_sha, sb0_sha = sha(nb), sha(sb)
    g = f.copy(order="F")
    rc, its, trm = oracle_lib.minmax(g, nb, sb, n[0], n[1], n[2], MM_ITERS, DX, h1)
    print("minmax", time.time() - t0, its, flush=True)
    assert rc == 0
    smp = lambda a: np.ascontiguousarray(a[::16, ::8, ::8])
    np.savez_compressed(os.path.join(HERE, f"twocube10_512cubed_s{sweeps}.npz"), dx=DX, h=h, h1=h1, n=np.array(n),
                        pad_lo=np.array(PAD_LO), pad_hi=np.array(PAD_HI), xLo=xLo, sweeps=sweeps, mm_iters=its,
                        phi0_sha=sha(phi0), reinit_sha=sha(f), reinit_sample=smp(f), rms=tr, NB0_sha=nb0_sha, SB0_sha=sb0_sha,
                        minmax_sha=sha(g), minmax_sample=smp(g), rms_minmax=trm, NB_sha=sha(nb), SB_sha=sha(sb))
    print("done", its, tr[-1], trm[-1] if len(trm) else None, time.time() - t0)


if __name__ == "__main__":
    main()
