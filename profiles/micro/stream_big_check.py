"""k_reinit_gs_stream against k_reinit_gs_persist on larger ragged grids (deep tile columns exist: the exact previous-sweep test
and the 16-byte continued loader run), both continuation policies, both lane maps: fields bit for bit, RMS traces bit for bit
within a tile shape (the column sums are added in the order of the tile columns, which depends on the shape)."""
import os
import sys

import torch

sys.path.insert(0, ".")
import levelsetfortran_amd as lsf  # noqa: E402
from levelsetfortran_amd import fields  # noqa: E402

bad = 0
for npts, sweeps in (((230, 214, 198), 40), ((300, 300, 300), 40), ((512, 512, 512), 24)):
    phi0, dx = fields.two_sphere_phi0_device(npts, "cuda:0")
    h = fields.reinit_step(dx)
    n = [v - 1 for v in npts]
    for arith in ("strict", "fast"):
        base_field = None
        for shape in ("c1x4", "2x2"):
            base = None
            for name, env in (("persist", {"LSF_GS_STREAM": "0"}), ("stream2", {"LSF_GS_STREAM": "1", "LSF_GS_CONT": "2"}),
                              ("stream1", {"LSF_GS_STREAM": "1", "LSF_GS_CONT": "1"})):
                for k in ("LSF_GS_STREAM", "LSF_GS_CONT"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                os.environ["LSF_GS_SKEW_W"] = shape
                t = phi0.clone()
                rep = lsf.reinit(t, None, None, *n, sweeps - 1, dx, h, tol=0.0, order="gs", arith=arith)
                if base is None:
                    base = (t, rep.rms)
                elif not (torch.equal(t, base[0]) and rep.rms == base[1]):
                    bad += 1
                    print("MISMATCH", npts, arith, shape, name, float((t - base[0]).abs().max()), flush=True)
            if base_field is None:
                base_field = base[0]
            elif not torch.equal(base[0], base_field):
                bad += 1
                print("MISMATCH between shapes", npts, arith, flush=True)
        print(npts, arith, "checked", flush=True)
print("big check:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
