"""k_reinit_jacobi_strict_sh against the per-cell STRICT kernel (LSF_JAC_SH=0) at full size: same bits after a few sweeps.
usage: python3 profiles/micro/jss_check.py [N=512] [sweeps=3]   (N = points per axis)"""
import os
import sys

sys.path.insert(0, ".")
import torch

import levelsetfortran_amd as lsf
from levelsetfortran_amd import fields

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
phi0, dx = fields.two_sphere_phi0((N, N, N))
h = fields.reinit_step(dx)
dev = torch.device("cuda", 0)
out = []
for env in ("0", None):
    if env is None:
        os.environ.pop("LSF_JAC_SH", None)
    else:
        os.environ["LSF_JAC_SH"] = env
    t = torch.from_numpy(phi0.reshape(-1, order="F").copy()).to(dev)
    r = lsf.reinit(t, None, None, N - 1, N - 1, N - 1, S - 1, dx, h, tol=0.0, order="jacobi", arith="strict")
    torch.cuda.synchronize()
    out.append(t)
    print("LSF_JAC_SH", env, "rms", list(r.rms)[:S], flush=True)
same = bool((out[0] == out[1]).all())
print("N", N, "sweeps", S, "bitwise equal:", same, "max diff", float((out[0] - out[1]).abs().max()), flush=True)
sys.exit(0 if same else 1)
