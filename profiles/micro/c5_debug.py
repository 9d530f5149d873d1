"""where does the fp32 decomposed sweep (eight blocks on one device) leave the single-domain sweep?  python3 profiles/micro/c5_debug.py N [sweeps]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import levelsetfortran_amd as lsf
from levelsetfortran_amd import fields
from test_gpu_configs45 import _build, _as_host_field
for N in [int(v) for v in sys.argv[1].split(',')]:
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    phi, dx = _build(N, ((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)), 0.5, torch.float32)
    h = fields.reinit_step(dx)
    host = _as_host_field(phi)
    try:
        rep1 = lsf.reinit(phi.reshape(-1), None, None, N - 1, N - 1, N - 1, s - 1, dx, h, tol=0.0, order="jacobi")
    except Exception as e:
        print(N, "single-domain failed", e); continue
    want = _as_host_field(phi)
    del phi; torch.cuda.empty_cache(); lsf._lib.load().lsf_release_workspace()
    for dims in ((2, 2, 2), (1, 2, 2), (1, 1, 2), (2, 1, 1)):
        got = host.copy(order="F")
        nb = dims[0] * dims[1] * dims[2]
        try:
            rep = lsf.reinit_multi(got, N - 1, N - 1, N - 1, s - 1, dx, h, [0] * nb, dims=dims, tol=0.0)
            err = None
        except Exception as e:
            err = str(e)[:80]
        bad = np.argwhere(~(got == want))
        nan = int(np.isnan(got).sum())
        msg = f"N={N} dims={dims}: {'ok' if err is None else err}; differing points {len(bad)}, NaN {nan}"
        if len(bad):
            msg += f"; first {bad[0].tolist()} last {bad[-1].tolist()} i-range {bad[:,0].min()}..{bad[:,0].max()} j {bad[:,1].min()}..{bad[:,1].max()} k {bad[:,2].min()}..{bad[:,2].max()}"
        print(msg, flush=True)
        lsf._lib.load().lsf_release_workspace()
