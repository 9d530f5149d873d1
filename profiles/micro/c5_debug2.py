"""python3 profiles/micro/c5_debug2.py  -- which blocks / planes of a decomposed sweep go wrong at size"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import levelsetfortran_amd as lsf
from levelsetfortran_amd import fields
from test_gpu_configs45 import _build, _as_host_field

def case(N, dtype, dims, arith="fast", env=None):
    for k_, v_ in (env or {}).items():
        os.environ[k_] = v_
    phi, dx = _build(N, ((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)), 0.5, dtype)
    h = fields.reinit_step(dx)
    host = _as_host_field(phi).copy(order="F")
    kw = {} if dtype == torch.float32 else {"arith": arith}
    lsf.reinit(phi.reshape(-1), None, None, N - 1, N - 1, N - 1, 0, dx, h, tol=0.0, order="jacobi", **kw)
    want = _as_host_field(phi)
    del phi; torch.cuda.empty_cache(); lsf._lib.load().lsf_release_workspace()
    nb = dims[0] * dims[1] * dims[2]
    try:
        lsf.reinit_multi(host, N - 1, N - 1, N - 1, 0, dx, h, [0] * nb, dims=dims, tol=0.0, **kw)
        err = "ok"
    except Exception as e:
        err = str(e)[:40]
    bad = ~(host == want)
    msg = f"N={N} {str(dtype)[6:]} {arith} dims={dims} env={env}: {err}; differing {int(bad.sum())}"
    if bad.any():
        perk = bad.sum(axis=(0, 1))
        ks = np.nonzero(perk)[0]
        msg += f"; k planes {ks.min()}..{ks.max()} ({len(ks)}); first planes counts {[(int(k), int(perk[k])) for k in ks[:6]]}"
        perj = bad.sum(axis=(0, 2)); js = np.nonzero(perj)[0]
        msg += f"; j {js.min()}..{js.max()} ({len(js)})"
        peri = bad.sum(axis=(1, 2)); is_ = np.nonzero(peri)[0]
        msg += f"; i {is_.min()}..{is_.max()} ({len(is_)})"
    print(msg, flush=True)
    for k_ in (env or {}):
        os.environ.pop(k_)
    lsf._lib.load().lsf_release_workspace()

case(1024, torch.float64, (1, 1, 2), "fast")
case(1024, torch.float64, (1, 1, 2), "strict")
case(768, torch.float32, (1, 1, 2))
case(896, torch.float32, (1, 1, 2))
case(1024, torch.float32, (1, 1, 2))
case(1024, torch.float32, (1, 1, 2), env={"LSF_JAC_SH": "0"})
case(1024, torch.float32, (1, 1, 2), env={"LSF_MULTI_TRANSPORT": "mock"})
