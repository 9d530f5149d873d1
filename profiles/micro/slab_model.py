"""One-GPU scaling model of the exact ordering across z slabs.

Needs the experiment build (make -C levelsetfortran_amd/csrc OUT=../../build/exp/liblsf_x.so EXTRA=-DLSF_EXPERIMENTS) and
LSF_LIB_PATH pointing at it.  LSF_SLAB_MODEL=d/D makes lsf_reinit_multi(LSF_ORDER_GS) run ONE slab of a D-slab run alone on the
device, with everything it would wait for from its neighbours granted in advance (wrong field, pure throughput): the time of
its launch is what a device of a D-device node needs for its share of the tile graph if communication were free -- an upper
bound of the speed-up D devices can reach: T(1) / max over slabs T(d/D).

  LSF_LIB_PATH=$PWD/build/exp/liblsf_x.so python profiles/micro/slab_model.py [sizes=512,1024] [D=1,2,4,8] [sweeps=32] [arith=fast,strict]
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import levelsetfortran_amd as lsf  # noqa: E402
from levelsetfortran_amd import _lib, fields  # noqa: E402

kv = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
sizes = [int(v) for v in kv.get("sizes", "512,1024").split(",")]
Ds = [int(v) for v in kv.get("D", "1,2,4,8").split(",")]
sweeps = int(kv.get("sweeps", "32"))
ariths = kv.get("arith", "fast,strict").split(",")
lib = _lib.load()


def one(phi0, n, dx, h, arith, model):
    if model:
        os.environ["LSF_SLAB_MODEL"] = model
    else:
        os.environ.pop("LSF_SLAB_MODEL", None)
    best = None
    for _ in range(2):
        a = phi0.copy(order="F")
        lsf.reinit_multi(a, n, n, n, sweeps - 1, dx, h, [0], tol=0.0, arith=arith, order="gs")
        s, b, f, sw, ks = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        lib.lsf_slabs_info(ctypes.byref(s), ctypes.byref(b), ctypes.byref(f), ctypes.byref(sw), ctypes.byref(ks))
        ms = ks.value * 1e3 / sweeps
        best = ms if best is None else min(best, ms)
    return best


for N in sizes:
    phi0, dx = fields.two_sphere_phi0((N, N, N))
    h = fields.reinit_step(dx)
    n = N - 1
    for arith in ariths:
        t1 = one(phi0, n, dx, h, arith, None)
        print(f"{N}^3 {arith}: the whole tile graph on one device: {t1:.3f} ms per sweep ({sweeps} sweeps per launch)", flush=True)
        for D in Ds:
            if D == 1:
                continue
            ts = {}
            for d in sorted({0, D // 2, D - 1}):
                ts[d] = one(phi0, n, dx, h, arith, f"{d}/{D}")
            worst = max(ts.values())
            print(f"   D = {D}: slab " + ", ".join(f"{d}: {t:.3f} ms" for d, t in ts.items()) + f" -> speed-up <= {t1 / worst:.2f} x ({t1 / worst / D * 100:.0f} % of {D})",
                  flush=True)
    del phi0
