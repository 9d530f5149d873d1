"""exact ordering across z slabs on ONE device (the rehearsal): kernel time per sweep from lsf_slabs_info against the single
launch of lsf_reinit.  python3 profiles/micro/slab_bench.py [N=512] [K=64]"""
import ctypes, os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import levelsetfortran_amd as L
from levelsetfortran_amd import fields, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = _lib.load()
phi0, dx = fields.two_sphere_phi0((N, N, N)); n = N - 1; h = fields.reinit_step(dx)
cells = float(n - 1) ** 3
def info():
    v = [ctypes.c_int(0) for _ in range(4)]; ks = ctypes.c_double(0)
    _lib.check(lib.lsf_slabs_info(*[ctypes.byref(x) for x in v], ctypes.byref(ks)))
    return [x.value for x in v], ks.value
for arith in ("fast", "strict"):
    a = phi0.copy(order="F")
    lib.lsf_profile(1)
    L.reinit(a, None, None, n, n, n, K - 1, dx, h, tol=0.0, order="gs", arith=arith)
    a = phi0.copy(order="F")
    L.reinit(a, None, None, n, n, n, K - 1, dx, h, tol=0.0, order="gs", arith=arith)
    ms = ctypes.c_double(0)
    _lib.check(lib.lsf_profile_get(ctypes.byref(ms), None, None, None, None))
    lib.lsf_profile(0)
    print(f"{arith}: single launch (lsf_reinit, one block per tile): kernel {ms.value / K:.3f} ms per sweep", flush=True)
    for slabs, fine in ((1, 0), (2, 0), (3, 0), (2, 1)):
        os.environ["LSF_SLAB_FINEGRAINED"] = str(fine)
        for rep in range(2):
            b = phi0.copy(order="F")
            t = time.time()
            r = L.reinit_multi(b, n, n, n, K - 1, dx, h, [0] * slabs, tol=0.0, arith=arith, order="gs")
            wall = time.time() - t
        (ns, grid, fg, sw), ks = info()
        print(f"{arith}: {slabs} slab(s) fine={fg} grid={grid}: kernel {ks / K * 1e3:.3f} ms per sweep ({cells * K / ks:.3e} cell-updates/s), "
              f"call {wall:.2f} s, equal to single launch: {np.array_equal(a, b)}", flush=True)
