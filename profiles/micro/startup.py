import time, ctypes, os, sys
t0 = time.perf_counter()
lib = ctypes.CDLL(os.environ.get("LIB") or os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "levelsetfortran_amd", "liblsf_hip.so"))
t1 = time.perf_counter()
lib.lsf_device_count.restype = ctypes.c_int
n = lib.lsf_device_count()
t2 = time.perf_counter()
import numpy as np
phi = np.ones((8, 8, 8), order="F"); nb = np.zeros((8, 8, 8), dtype=np.int32, order="F"); sb = nb.copy(order="F")
lib.lsf_narrowband.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_double]
r = lib.lsf_narrowband(phi.ctypes.data, nb.ctypes.data, sb.ctypes.data, 7, 7, 7, 0.1)
t3 = time.perf_counter()
r = lib.lsf_narrowband(phi.ctypes.data, nb.ctypes.data, sb.ctypes.data, 7, 7, 7, 0.1)
t4 = time.perf_counter()
print("dlopen %.1f ms, device count %.1f ms, first call (context + code object + kernel) %.1f ms, second call %.2f ms; devices %d rc %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, n, r))
