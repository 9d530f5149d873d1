"""CPU stand-in for levelsetfortran_amd.distributed.HipBackend, built on the oracle.

Lives under tests/ on purpose: only tests may use the oracle.  It lets the world_size-2 gloo tests
run the REAL decomposition / halo-exchange / reduction code of distributed.py on a machine without a
GPU.
"""
import contextlib
import ctypes

import numpy as np
import torch

import oracle_lib


class _NullStream:
    cuda_stream = 0

    def wait_stream(self, other):
        pass


class OracleBackend:
    def __init__(self):
        self.L = oracle_lib.lib()
        i9, i3 = ctypes.c_int * 9, ctypes.c_int * 3
        dp = ctypes.POINTER(ctypes.c_double)
        self.L.lsf_oracle_jacobi_box.restype = None
        self.L.lsf_oracle_jacobi_box.argtypes = [ctypes.c_void_p] * 3 + [i9, i3, i3, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
        self.L.lsf_oracle_bc_box.restype = None
        self.L.lsf_oracle_bc_box.argtypes = [ctypes.c_void_p] * 2 + [i9, i3, i3, ctypes.c_double, ctypes.c_void_p]
        self.compute = _NullStream()
        self.comm = _NullStream()
        self._i9, self._i3 = i9, i3

    def empty(self, n, dtype=None):
        return torch.empty(n, dtype=dtype or torch.float64)

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a.ravel(order="F")).copy())

    def to_numpy(self, t, shape):
        return t.numpy().reshape(shape, order="F").copy(order="F")

    def _box(self, b):
        return self._i9(*b.ext, *b.g0, *b.n)

    def _lohi(self, region):
        return self._i3(*[r[0] for r in region]), self._i3(*[r[1] for r in region])

    def sweep(self, a_in, a_out, phiS, b, region, dx, h, sumsq, stream):
        lo, hi = self._lohi(region)
        self.L.lsf_oracle_jacobi_box(a_in.data_ptr(), a_out.data_ptr(), phiS.data_ptr(), self._box(b), lo, hi, dx, h,
                                     sumsq.data_ptr())

    def bc(self, a_in, a_out, b, region, dx, sumsq, stream):
        lo, hi = self._lohi(region)
        self.L.lsf_oracle_bc_box(a_in.data_ptr(), a_out.data_ptr(), self._box(b), lo, hi, dx, sumsq.data_ptr())

    def _view(self, f, b):
        return f.view(b.ext[2], b.ext[1], b.ext[0])  # (k,j,i), i fastest

    def pack(self, f, b, region, buf, stream):
        (i0, i1), (j0, j1), (k0, k1) = region
        buf.copy_(self._view(f, b)[k0:k1, j0:j1, i0:i1].reshape(-1))

    def unpack(self, f, b, region, buf, stream):
        (i0, i1), (j0, j1), (k0, k1) = region
        self._view(f, b)[k0:k1, j0:j1, i0:i1] = buf.view(k1 - k0, j1 - j0, i1 - i0)

    def stream_ctx(self, stream):
        return contextlib.nullcontext()

    def wait(self, waiter, waited):
        pass

    def synchronize(self):
        pass
