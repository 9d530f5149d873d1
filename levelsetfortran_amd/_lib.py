"""ctypes binding of liblsf_hip.so (include/lsf.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (``make -C
levelsetfortran_amd/csrc``).  There is no CPU fallback anywhere in this package: if the library is
missing, or no gfx950 device is usable, the compute entry points raise.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_double, c_int, c_int32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSF_LIB_PATH") or os.path.join(_HERE, "liblsf_hip.so")  # override: experiments only

# include/lsf.h
LSF_OK, LSF_ERR_NAN, LSF_ERR_INVALID, LSF_ERR_HIP, LSF_ERR_NO_DEVICE = 0, 1, 2, 3, 4
LSF_ORDER_GS, LSF_ORDER_JACOBI = 0, 1
LSF_ARITH_FAST, LSF_ARITH_STRICT = 0x000, 0x100


LSF_MIRROR_TRUST, LSF_MIRROR_LAZY = 1, 2  # include/lsf.h: lsf_mirror flags
LSF_TRANSPORT_PEER, LSF_TRANSPORT_RCCL, LSF_TRANSPORT_MOCK = 0, 1, 2  # include/lsf.h: lsf_multi_configure


class LsfBox(ctypes.Structure):
    """struct lsf_box (include/lsf.h)."""

    _fields_ = [(n, c_int) for n in ("lx", "ly", "lz", "gx0", "gy0", "gz0", "nx", "ny", "nz")]


class LsfError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"liblsf_hip error {code}: {message}")
        self.code = code


class LsfNaNError(LsfError):
    """RMS became NaN: the reference executes STOP here (subs.f90:926 / set3d.f90:458)."""


_dp, _ip, _i3 = POINTER(c_double), POINTER(c_int32), POINTER(c_int * 3)

# name -> (restype, argtypes); every symbol include/lsf.h declares
SIGNATURES = {
    "lsf_version": (c_int, []),
    "lsf_last_error": (ctypes.c_char_p, []),
    "lsf_device_count": (c_int, []),
    "lsf_set_device": (c_int, [c_int]),
    "lsf_release_workspace": (c_int, []),
    "lsf_skew_wide_fits": (c_int, [c_int, c_int, c_int]),
    "lsf_profile": (c_int, [c_int]),
    "lsf_copy_bandwidth": (c_int, [ctypes.c_size_t, c_int, POINTER(c_double)]),
    "lsf_profile_kernel": (ctypes.c_char_p, []),
    "lsf_profile_get": (c_int, [POINTER(c_double), POINTER(c_double), POINTER(c_double),
                                POINTER(ctypes.c_longlong), POINTER(c_int)]),
    "lsf_reinit": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_int,
                           POINTER(c_int), c_void_p, c_int]),
    "lsf_reinit_device": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double,
                                  c_int, c_int, POINTER(c_int), c_void_p, c_int, c_void_p]),
    "lsf_minmax": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_double,
                           c_double, c_int, POINTER(c_int), c_void_p, c_int]),
    "lsf_minmax_device": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_double,
                                  c_double, c_int, POINTER(c_int), c_void_p, c_int, c_void_p]),
    "lsf_narrowband": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_double]),
    "lsf_narrowband_device": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_void_p]),
    "lsf_phi0": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                         c_void_p, c_int]),
    "lsf_phi0_device": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p, c_int, c_void_p]),
    "lsf_advect_nodes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_int, c_int]),
    "lsf_advect_nodes_device": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_int,
                                        c_int, c_void_p]),
    "lsf_mirror": (c_int, [c_int]),
    "lsf_mirror_sync": (c_int, [c_void_p]),
    "lsf_mirror_forget": (c_int, [c_void_p]),
    "lsf_snapshot": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int]),
    "lsf_sumsq_diff": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_double)]),
    "lsf_write_vti": (c_int, [ctypes.c_char_p, c_void_p, c_int, c_int, c_int, c_double, c_void_p]),
    "lsf_stl_read": (c_int, [ctypes.c_char_p, POINTER(c_int), POINTER(c_int)]),
    "lsf_stl_get": (c_int, [c_void_p, c_void_p]),
    "lsf_box_reserve": (c_int, [c_void_p, ctypes.c_size_t]),
    "lsf_reinit_multi": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_int, c_void_p, c_int,
                                 c_void_p, POINTER(c_int), c_void_p, c_int]),
    "lsf_reinit_multi_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_int, c_void_p,
                                     c_int, c_void_p, POINTER(c_int), c_void_p, c_int]),
    "lsf_multi_create": (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, POINTER(c_void_p)]),
    "lsf_multi_destroy": (c_int, [c_void_p]),
    "lsf_multi_block": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int)]),
    "lsf_multi_scatter": (c_int, [c_void_p, c_void_p]),
    "lsf_multi_upload_block": (c_int, [c_void_p, c_int, c_void_p]),
    "lsf_multi_run": (c_int, [c_void_p, c_int, c_double, c_double, c_double, c_int, POINTER(c_int), c_void_p, c_int]),
    "lsf_multi_gather": (c_int, [c_void_p, c_void_p]),
    "lsf_slabs_info": (c_int, [POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_double)]),
    "lsf_peer_selftest": (c_int, [c_int, c_int, POINTER(c_int)]),
    "lsf_multi_defaults_get": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "lsf_multi_defaults": (c_int, [c_int, c_int]),
    "lsf_multi_configure": (c_int, [c_void_p, c_int, c_int]),
    "lsf_multi_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_double),
                                POINTER(c_double), POINTER(c_double), POINTER(c_int)]),
    "lsf_jacobi_sweep_box": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(LsfBox), c_int * 3, c_int * 3,
                                     c_double, c_double, c_int, c_void_p, c_void_p]),
    "lsf_bc_box": (c_int, [c_void_p, c_void_p, POINTER(LsfBox), c_int * 3, c_int * 3, c_double, c_void_p,
                           c_void_p]),
    "lsf_sumsq_begin": (c_int, [c_void_p]),
    "lsf_sumsq_end": (c_int, [c_void_p]),
    "lsf_pack_boxes": (c_int, [c_void_p, POINTER(LsfBox), c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lsf_unpack_boxes": (c_int, [c_void_p, POINTER(LsfBox), c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lsf_pack_boxes_f32": (c_int, [c_void_p, POINTER(LsfBox), c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lsf_unpack_boxes_f32": (c_int, [c_void_p, POINTER(LsfBox), c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lsf_pack_box": (c_int, [c_void_p, POINTER(LsfBox), c_int * 3, c_int * 3, c_void_p, c_void_p]),
    "lsf_unpack_box": (c_int, [c_void_p, POINTER(LsfBox), c_int * 3, c_int * 3, c_void_p, c_void_p]),
    "lsf_reinit_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_int,
                               POINTER(c_int), c_void_p, c_int]),
    "lsf_reinit_f32_device": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double,
                                      c_int, POINTER(c_int), c_void_p, c_int, c_void_p]),
    "lsf_jacobi_sweep_box_f32": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(LsfBox), c_int * 3, c_int * 3,
                                         c_double, c_double, c_int, c_void_p, c_void_p]),
    "lsf_bc_box_f32": (c_int, [c_void_p, c_void_p, POINTER(LsfBox), c_int * 3, c_int * 3, c_double, c_void_p,
                               c_void_p]),
    "lsf_pack_box_f32": (c_int, [c_void_p, POINTER(LsfBox), c_int * 3, c_int * 3, c_void_p, c_void_p]),
    "lsf_unpack_box_f32": (c_int, [c_void_p, POINTER(LsfBox), c_int * 3, c_int * 3, c_void_p, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load liblsf_hip.so and bind every symbol of include/lsf.h.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  levelsetfortran_amd has no CPU fallback."
            )
        # One HIP runtime per process: torch ships a libamdhip64 of its own, and whichever copy is loaded first serves every later
        # `libamdhip64.so` dependency.  Every test, the bench and the drivers run with torch imported first; a process that loaded this
        # library first and initialised torch.cuda afterwards found "No HIP GPUs are available" (round 6, profiles/micro/time_entries.py).
        # So: torch first, always -- the package needs it for device memory and streams anyway.
        try:
            import torch  # noqa: F401
        except ImportError:  # host-only use (lsf_stl_read, the ABI checks): nothing to keep consistent
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc == LSF_OK:
        return
    msg = (load().lsf_last_error() or b"").decode("utf-8", "replace")
    if rc == LSF_ERR_NAN:
        raise LsfNaNError(rc, msg)
    raise LsfError(rc, msg)


def int3(v) -> "ctypes.Array":
    return (c_int * 3)(int(v[0]), int(v[1]), int(v[2]))
