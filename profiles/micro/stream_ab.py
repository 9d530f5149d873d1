"""Column continuation (k_reinit_gs_stream) against the one-block-per-tile launch (k_reinit_gs_persist), in ONE process on
ONE box: bit-identity of the fields and RMS traces on a set of grids, then kernel time per sweep (lsf_profile events).

  python profiles/micro/stream_ab.py [check] [time] [sizes=512,256] [sweeps=64] [shapes=default,c1x4,2x2]

Environment switches are read by the library on every call, so one process can run all variants.
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import levelsetfortran_amd as lsf  # noqa: E402
from levelsetfortran_amd import _lib, fields  # noqa: E402

VARIANTS = {"persist": {"LSF_GS_STREAM": "0"}, "stream": {"LSF_GS_STREAM": "1"}, "nocont": {"LSF_GS_STREAM": "1", "LSF_GS_CONT": "0"}}


def setenv(d):
    for k in ("LSF_GS_STREAM", "LSF_GS_CONT", "LSF_GS_SKEW_W"):
        os.environ.pop(k, None)
    os.environ.update(d)


def run(phi0, npts, sweeps, arith, env, tol=0.0, prof=False):
    setenv(env)
    nx, ny, nz = (v - 1 for v in npts)
    dx = 3.0 / (npts[0] - 1)
    h = fields.reinit_step(dx)
    t = phi0.clone()
    lib = _lib.load()
    if prof:
        lib.lsf_profile(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = lsf.reinit(t, None, None, nx, ny, nz, sweeps - 1, dx, h, tol=tol, order="gs", arith=arith)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kms = None
    if prof:
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n, s = ctypes.c_longlong(), ctypes.c_int()
        lib.lsf_profile_get(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(n), ctypes.byref(s))
        kms = a.value
        lib.lsf_profile(0)
    return t, rep, wall, kms


def check(shapes=(None, "c1x4", "2x2", "c1x2", "4x2")):
    bad = 0
    grids = [(40, 33, 27), (70, 21, 45), (65, 8, 30), (23, 23, 5), (96, 96, 96), (130, 75, 101), (160, 160, 160)]
    for npts in grids:
        phi_h, dx = fields.two_sphere_phi0(npts)
        phi0 = torch.from_numpy(np.ascontiguousarray(phi_h.ravel(order="F"))).cuda()
        for arith in ("strict", "fast"):
            for shape in shapes:
                base = None
                for name, env in VARIANTS.items():
                    e = dict(env)
                    if shape:
                        e["LSF_GS_SKEW_W"] = shape
                    t, rep, _, _ = run(phi0, npts, 19, arith, e)
                    if base is None:
                        base = (t, rep)
                        continue
                    same = torch.equal(t, base[0]) and rep.count == base[1].count and rep.rms == base[1].rms
                    if not same:
                        bad += 1
                        d = (t - base[0]).abs().max().item()
                        print(f"MISMATCH {npts} {arith} shape={shape} {name}: max diff {d:.3e} count {rep.count}/{base[1].count}", flush=True)
        print(f"grid {npts}: checked", flush=True)
    # a stop inside the run
    npts = (96, 96, 96)
    phi_h, dx = fields.two_sphere_phi0(npts)
    phi0 = torch.from_numpy(np.ascontiguousarray(phi_h.ravel(order="F"))).cuda()
    _, rep0, _, _ = run(phi0, npts, 40, "strict", VARIANTS["persist"])
    tols = [0.5 * (rep0.rms[k] + rep0.rms[k + 1]) for k in (6, 17, 30) if rep0.rms[k + 1] < rep0.rms[k]]
    for tol in tols:
        base = None
        for name, env in VARIANTS.items():
            t, rep, _, _ = run(phi0, npts, 60, "strict", env, tol=tol)
            if base is None:
                base = (t, rep)
                print(f"stop test tol {tol}: {rep.count} sweeps, converged {rep.converged}")
            elif not (torch.equal(t, base[0]) and rep.count == base[1].count):
                bad += 1
                print(f"MISMATCH stop tol={tol} {name}: count {rep.count}/{base[1].count}", flush=True)
    print("check:", "OK" if bad == 0 else f"{bad} MISMATCHES", flush=True)
    return bad


def timing(sizes, sweeps, shapes):
    for n in sizes:
        npts = (n, n, n)
        phi0, dx = fields.two_sphere_phi0_device(npts, "cuda:0")
        for arith in ("fast", "strict"):
            for shape in shapes:
                for name, env in VARIANTS.items():
                    e = dict(env)
                    if shape != "default":
                        e["LSF_GS_SKEW_W"] = shape
                    run(phi0, npts, sweeps, arith, e)  # warm-up (plans, buffers)
                    best = None
                    for _ in range(2):
                        _, rep, wall, kms = run(phi0, npts, sweeps, arith, e, prof=True)
                        best = kms if best is None else min(best, kms)
                    print(f"{n}^3 {arith:6s} {shape:7s} {name:8s}: kernel {best / sweeps:.3f} ms/sweep ({sweeps} sweeps)", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    kv = dict(a.split("=", 1) for a in args if "=" in a)
    rc = 0
    if "check" in args:
        rc = check(tuple(None if v == "default" else v for v in kv["shapes"].split(",")) if "shapes" in kv else (None, "c1x4", "2x2", "c1x2", "4x2"))
    if "time" in args:
        timing([int(v) for v in kv.get("sizes", "512,256").split(",")], int(kv.get("sweeps", "64")), kv.get("shapes", "default").split(","))
    sys.exit(1 if rc else 0)
