"""Single-precision Jacobi reinit (BASELINE.json configuration 5, include/lsf.h "single precision").

The reference is fp64 only, so nothing here can be bit-identical to it.  What is checked, through the C ABI:
  * the fp32 field tracks the fp64 ORACLE (Jacobi ordering, same fp32-rounded input) within a stated tolerance,
    with the inside/outside sign exact away from the zero level set;
  * properties that need no reference: host seam == device seam, block-decomposed == single-domain bit for bit
    (odd pair alignments, the thin x rim), constant fields, grids with no WENO cell, argument errors.
Tolerances: one sweep changes phi by h*sgn*(1-gM) with h = 0.1 dx/sqrt(12); fp32 rounding of phi (6e-8 per
update) dominates the error of the one-sided differences, so after K sweeps the fields agree to a few K * 6e-8.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

F32_EPS = float(np.finfo(np.float32).eps)


def _dev(a32):
    return torch.from_numpy(np.ascontiguousarray(a32.ravel(order="F"))).to("cuda:0")


def _back(t, shape):
    return t.cpu().numpy().reshape(shape, order="F")


@pytest.mark.parametrize("npts,kind", [((40, 33, 27), "two"), ((24, 24, 24), "one"), ((48, 41, 44), "two"),
                                       ((12, 9, 8), "two")])
def test_f32_tracks_the_fp64_oracle(oracle, npts, kind):
    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import fields

    phi0, dx = (fields.two_sphere_phi0 if kind == "two" else fields.sphere_phi0)(npts)
    nx, ny, nz = (v - 1 for v in npts)
    h = fields.reinit_step(dx)
    sweeps = 9
    in32 = np.asfortranarray(phi0.astype(np.float32))
    ref = np.asfortranarray(in32.astype(np.float64))
    _, n_ref, tr_ref = oracle.reinit(ref, nx, ny, nz, sweeps - 1, dx, h, tol=0.0, order=oracle.JACOBI)
    d = _dev(in32)
    rep = lsf.reinit(d, None, None, nx, ny, nz, sweeps - 1, dx, h, tol=0.0, order="jacobi", arith="fast")
    got = _back(d, in32.shape)
    assert got.dtype == np.float32 and rep.count == n_ref == sweeps
    err = got.astype(np.float64) - ref
    assert np.isfinite(got).all()
    assert np.abs(err).max() < 4 * sweeps * F32_EPS, np.abs(err).max()
    assert np.sqrt(np.mean(err ** 2)) < sweeps * F32_EPS
    # inside/outside: exact wherever the fp64 value is not within the tolerance of zero
    far = np.abs(ref) > 4 * sweeps * F32_EPS
    assert np.array_equal(np.signbit(got[far]), np.signbit(ref[far]))
    # the RMS trace (double accumulation of fp32 differences) follows the oracle's
    assert np.allclose(rep.rms, tr_ref[:sweeps], rtol=2e-3, atol=1e-9)


def test_f32_host_seam_equals_device_seam():
    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import fields

    npts = (40, 33, 27)
    phi0, dx = fields.two_sphere_phi0(npts)
    nx, ny, nz = (v - 1 for v in npts)
    h = fields.reinit_step(dx)
    in32 = np.asfortranarray(phi0.astype(np.float32))
    host = in32.copy(order="F")
    rep_h = lsf.reinit(host, None, None, nx, ny, nz, 12, dx, h, tol=0.0, order="jacobi")
    d = _dev(in32)
    rep_d = lsf.reinit(d, None, None, nx, ny, nz, 12, dx, h, tol=0.0, order="jacobi")
    assert np.array_equal(host, _back(d, in32.shape))
    assert rep_h.rms == rep_d.rms and rep_h.count == 13


def test_f32_stops_on_tolerance_and_keeps_far_field_finite():
    """The reference's phi0 is exactly 1.0 outside the search box (set3d.f90:161): first differences are exactly 0
    there, IS = 0 and eps = the floor.  fp32 must take that through the WENO weights without 0/0 -- that is what
    the fp32 epsilon floor and the normalisation of the q_k are for."""
    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import fields

    npts = (48, 45, 50)
    x, y, z, dx = fields.grid_axes(npts)
    d = np.sqrt(x[:, None, None] ** 2 + y[None, :, None] ** 2 + z[None, None, :] ** 2) - 1.0
    phi0 = np.where(np.abs(d) < 4 * dx, d / np.sqrt(d * d + dx * dx), np.sign(d))
    nx, ny, nz = (v - 1 for v in npts)
    h = fields.reinit_step(dx)
    in32 = np.asfortranarray(phi0.astype(np.float32))
    assert (np.abs(in32) == 1.0).mean() > 0.5
    dev = _dev(in32)
    full = lsf.reinit(dev, None, None, nx, ny, nz, 19, dx, h, tol=0.0, order="jacobi")
    assert np.isfinite(_back(dev, in32.shape)).all() and full.count == 20 and np.isfinite(full.rms).all()
    # stop test: first sweep whose RMS is below tol is the last one executed (subs.f90:915)
    tol = full.rms[9] * 1.0001
    expect = next(s for s, r in enumerate(full.rms) if r < tol) + 1
    dev2 = _dev(in32)
    rep = lsf.reinit(dev2, None, None, nx, ny, nz, 19, dx, h, tol=tol, order="jacobi")
    assert rep.converged and rep.count == expect and rep.rms == full.rms[:expect]
    const = torch.full(((nx + 1) * (ny + 1) * (nz + 1),), 0.75, dtype=torch.float32, device="cuda:0")
    lsf.reinit(const, None, None, nx, ny, nz, 3, dx, h, tol=0.0, order="jacobi")
    c = _back(const, in32.shape)
    # a constant field has zero gradient: every interior cell moves by the same h*sgn*(1-0) per sweep
    assert np.isfinite(c).all() and np.ptp(c[1:-1, 1:-1, 1:-1]) == 0.0 and c[5, 5, 5] > 0.75


def test_f32_exists_for_jacobi_fast_only():
    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import LsfError

    n = 15
    d = torch.zeros((n + 1) ** 3, dtype=torch.float32, device="cuda:0")
    for kw in ({"order": "gs"}, {"order": "jacobi", "arith": "strict"}):
        with pytest.raises(LsfError) as e:
            lsf.reinit(d, None, None, n, n, n, 1, 0.1, 0.01, **kw)
        assert e.value.code == 2 and "fp32" in str(e.value)


def test_f32_against_fp64_gpu_at_256():
    """BASELINE-sized check without an oracle run: fp32 vs the fp64 GPU Jacobi path (itself pinned to the oracle)."""
    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import fields

    N, sweeps = 256, 32
    phi64, dx = fields.two_sphere_phi0_device((N, N, N), torch.device("cuda:0"))
    h = fields.reinit_step(dx)
    phi32 = phi64.to(torch.float32)
    phi64 = phi32.to(torch.float64)  # same input
    lsf.reinit(phi64, None, None, N - 1, N - 1, N - 1, sweeps - 1, dx, h, tol=0.0, order="jacobi")
    lsf.reinit(phi32, None, None, N - 1, N - 1, N - 1, sweeps - 1, dx, h, tol=0.0, order="jacobi")
    err = phi32.to(torch.float64) - phi64
    assert float(err.abs().max()) < 4 * sweeps * F32_EPS
    assert float(err.pow(2).mean().sqrt()) < sweeps * F32_EPS
    far = phi64.abs() > 4 * sweeps * F32_EPS
    assert bool((torch.signbit(phi32)[far] == torch.signbit(phi64)[far]).all())


# ------------------------------------------------------------------------------------------------
# block-decomposed fp32 == single-domain fp32, bit for bit (several ranks sharing the one GPU; gloo through
# pinned host buffers).  Odd owned ranges put the cell pairs of the kernel on different rows than the
# single-domain sweep does; x cuts exercise the thin-rim variant.
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dims, npts, sweeps, outdir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from levelsetfortran_amd import distributed as D, fields

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n = tuple(v - 1 for v in npts)
    b = D.make_block(rank, dims, n)
    rng = tuple((g, g + e) for g, e in zip(b.g0, b.ext))
    phi_np, dx = fields.two_sphere_phi0(npts, ranges=rng)
    h = fields.reinit_step(dx)
    be = D.HipBackend(dev, host_staging=True, dtype="f32")
    dr = D.DistributedReinit(be, b, dx, h)
    out, nsw, rms = dr.run(be.from_numpy(phi_np), sweeps - 1, tol=0.0)
    assert out.dtype == torch.float32
    own = tuple(slice(lo, hi) for lo, hi in b.own_local)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), own=np.array(b.own), data=be.to_numpy(out, b.ext)[own], nsw=nsw,
             rms=np.array(rms))
    dist.destroy_process_group()


@pytest.mark.parametrize("dims,npts", [((2, 1, 1), (48, 140, 20)), ((1, 2, 1), (30, 47, 26)), ((2, 2, 1), (41, 45, 24))])
def test_f32_decomposed_equals_single_domain(tmp_path, dims, npts):
    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import fields

    sweeps = 6
    world = int(np.prod(dims))
    mp.spawn(_worker, args=(world, _free_port(), dims, npts, sweeps, str(tmp_path)), nprocs=world, join=True)
    phi0, dx = fields.two_sphere_phi0(npts)
    nx, ny, nz = (v - 1 for v in npts)
    ref = np.asfortranarray(phi0.astype(np.float32))
    rep = lsf.reinit(ref, None, None, nx, ny, nz, sweeps - 1, dx, fields.reinit_step(dx), tol=0.0, order="jacobi")
    got = np.full_like(ref, np.nan)
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        sl = tuple(slice(int(s), int(e)) for s, e in z["own"])
        got[sl] = z["data"]
        assert int(z["nsw"]) == sweeps
        assert np.allclose(z["rms"], rep.rms, rtol=1e-6, atol=0)  # same squares, summed in another order
    assert np.array_equal(got, ref)
