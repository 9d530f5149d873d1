"""World-size-2 (and 4) runs of the REAL distributed sweep loop over gloo on CPU, with the oracle as the
arithmetic backend (tests/oracle_backend.py).  The decomposed Jacobi result must equal the
single-domain Jacobi sweep bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


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
    from oracle_backend import OracleBackend

    n = tuple(v - 1 for v in npts)
    b = D.make_block(rank, dims, n)
    rng = tuple((g, g + e) for g, e in zip(b.g0, b.ext))
    phi_np, dx = fields.two_sphere_phi0(npts, ranges=rng)
    h = fields.reinit_step(dx)
    be = OracleBackend()
    dr = D.DistributedReinit(be, b, dx, h)
    out, nsw, rms = dr.run(be.from_numpy(phi_np), sweeps - 1, tol=0.0)
    own = tuple(slice(lo, hi) for lo, hi in b.own_local)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), own=np.array(b.own), data=be.to_numpy(out, b.ext)[own], nsw=nsw,
             rms=np.array(rms))
    dist.destroy_process_group()


@pytest.mark.parametrize("dims", [(2, 1, 1), (1, 2, 1), (1, 1, 2), (2, 2, 1)])
def test_decomposed_jacobi_equals_single_domain(oracle, tmp_path, dims):
    from levelsetfortran_amd import fields

    npts, sweeps = (30, 26, 24), 5
    world = int(np.prod(dims))
    mp.spawn(_worker, args=(world, _free_port(), dims, npts, sweeps, str(tmp_path)), nprocs=world, join=True)
    phi0, dx = fields.two_sphere_phi0(npts)
    nx, ny, nz = (v - 1 for v in npts)
    ref = phi0.copy(order="F")
    rc, n, tr = oracle.reinit(ref, nx, ny, nz, sweeps - 1, dx, fields.reinit_step(dx), tol=0.0, order=oracle.JACOBI)
    got = np.full_like(ref, np.nan)
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        sl = tuple(slice(int(s), int(e)) for s, e in z["own"])
        got[sl] = z["data"]
        assert int(z["nsw"]) == sweeps
        assert np.allclose(z["rms"], tr, rtol=1e-12, atol=0)  # same sum, different association
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("check_every", [1, 3, 8])
def test_distributed_stop_test(oracle, tmp_path, check_every):
    """run() follows subs.f90:915: it stops after the first sweep whose global RMS is < tol -- whatever the number of sweeps
    between two looks at the RMS (one all_reduce and one host read per window; a window that holds the stop sweep is repeated
    from its kept start): same sweep count, same field, same trace as the single-domain sweep."""
    from levelsetfortran_amd import fields

    npts = (30, 26, 24)
    phi0, dx = fields.two_sphere_phi0(npts)
    ref = phi0.copy(order="F")
    nx, ny, nz = (v - 1 for v in npts)
    rc, n_ref, tr = oracle.reinit(ref, nx, ny, nz, 50, dx, fields.reinit_step(dx), tol=4.55e-3, order=oracle.JACOBI)
    assert 1 < n_ref < 50
    mp.spawn(_worker_tol, args=(2, _free_port(), (2, 1, 1), npts, 4.55e-3, str(tmp_path), check_every), nprocs=2, join=True)
    got = np.full_like(ref, np.nan)
    for r in range(2):
        z = np.load(tmp_path / f"t{r}.npz")
        assert int(z["nsw"]) == n_ref
        assert np.allclose(z["rms"], tr[:n_ref], rtol=1e-12, atol=0)
        sl = tuple(slice(int(s), int(e)) for s, e in z["own"])
        got[sl] = z["data"]
    assert np.array_equal(got, ref)


def _worker_tol(rank, world, port, dims, npts, tol, outdir, check_every):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from levelsetfortran_amd import distributed as D, fields
    from oracle_backend import OracleBackend

    n = tuple(v - 1 for v in npts)
    b = D.make_block(rank, dims, n)
    rng = tuple((g, g + e) for g, e in zip(b.g0, b.ext))
    phi_np, dx = fields.two_sphere_phi0(npts, ranges=rng)
    be = OracleBackend()
    dr = D.DistributedReinit(be, b, dx, fields.reinit_step(dx))
    out, nsw, rms = dr.run(be.from_numpy(phi_np), 50, tol=tol, check_every=check_every)
    own = tuple(slice(lo, hi) for lo, hi in b.own_local)
    np.savez(os.path.join(outdir, f"t{rank}.npz"), nsw=nsw, own=np.array(b.own), data=be.to_numpy(out, b.ext)[own], rms=np.array(rms))
    dist.destroy_process_group()
