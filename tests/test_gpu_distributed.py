"""Two / four ranks sharing the one GPU of the test box (gloo transport through pinned host buffers, since RCCL
refuses two ranks on one device):
the REAL HIP backend -- lsf_jacobi_sweep_box / lsf_bc_box / lsf_pack_box / lsf_unpack_box on a comm and a compute
stream -- driven by the real decomposition loop must reproduce the single-domain Jacobi sweep bit for bit."""
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
    be = D.HipBackend(dev, arith="strict", host_staging=True)  # gloo cannot read HBM
    dr = D.DistributedReinit(be, b, dx, h)
    out, nsw, rms = dr.run(be.from_numpy(phi_np), sweeps - 1, tol=0.0)
    own = tuple(slice(lo, hi) for lo, hi in b.own_local)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), own=np.array(b.own), data=be.to_numpy(out, b.ext)[own], nsw=nsw,
             rms=np.array(rms))
    dist.destroy_process_group()


@pytest.mark.parametrize("dims", [(2, 1, 1), (1, 1, 2), (2, 2, 1)])
def test_two_ranks_one_gpu_equal_single_domain(tmp_path, dims):
    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import fields

    npts, sweeps = (48, 40, 44), 6
    world = int(np.prod(dims))
    mp.spawn(_worker, args=(world, _free_port(), dims, npts, sweeps, str(tmp_path)), nprocs=world, join=True)
    phi0, dx = fields.two_sphere_phi0(npts)
    nx, ny, nz = (v - 1 for v in npts)
    ref = phi0.copy(order="F")
    rep = lsf.reinit(ref, None, None, nx, ny, nz, sweeps - 1, dx, fields.reinit_step(dx), tol=0.0, order="jacobi",
                     arith="strict")
    got = np.full_like(ref, np.nan)
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        sl = tuple(slice(int(s), int(e)) for s, e in z["own"])
        got[sl] = z["data"]
        assert int(z["nsw"]) == sweeps
        assert np.allclose(z["rms"], rep.rms, rtol=1e-11, atol=0)
    assert np.array_equal(got, ref)


def test_device_field_block_is_a_slice_of_the_whole_field():
    """bench.py builds every rank's block of the synthetic phi0 directly in HBM (fields.two_sphere_phi0_device with
    ranges): it must be the corresponding slice of the single-domain field."""
    from levelsetfortran_amd import fields

    npts = (40, 33, 27)
    dev = torch.device("cuda", 0)
    whole, dx = fields.two_sphere_phi0_device(npts, dev)
    whole = whole.reshape(npts[2], npts[1], npts[0])
    rng = ((5, 31), (0, 20), (9, 27))
    part, dx2 = fields.two_sphere_phi0_device(npts, dev, ranges=rng)
    assert dx == dx2
    part = part.reshape(rng[2][1] - rng[2][0], rng[1][1] - rng[1][0], rng[0][1] - rng[0][0])
    assert torch.equal(part, whole[rng[2][0]:rng[2][1], rng[1][0]:rng[1][1], rng[0][0]:rng[0][1]])
    host, _ = fields.two_sphere_phi0(npts)
    assert np.allclose(whole.cpu().numpy().transpose(2, 1, 0), host, rtol=0, atol=1e-14)


def test_bench_multi_rank_control_flow_on_one_gpu():
    """bench.py --gpus 2 under torch.distributed.run, both ranks sharing this box's GPU over gloo (LSF_BENCH_SHARED_GPU:
    RCCL refuses two ranks on one device).  Not a measurement: it checks that the N > 1 path prints exactly one JSON
    line with the whole-job aggregate and the block-decomposed secondary measurement."""
    import json
    import subprocess

    env = dict(os.environ, LSF_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "96", "--steps", "4",
           "--warmup", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    ent = d["decomposed"]["entries"]
    weak = [e for e in ent if e["scaling"] == "weak"]
    # N > 1: the headline is the path that communicates (weak-scaling decomposed Jacobi sweep), stated in metric and config; the
    # replicas of the exact ordering are an entry of their own, and so is the one-GPU rate of the headline's own path
    assert d["headline_path"] == "decomposed-jacobi-weak" and "block-decomposed Jacobi" in d["metric"] and "Jacobi" in d["config"]["ordering"]
    assert d["value"] == weak[0]["value"] and d["ms_per_step"] == weak[0]["ms_per_step"] and d["config"]["grid"] == [96, 96, 192]
    cells = 4 * 94.0 * 94.0 * 190.0
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * 4 - cells) < 1e-6 * cells  # value = cells of the whole job / time
    rep = d["replicas_gs"]
    cells_rep = 2 * 4 * 94.0 ** 3
    assert abs(rep["value"] * rep["ms_per_step"] * 1e-3 * 4 - cells_rep) < 1e-6 * cells_rep and "replicas" in rep["note"]
    assert d["same_path_one_gpu"]["value"] > 0
    # what was computed from the replicas' timing sits with the replicas; the headline's VALU roofline is its own:
    # per GPU, from the decomposed Jacobi value, with the Jacobi ordering's operation count
    v = d["roofline_fp64_valu"]
    assert v["useful_ops_per_cell"] == 259.0 and abs(v["achieved"] - 505.0 * d["value"] / 2 / 1e12) < 1e-9 * v["achieved"]
    assert rep["roofline_fp64_valu"]["useful_ops_per_cell"] == 313.0 and "step_breakdown_ms" in rep and "step_breakdown_ms" not in d
    strong = [e for e in ent if e["scaling"] == "strong"]
    assert len(weak) == 1 and weak[0]["value"] > 0 and weak[0]["global_grid"] == [96, 96, 192] and weak[0]["dims"] == [1, 1, 2]
    # fixed global grid split over the ranks: non-cubic local blocks (48 owned + 3 ghost points along z, the axis two
    # ranks cut: the unit-stride axis x is cut last)
    assert len(strong) == 1 and strong[0]["global_grid"] == [96, 96, 96] and strong[0]["local_block"] == [96, 96, 51]
    assert strong[0]["value"] > 0 and "error" not in d["decomposed"]
    # every decomposed entry states its share of the job's HBM roofline and what moved its halos (gloo in this rehearsal:
    # no RCCL rank; on a node with a GPU per rank rccl_ranks = the world size as the RCCL backend reports it)
    for e in weak + strong:
        assert 0 < e["roofline"]["frac"] < 1 and e["roofline"]["peak"] == 2 * 8000.0 and e["rccl_ranks"] == 0 and e["transport"] == "gloo"
        # every decomposed entry carries the in-run evidence that its path IS the single-domain sweep (same ranks,
        # transport and decomposition on a 64^3 grid here, 256^3 in a real job: field by SHA-256 per owned block, RMS trace)
        par = e["parity"]
        assert par["ok"] is True and par["field_sha_equal"] is True and par["rms_trace_equal"] is True and par["blocks_differing"] == []
        assert par["grid"] == [64, 64, 64] and par["ranks"] == 2 and par["dims"] == [1, 1, 2] and par["sweeps"] == 8
    assert d["rccl_ranks"] == 0
    assert d["parity"]["ok"] is True and "parity_failed" not in d["decomposed"]  # the headline is the weak entry: its record on top


def test_bench_parity_mismatch_fails_the_job_after_the_line():
    """a decomposed field that is NOT the single-domain field (one owned value of the last rank moved by one unit in the last place:
    LSF_BENCH_PARITY_SABOTAGE) must show in the record of every entry of the path and end the job non-zero -- after the line"""
    import json
    import subprocess

    env = dict(os.environ, LSF_BENCH_SHARED_GPU="1", LSF_BENCH_PARITY_SABOTAGE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "96", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    par = d["parity"]
    assert par["ok"] is False and par["field_sha_equal"] is False and par["rms_trace_equal"] is True and par["blocks_differing"] == [1]
    assert d["decomposed"]["parity_failed"]


def test_bench_strong_scaling_headline_on_one_gpu():
    """bench.py --mode jacobi --global G: the decomposed sweep on a FIXED global grid is the headline ("strong"); BASELINE
    configuration 4 is `--gpus 4 --global 1024`, rehearsed here as 4 ranks sharing the GPU on a 64^3 grid (2x2x1, x uncut)."""
    import json
    import subprocess

    env = dict(os.environ, LSF_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--mode", "jacobi", "--global", "64",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["config"]["grid"] == [64, 64, 64]
    cells = 3 * 62.0 ** 3
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * 3 - cells) < 1e-6 * cells
    assert d["roofline"]["peak"] == 4 * 8000.0 and abs(d["roofline"]["achieved"] - d["value"] * 24.0 / 1e9) < 1e-6 * d["roofline"]["achieved"]
    assert d["rccl_ranks"] == 0  # gloo rehearsal
    par = d["parity"]  # the headline's own evidence: 4 ranks, 1 x 2 x 2, against rank 0's single-domain sweep
    assert par["ok"] is True and par["field_sha_equal"] is True and par["rms_trace_equal"] is True
    assert par["ranks"] == 4 and par["dims"] == [1, 2, 2] and par["grid"] == [64, 64, 64]


def test_sumsq_bracket_equals_per_call_reduction():
    """lsf_sumsq_begin / lsf_sumsq_end (include/lsf.h): the bracketed box calls leave the same field and, to rounding,
    the same sum of squares as the calls reduced one by one; misuse is refused."""
    import ctypes

    from levelsetfortran_amd import _lib, distributed as D, fields

    npts = (70, 41, 37)
    n = tuple(v - 1 for v in npts)
    b = D.make_block(0, (1, 1, 1), n)
    dev = torch.device("cuda", 0)
    be = D.HipBackend(dev, arith="strict")
    phi_np, dx = fields.two_sphere_phi0(npts)
    h = fields.reinit_step(dx)
    a = be.from_numpy(phi_np)
    regions = [[(1, 30), (1, 40), (1, 36)], [(30, 69), (1, 40), (1, 36)]]  # two halves of the interior
    outs, sums = [], []
    for bracket in (False, True):
        out = a.clone()
        ss = be.zeros(1)
        if bracket:
            be.sumsq_begin(be.compute)
        for r in regions:
            be.sweep(a, out, a, b, r, dx, h, ss, be.compute)
        be.bc(a, out, b, [(0, npts[0]), (0, npts[1]), (0, npts[2])], dx, ss, be.compute)
        if bracket:
            be.sumsq_end(be.compute)
        be.synchronize()
        outs.append(out.cpu().numpy())
        sums.append(float(ss.item()))
    assert np.array_equal(outs[0], outs[1])
    assert sums[0] > 0 and abs(sums[0] - sums[1]) <= 1e-13 * sums[0]
    lib = _lib.load()
    st = be.compute.cuda_stream
    assert lib.lsf_sumsq_end(st) == _lib.LSF_ERR_INVALID
    assert lib.lsf_sumsq_begin(st) == _lib.LSF_OK and lib.lsf_sumsq_begin(st) == _lib.LSF_ERR_INVALID
    # the per-stream buffer of partial sums can be sized up front (then the box calls neither allocate nor synchronise) --
    # but not inside a bracket
    assert lib.lsf_box_reserve(st, 1 << 18) == _lib.LSF_ERR_INVALID
    assert lib.lsf_sumsq_end(st) == _lib.LSF_OK
    assert lib.lsf_box_reserve(st, 1 << 18) == _lib.LSF_OK
