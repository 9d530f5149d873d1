"""lsf_reinit_multi with LSF_ORDER_GS (include/lsf.h; csrc/lsf_gs_slabs.hpp, k_reinit_gs_slab): the reference's in-place
ordering (subs.f90:743-852) executed by one dataflow launch per z slab, every slab on the same tile graph, cut planes,
tile flags, hyperplane counters and the stop verdict stored straight into the neighbour's memory.  A one-GPU box names its
device once per slab: every launch, peer store, flag and mirror of the multi-GPU run is there, only the stores stay on the
card.  The result must be lsf_reinit's -- hence, in STRICT arithmetic, the reference's -- bit for bit: field, sweep count,
RMS trace, stop sweep."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lsf():
    import torch

    assert torch.cuda.is_available()
    import levelsetfortran_amd

    return levelsetfortran_amd


@pytest.fixture()
def env():
    """environment switches of the library, restored afterwards"""
    saved = dict(os.environ)
    yield os.environ
    for k in set(os.environ) - set(saved):
        del os.environ[k]
    os.environ.update(saved)


def _field(npts):
    from levelsetfortran_amd import fields

    phi0, dx = fields.two_sphere_phi0(npts)
    return phi0, tuple(v - 1 for v in npts), dx, fields.reinit_step(dx)


def _both(lsf, npts, slabs, iters, arith, tol=0.0):
    phi0, n, dx, h = _field(npts)
    want = phi0.copy(order="F")
    r1 = lsf.reinit(want, None, None, *n, iters, dx, h, tol=tol, order="gs", arith=arith)
    got = phi0.copy(order="F")
    r = lsf.reinit_multi(got, *n, iters, dx, h, [0] * slabs, tol=tol, arith=arith, order="gs")
    return want, r1, got, r


# tile layers in z (8 planes each by default): 5, 9, 13, 12, 16 -- ragged last layers, slabs of one layer, a slab per layer
@pytest.mark.parametrize("npts,slabs,iters", [((60, 50, 40), 1, 5), ((60, 50, 40), 2, 9), ((96, 80, 72), 3, 20), ((71, 83, 97), 2, 12),
                                              ((33, 47, 98), 3, 10), ((40, 36, 26), 3, 17), ((128, 128, 128), 2, 70)])
@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_slabs_equal_the_single_device_field_bitwise(lsf, npts, slabs, iters, arith, env):
    if npts == (128, 128, 128):
        env["LSF_DF_BATCH"] = "32" if arith == "strict" else "64"  # three resp. two batches: control words are reset in between
    want, r1, got, r = _both(lsf, npts, slabs, iters, arith)
    assert r.count == r1.count == iters + 1
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    assert r.rms == r1.rms  # the same tile-column sums added in the same order by the same epilogue


def test_slabs_strict_is_the_oracle_bitwise(lsf, oracle):
    """not only equal to the single-device launch: equal to the CPU restatement of the reference (oracle/, bit-identical to the
    reference's own build) -- the parity claim of the sharded exact ordering stands on its own"""
    npts = (36, 30, 44)
    phi0, n, dx, h = _field(npts)
    want = phi0.copy(order="F")
    rc, cnt, trace = oracle.reinit(want, *n, 11, dx, h, tol=0.0)
    got = phi0.copy(order="F")
    r = lsf.reinit_multi(got, *n, 11, dx, h, [0, 0, 0], tol=0.0, arith="strict", order="gs")
    assert r.count == cnt == 12
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    assert np.allclose(trace[:cnt], r.rms, rtol=1e-12, atol=0)  # the reference adds cell by cell, the kernel tile column by tile column


@pytest.mark.parametrize("lo,hi,cap,hf", [(3, 30, 40, 1.0), (66, 100, 110, 0.1)])
def test_slabs_stop_at_the_reference_stop_sweep(lsf, env, lo, hi, cap, hf):
    """tol > 0: the sweep whose RMS falls below it is the last one on every slab, later sweeps already in flight are abandoned
    and the field returned is that sweep's -- in the first batch of the launch and (a tenth of the time step: the trace still
    falls after 64 sweeps) in the second"""
    npts = (64, 60, 56)
    phi0, n, dx, h = _field(npts)
    h *= hf
    env["LSF_DF_BATCH"] = "64"  # (calls of more than 64 sweeps would otherwise run 256 per launch)
    probe = phi0.copy(order="F")
    ref = lsf.reinit(probe, None, None, *n, cap, dx, h, tol=0.0, order="gs", arith="fast")
    tr = np.array(ref.rms)
    # a tolerance crossed for the first time at a record minimum of the trace
    rec = [i for i in range(lo, hi) if tr[i] < tr[:i].min()]
    assert rec, "the trace has no record minimum in the window"
    k = rec[len(rec) // 2]
    tol = 0.5 * (tr[k] + tr[:k].min())
    want = phi0.copy(order="F")
    r1 = lsf.reinit(want, None, None, *n, cap, dx, h, tol=tol, order="gs", arith="fast")
    got = phi0.copy(order="F")
    r = lsf.reinit_multi(got, *n, cap, dx, h, [0, 0, 0], tol=tol, arith="fast", order="gs")
    assert r1.count == k + 1 and r.count == k + 1
    assert r.converged and r.rms == r1.rms
    assert np.array_equal(got, want)


@pytest.mark.parametrize("switch", [{"LSF_GS_SKEW_W": "c1x2"}, {"LSF_GS_SKEW_W": "c1x4"}, {"LSF_GS_SKEW_W": "1x1"}, {"LSF_GS_SKEW_W": "4x2"},
                                    {"LSF_GS_MARCH": "x"}, {"LSF_SLAB_FINEGRAINED": "1"}, {"LSF_SLAB_GRID": "7"}])
def test_slabs_tile_shapes_march_axis_memory_kind_and_grid(lsf, env, switch):
    """every tile shape of the single launch (three lanes per cell / one lane per cell), the untransposed march, fine-grained
    allocations (what a multi-device run uses for everything a neighbour stores into) and a grid far below the device's
    capacity (the launch is a loop over tickets: any number of resident blocks must do)"""
    env.update(switch)
    want, r1, got, r = _both(lsf, (70, 66, 90), 2, 13, "strict")
    assert r.count == r1.count == 14
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    assert r.rms == r1.rms
    info = [ctypes.c_int(0) for _ in range(4)]
    ks = ctypes.c_double(0)
    from levelsetfortran_amd import _lib

    _lib.check(_lib.load().lsf_slabs_info(*[ctypes.byref(v) for v in info], ctypes.byref(ks)))
    assert info[0].value == 2 and info[3].value == 14 and ks.value > 0
    assert info[2].value == (1 if "LSF_SLAB_FINEGRAINED" in switch else 0)
    if "LSF_SLAB_GRID" in switch:
        assert info[1].value == 7


def test_slabs_refusals(lsf, env):
    from levelsetfortran_amd import _lib

    phi0, n, dx, h = _field((40, 40, 40))
    a = phi0.copy(order="F")
    with pytest.raises(_lib.LsfError, match="fewer tile layers"):
        lsf.reinit_multi(a, *n, 3, dx, h, [0] * 6, tol=0.0, arith="fast", order="gs")  # 5 layers of 8 planes
    with pytest.raises(_lib.LsfError, match="z slabs"):
        lsf.reinit_multi(a, *n, 3, dx, h, [0, 0], dims=(2, 1, 1), tol=0.0, arith="fast", order="gs")
    with pytest.raises(_lib.LsfError, match="GPU_MAX_HW_QUEUES"):
        lsf.reinit_multi(a, *n, 3, dx, h, [0] * 4, tol=0.0, arith="fast", order="gs")
    f = phi0.astype(np.float32, order="F")
    with pytest.raises(_lib.LsfError, match="single precision"):
        lsf.reinit_multi(f, *n, 3, dx, h, [0, 0], tol=0.0, arith="fast", order="gs")
    assert np.array_equal(a, phi0)  # refused calls leave the field alone


def test_slabs_time_out_is_an_error_not_a_hang(lsf, env):
    """a tile that cannot get its predecessor gives up after the time-out and tells every slab: with a bound of 0 ticks the very
    first wait fails -- the call returns LSF_ERR_HIP, the process lives, the next call works"""
    from levelsetfortran_amd import _lib

    phi0, n, dx, h = _field((60, 50, 40))
    a = phi0.copy(order="F")
    env["LSF_GS_TIMEOUT_TICKS"] = "0"
    with pytest.raises(_lib.LsfError, match="time-out"):
        lsf.reinit_multi(a, *n, 6, dx, h, [0, 0], tol=0.0, arith="fast", order="gs")
    del env["LSF_GS_TIMEOUT_TICKS"]
    want, r1, got, r = _both(lsf, (60, 50, 40), 2, 6, "fast")
    assert np.array_equal(got, want) and r.rms == r1.rms


def test_bench_slab_entries_helper_and_child(lsf):
    """bench.py's measurement of the sharded exact ordering (rank 0 of an N > 1 job runs it on 1, 2, 4 ... N devices, in a child
    process): here with device 0 named once per slab, in process and through the child"""
    import importlib.util

    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ent = bench._slab_entries(2, 96, 5, 2, "fast", devices_of=lambda nd: [0] * nd)
    assert [e["n_gpus"] for e in ent] == [1, 2]
    for e in ent:
        assert e.get("error") is None and e["value"] > 0 and e["ordering"] == "gs" and e["equal_to_first_entry"]
        assert 0 < e["roofline"]["frac"] < 1 and e["roofline"]["peak"] == 8000.0  # one physical device
        assert e["blocks_per_slab"] > 0 and e["call_s"] > e["ms_per_step"] * 5e-3
        par = e["parity"]  # in-run evidence: the slabs' field and RMS trace ARE lsf_reinit's (exact ordering: equal, no tolerance)
        assert par["ok"] is True and par["field_sha_equal"] is True and par["rms_trace_equal"] is True and par["rms_trace_rtol"] == 0.0
        assert par["ordering"] == "gs" and par["grid"] == [64, 64, 64] and par["sweeps"] == 16
    child = bench._slab_entries_in_a_child(1, 64, 3, 1, "strict")
    assert len(child) == 1 and child[0].get("error") is None and child[0]["n_gpus"] == 1 and child[0]["arith"] == "strict"


@pytest.mark.parametrize("n,slabs", [(256, 3), (512, 2)])
def test_slabs_at_baseline_sizes_equal_the_reference_itself(lsf, n, slabs):
    """BASELINE sizes: the sharded exact ordering against the reference's OWN reinit (tests/golden/synth_big.npz, made by
    tests/golden/make_golden_big.py from the amdflang build of subs.f90): SHA-256 of the whole field after 8 sweeps at 256^3
    (three slabs) and 2 sweeps at 512^3 (two slabs; the one-lane-per-cell tile, selected by size)."""
    import hashlib

    from conftest import GOLDEN
    from levelsetfortran_amd import fields

    path = os.path.join(GOLDEN, "synth_big.npz")
    if not os.path.exists(path):
        pytest.skip("synth_big.npz not generated")
    g = np.load(path)
    sweeps = int(g[f"n{n}_sweeps"])
    phi, dx = fields.two_sphere_phi0((n, n, n))
    h = fields.reinit_step(dx)
    rep = lsf.reinit_multi(phi, n - 1, n - 1, n - 1, sweeps - 1, dx, h, [0] * slabs, tol=0.0, arith="strict", order="gs")
    assert rep.count == sweeps
    assert hashlib.sha256(phi.reshape(-1, order="F").tobytes()).hexdigest() == str(g[f"n{n}_sha"])
    assert np.allclose(rep.rms, g[f"n{n}_rms"], rtol=1e-7, atol=0)


def test_one_slab_both_launch_forms(lsf, env):
    """a slab that has its device to itself runs one block per tile (what a node runs), slabs that share a device a loop over
    tickets: with one slab both forms can be asked for"""
    import ctypes

    from levelsetfortran_amd import _lib

    grids = {}
    for loop in ("0", "1"):
        env["LSF_SLAB_LOOP"] = loop
        want, r1, got, r = _both(lsf, (70, 66, 90), 1, 13, "fast")
        assert np.array_equal(got, want) and r.rms == r1.rms
        g = ctypes.c_int(0)
        _lib.check(_lib.load().lsf_slabs_info(None, ctypes.byref(g), None, None, None))
        grids[loop] = g.value
    assert grids["0"] > grids["1"] > 0  # every tile of the batch against the blocks the device holds at once


def test_slabs_single_sweep(lsf):
    """iter = 0: one sweep, one batch of one"""
    want, r1, got, r = _both(lsf, (40, 36, 26), 2, 0, "strict")
    assert r.count == r1.count == 1 and np.array_equal(got, want) and r.rms == r1.rms


@pytest.mark.parametrize("arith", ["fast", "strict"])
def test_slabs_counters_start_at_a_slabs_first_hyperplane(lsf, env, arith):
    """A slab's hyperplane counter counts its leading complete hyperplanes; those in front of its first tile hold nothing and are
    complete from the start.  Found by the round-4 soak (profiles/r04_soak.txt): 154 x 186 x 239 points, single-wavefront tiles
    (60 tile layers in z, 108 tiles per hyperplane), two slabs on one device -- the upper slab's tiles of sweep 2 waited for the
    lower slab's counter of sweep 1, which stayed at 0 until that slab's first tile (hyperplane 37) was done, and there were more
    of them in front of it in the list than the ticket loop has blocks: a time-out.  (On slabs with a device each the same wait
    was a stall, not a deadlock.)"""
    env["LSF_GS_SKEW_W"] = "c1x1"
    want, r1, got, r = _both(lsf, (154, 186, 239), 2, 5, arith)
    assert np.array_equal(got, want) and r.rms == r1.rms and r.count == r1.count == 6
