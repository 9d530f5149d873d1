"""lsf_reinit_multi / lsf_multi_* (include/lsf.h): the block-decomposed Jacobi sweep driven by ONE process -- one host
thread, compute stream and communication stream per block, 3-cell face halos by peer copies, RMS judged one sweep
late.  A one-GPU box names its device several times: every block, halo message, event and reduction of the 8-GPU
schedule is there, only the copies stay on the card.  The result must be the single-domain Jacobi sweep bit for bit."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lsf():
    import torch

    assert torch.cuda.is_available()
    import levelsetfortran_amd

    return levelsetfortran_amd


def _single(lsf, phi0, n, iters, dx, h, tol, arith):
    a = phi0.copy(order="F")
    rep = lsf.reinit(a, None, None, n[0], n[1], n[2], iters, dx, h, tol=tol, order="jacobi", arith=arith)
    return a, rep


@pytest.mark.parametrize("npts,dims", [((70, 45, 52), (2, 2, 2)), ((70, 45, 52), (2, 2, 1)), ((41, 90, 33), (1, 4, 1)),
                                       ((30, 31, 64), (1, 1, 3)), ((140, 20, 22), (4, 1, 2)), ((36, 36, 36), (1, 1, 1))])
@pytest.mark.parametrize("arith", ["strict", "fast"])
@pytest.mark.parametrize("small", ["192", "0"])
def test_multi_equals_single_domain_bitwise(lsf, npts, dims, arith, small, monkeypatch):
    """small = 192 (default): blocks this size run a sweep as ONE launch over all owned cells after the exchange; 0: core beside the
    exchange, then the rims (what blocks of 192 points and more per axis do).  The face slabs travel in one pack and one unpack
    launch either way (lsf_pack_boxes).  Both equal the single-domain sweep bit for bit."""
    from levelsetfortran_amd import fields

    monkeypatch.setenv("LSF_MULTI_SMALL", small)
    phi0, dx = fields.two_sphere_phi0(npts)
    n = tuple(v - 1 for v in npts)
    h = fields.reinit_step(dx)
    want, rep1 = _single(lsf, phi0, n, 6, dx, h, 0.0, arith)
    got = phi0.copy(order="F")
    nblk = dims[0] * dims[1] * dims[2]
    rep = lsf.reinit_multi(got, n[0], n[1], n[2], 6, dx, h, [0] * nblk, dims=dims, tol=0.0, arith=arith)
    assert rep.count == rep1.count == 7
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    assert np.allclose(rep.rms, rep1.rms, rtol=1e-12, atol=0)  # block sums added in rank order vs one fixed-order sum


def test_multi_stop_sweep_and_default_dims(lsf):
    """run to a tolerance: same stop sweep as the single domain although the RMS is judged one sweep late (the extra
    sweep writes the other buffer); default decomposition for 8 blocks is 2x2x2 (BASELINE configuration 5)."""
    from levelsetfortran_amd import fields

    npts = (64, 60, 56)
    phi0, dx = fields.two_sphere_phi0(npts)
    n = tuple(v - 1 for v in npts)
    h = fields.reinit_step(dx)
    _, ref = _single(lsf, phi0, n, 40, dx, h, 0.0, "fast")
    tr = np.array(ref.rms)
    k = int(np.argmin(tr[:30]))
    tol = 0.5 * (tr[k] + tr[:k].min()) if k > 0 else 2 * tr[0]
    want, rep1 = _single(lsf, phi0, n, 40, dx, h, tol, "fast")
    got = phi0.copy(order="F")
    rep = lsf.reinit_multi(got, n[0], n[1], n[2], 40, dx, h, [0] * 8, tol=tol, arith="fast")
    assert rep.converged and rep1.converged and rep.count == rep1.count == k + 1
    assert np.array_equal(got, want)


@pytest.mark.parametrize("transport", ["peer", "mock"])
@pytest.mark.parametrize("check_every", [1, 3, 8, 64])
def test_multi_judging_window_and_transport_do_not_change_the_result(lsf, transport, check_every):
    """The host looks at the RMS once per window of check_every sweeps, one window late; a window that holds the stop sweep is
    repeated from its kept start.  Stop sweep, field and trace must be those of the single-domain sweep whatever the window
    (stop sweep in the first / a middle / the last window, at the start / inside / at the end of a window) and whichever
    transport moves the halos: peer copies pushed by the sender, or the RCCL schedule (one group call per sweep, no
    events between blocks) carried out by the stand-in transport."""
    from levelsetfortran_amd import fields

    npts = (48, 44, 40)
    phi0, dx = fields.two_sphere_phi0(npts)
    n = tuple(v - 1 for v in npts)
    h = fields.reinit_step(dx)
    _, ref = _single(lsf, phi0, n, 40, dx, h, 0.0, "fast")
    tr = np.array(ref.rms)
    # a sweep can be made the stop sweep by the tolerance if its RMS is lower than every earlier one
    recs = [k for k in range(len(tr)) if k == 0 or tr[k] < tr[:k].min()]
    assert len(recs) >= 8 and recs[-1] >= 12, recs
    for k in sorted({recs[0], recs[2], recs[4], recs[len(recs) // 2], recs[-3], recs[-2], recs[-1]}):
        tol = 0.5 * (tr[k] + tr[:k].min()) if k > 0 else 2 * tr[0]
        want, rep1 = _single(lsf, phi0, n, 40, dx, h, tol, "fast")
        got = phi0.copy(order="F")
        rep = lsf.reinit_multi(got, n[0], n[1], n[2], 40, dx, h, [0] * 4, dims=(1, 2, 2), tol=tol, arith="fast",
                               check_every=check_every, transport=transport)
        assert rep.converged and rep.count == rep1.count == k + 1, (k, rep.count)
        assert np.array_equal(got, want), k
        assert np.allclose(rep.rms, rep1.rms, rtol=1e-12, atol=0)
    # no stop: the iteration cap ends the run in the middle of a window
    want, rep1 = _single(lsf, phi0, n, 10, dx, h, 0.0, "strict")
    got = phi0.copy(order="F")
    rep = lsf.reinit_multi(got, n[0], n[1], n[2], 10, dx, h, [0] * 8, tol=0.0, arith="strict", check_every=check_every, transport=transport)
    assert rep.count == 11 and np.array_equal(got, want)


def test_multi_rccl_transport_refusals_and_single_rank(lsf):
    """The RCCL transport (ncclCommInitAll over the device list, ncclSend / ncclRecv per neighbour) wants a device per
    block: on a one-GPU box it can be configured for one block only (communicator created and destroyed, no message) and
    is refused with several blocks on one device; what its schedule does is covered by the stand-in transport above."""
    from levelsetfortran_amd import _lib, fields

    lib = _lib.load()
    npts = (40, 36, 32)
    phi0, dx = fields.two_sphere_phi0(npts)
    n = tuple(v - 1 for v in npts)
    h = fields.reinit_step(dx)
    M = ctypes.c_void_p()
    devs = (ctypes.c_int * 2)(0, 0)
    _lib.check(lib.lsf_multi_create(n[0], n[1], n[2], devs, 2, None, 0, ctypes.byref(M)))
    try:
        assert lib.lsf_multi_configure(M, 8, _lib.LSF_TRANSPORT_RCCL) == _lib.LSF_ERR_INVALID
        assert b"distinct device" in lib.lsf_last_error()
        assert lib.lsf_multi_configure(M, 0, _lib.LSF_TRANSPORT_PEER) == _lib.LSF_ERR_INVALID
        assert lib.lsf_multi_configure(M, 8, 7) == _lib.LSF_ERR_INVALID
        ce, tp, rr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(-1)
        _lib.check(lib.lsf_multi_info(M, ctypes.byref(ce), ctypes.byref(tp), ctypes.byref(rr), None, None, None, None, None))
        assert (ce.value, tp.value, rr.value) == (8, _lib.LSF_TRANSPORT_PEER, 0)
    finally:
        _lib.check(lib.lsf_multi_destroy(M))
    want, rep1 = _single(lsf, phi0, n, 5, dx, h, 0.0, "fast")
    one = (ctypes.c_int * 1)(0)
    _lib.check(lib.lsf_multi_create(n[0], n[1], n[2], one, 1, None, 0, ctypes.byref(M)))
    try:
        _lib.check(lib.lsf_multi_configure(M, 4, _lib.LSF_TRANSPORT_RCCL))
        rr, ver = ctypes.c_int(-1), ctypes.c_int(0)
        _lib.check(lib.lsf_multi_info(M, None, None, ctypes.byref(rr), ctypes.byref(ver), None, None, None, None))
        assert rr.value == 1 and ver.value > 0
        got = phi0.copy(order="F")
        _lib.check(lib.lsf_multi_scatter(M, got.ctypes.data))
        done = ctypes.c_int(0)
        _lib.check(lib.lsf_multi_run(M, 5, dx, h, 0.0, _lib.LSF_ORDER_JACOBI | _lib.LSF_ARITH_FAST, ctypes.byref(done), None, 0))
        _lib.check(lib.lsf_multi_gather(M, got.ctypes.data))
        assert done.value == 6 and np.array_equal(got, want)
    finally:
        _lib.check(lib.lsf_multi_destroy(M))


def test_multi_host_enqueues_faster_than_the_device_executes(lsf):
    """Eight blocks of 128^3 (north_star's 256^3 on eight GPUs) sharing the test GPU, 32 sweeps in one judging window: the
    host threads never wait for the device inside a window, so their enqueue time per sweep must stay below what the device
    needs per sweep -- otherwise the driver, not the GPU, would set the pace."""
    from levelsetfortran_amd import _lib, fields

    lib = _lib.load()
    npts = (256, 256, 256)
    phi0, dx = fields.two_sphere_phi0(npts)
    n = tuple(v - 1 for v in npts)
    h = fields.reinit_step(dx)
    devs = (ctypes.c_int * 8)(*([0] * 8))
    M = ctypes.c_void_p()
    _lib.check(lib.lsf_multi_create(n[0], n[1], n[2], devs, 8, None, 0, ctypes.byref(M)))
    try:
        _lib.check(lib.lsf_multi_configure(M, 32, _lib.LSF_TRANSPORT_PEER))
        _lib.check(lib.lsf_multi_scatter(M, phi0.ctypes.data))
        done = ctypes.c_int(0)
        mode = _lib.LSF_ORDER_JACOBI | _lib.LSF_ARITH_FAST
        _lib.check(lib.lsf_multi_run(M, 7, dx, h, 0.0, mode, ctypes.byref(done), None, 0))  # warm-up: workspaces, plans
        _lib.check(lib.lsf_multi_run(M, 31, dx, h, 0.0, mode, ctypes.byref(done), None, 0))
        he, hc, wall, nsw = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        _lib.check(lib.lsf_multi_info(M, None, None, None, None, ctypes.byref(he), ctypes.byref(hc), ctypes.byref(wall), ctypes.byref(nsw)))
        assert done.value == 32 and nsw.value == 32
        host_us, calls_us, dev_us = he.value / 32 * 1e6, hc.value / 32 * 1e6, wall.value / 32 * 1e6
        print(f"8 blocks of 128^3 on one device, per sweep: slowest host thread {host_us:.0f} us enqueuing (of which {calls_us:.0f} us "
              f"inside its own calls, the rest waiting for the device's enqueue lock and its neighbours' threads), run {dev_us:.0f} us")
        # all eight threads together keep up with the one device ...
        assert host_us < dev_us, (host_us, dev_us)
        # ... and one thread alone with one eighth of it: what a node with a device per block asks of each thread
        assert calls_us < dev_us / 8 * 1.25, (calls_us, dev_us / 8)
    finally:
        _lib.check(lib.lsf_multi_destroy(M))


def test_multi_f32_equals_single_domain_f32(lsf):
    from levelsetfortran_amd import fields

    npts = (66, 48, 40)
    phi0, dx = fields.two_sphere_phi0(npts)
    phi0 = np.asfortranarray(phi0.astype(np.float32))
    n = tuple(v - 1 for v in npts)
    h = fields.reinit_step(dx)
    want = phi0.copy(order="F")
    rep1 = lsf.reinit(want, None, None, n[0], n[1], n[2], 5, dx, h, tol=0.0, order="jacobi", arith="fast")
    got = phi0.copy(order="F")
    rep = lsf.reinit_multi(got, n[0], n[1], n[2], 5, dx, h, [0] * 4, tol=0.0, arith="fast")
    assert rep.count == rep1.count == 6 and np.array_equal(got, want)


def test_multi_resident_pieces_and_refusals(lsf):
    """lsf_multi_create / scatter / run / run / gather: two runs continue each other (raster phase is irrelevant for
    the Jacobi ordering: 3 + 4 sweeps == 7 sweeps with the original sign field kept -- here the second run re-reads its
    sign field, so compare with two single-domain calls); the exact ordering and bad device lists are refused."""
    from levelsetfortran_amd import _lib, fields

    lib = _lib.load()
    npts = (50, 40, 44)
    phi0, dx = fields.two_sphere_phi0(npts)
    n = tuple(v - 1 for v in npts)
    h = fields.reinit_step(dx)
    want = phi0.copy(order="F")
    lsf.reinit(want, None, None, n[0], n[1], n[2], 2, dx, h, tol=0.0, order="jacobi", arith="strict")
    lsf.reinit(want, None, None, n[0], n[1], n[2], 3, dx, h, tol=0.0, order="jacobi", arith="strict")
    devs = (ctypes.c_int * 4)(0, 0, 0, 0)
    M = ctypes.c_void_p()
    _lib.check(lib.lsf_multi_create(n[0], n[1], n[2], devs, 4, None, 0, ctypes.byref(M)))
    try:
        g0, ext = (ctypes.c_int * 3)(), (ctypes.c_int * 3)()
        lo, hi = (ctypes.c_int * 3)(), (ctypes.c_int * 3)()
        dev = ctypes.c_int(-1)
        _lib.check(lib.lsf_multi_block(M, 3, g0, ext, lo, hi, ctypes.byref(dev)))
        # 4 blocks: 1 x 2 x 2 (x, the unit-stride axis, is cut last); block 3 = (0, 1, 1)
        assert dev.value == 0 and list(lo) == [0, 20, 22] and list(hi) == [50, 40, 44] and list(g0) == [0, 17, 19]
        got = phi0.copy(order="F")
        _lib.check(lib.lsf_multi_scatter(M, got.ctypes.data))
        done = ctypes.c_int(0)
        mode = _lib.LSF_ORDER_JACOBI | _lib.LSF_ARITH_STRICT
        _lib.check(lib.lsf_multi_run(M, 2, dx, h, 0.0, mode, ctypes.byref(done), None, 0))
        assert done.value == 3
        _lib.check(lib.lsf_multi_run(M, 3, dx, h, 0.0, mode, ctypes.byref(done), None, 0))
        assert done.value == 4
        _lib.check(lib.lsf_multi_gather(M, got.ctypes.data))
        assert np.array_equal(got, want)
        assert lib.lsf_multi_run(M, 1, dx, h, 0.0, _lib.LSF_ORDER_GS, ctypes.byref(done), None, 0) == _lib.LSF_ERR_INVALID
    finally:
        _lib.check(lib.lsf_multi_destroy(M))
    bad = (ctypes.c_int * 2)(0, 99)
    assert lib.lsf_multi_create(n[0], n[1], n[2], bad, 2, None, 0, ctypes.byref(M)) == _lib.LSF_ERR_NO_DEVICE
    three = (ctypes.c_int * 3)(2, 2, 2)
    assert lib.lsf_multi_create(n[0], n[1], n[2], devs, 4, three, 0, ctypes.byref(M)) == _lib.LSF_ERR_INVALID


def test_bench_single_process_entries_helper(lsf):
    """bench.py's rank-0 measurement of the lsf_multi driver (run on 1, 2, 4, ... N devices at N > 1) with one device."""
    import importlib.util
    import os

    from conftest import ROOT
    from levelsetfortran_amd import _lib

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ent = bench._single_process_entries(_lib.load(), 1, 64, 3, 1, "fast", transports=("peer", "rccl"))
    assert len(ent) == 1 and ent[0]["n_gpus"] == 1 and ent[0]["global_grid"] == [64, 64, 64] and ent[0]["value"] > 0
    assert ent[0]["transport"] == "peer" and ent[0]["rccl_ranks"] == 0 and 0 < ent[0]["roofline"]["frac"] < 1
    assert ent[0]["host_calls_ms_per_step"] > 0
    # the in-run parity record of the entry: lsf_reinit_multi on the entry's devices against lsf_reinit
    par = ent[0]["parity"]
    assert par["ok"] is True and par["field_sha_equal"] is True and par["rms_trace_equal"] is True and par["grid"] == [64, 64, 64]
    # ... and it can say no: four blocks on device 0 compared with a single-domain field of ANOTHER arithmetic
    bench._PARITY_SINGLE.clear()
    good = bench._parity_one_process([0, 0, 0, 0], 64, 4, "fast")
    assert good["ok"] is True and good["max_abs_field_diff"] == 0.0
    key = (64, 4, "fast", "jacobi")
    a, rms, phi0, dx = bench._PARITY_SINGLE[key]
    a[5, 6, 7] = np.nextafter(a[5, 6, 7], 2.0)
    badrec = bench._parity_one_process([0, 0], 64, 4, "fast")
    assert badrec["ok"] is False and badrec["field_sha_equal"] is False and badrec["rms_trace_equal"] is True
    bench._PARITY_SINGLE.clear()
