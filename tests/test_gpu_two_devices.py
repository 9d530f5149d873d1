"""First contact with a second GPU (VERDICT r5 "What's missing" 2-3): everything the multi-GPU paths do between DISTINCT devices --
peer copies, RCCL sends and receives (inside the C ABI and under torch.distributed), peer stores and remote flag polling of the
exact ordering's z slabs -- against the single-domain field, bit for bit.  The pool this suite has run on so far leases one GPU per
box: there every test here is SKIPPED with that reason (the same code paths run with device 0 named several times in
test_gpu_multi.py / test_gpu_slabs.py / test_gpu_distributed.py -- every launch, message and flag, only never between two
devices).  On a box with >= 2 GPUs they run, and a failure here is the first real evidence about those paths.
The reference is serial (README.md:17): the call site all of this stands in for is set3d.f90:308."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, sha

NDEV = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(NDEV < 2, reason=f"needs >= 2 GPUs, this box shows {NDEV}: the distinct-device paths stay unexercised here")]


@pytest.fixture(scope="module")
def lsf():
    import levelsetfortran_amd

    return levelsetfortran_amd


def _field(npts, dtype=np.float64):
    from levelsetfortran_amd import fields

    phi0, dx = fields.two_sphere_phi0(npts)
    return np.asfortranarray(phi0.astype(dtype)), tuple(v - 1 for v in npts), dx, fields.reinit_step(dx)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_peer_selftest_between_two_devices(lsf):
    """the litmus of what the slab launches assume of memory shared by two devices (include/lsf.h: lsf_peer_selftest): message
    passing in both directions at system scope, an atomic-max contest, both kernels resident at once"""
    assert lsf.peer_selftest(0, 1) == 0
    assert lsf.peer_selftest(1, 0) == 0


@pytest.mark.parametrize("transport", ["peer", "rccl"])
@pytest.mark.parametrize("arith", ["strict", "fast"])
@pytest.mark.parametrize("dims", [(1, 1, 2), (2, 1, 1)])
def test_multi_on_two_devices_equals_single_domain(lsf, transport, arith, dims):
    """lsf_reinit_multi with a block per DEVICE: hipMemcpyPeerAsync resp. ncclSend / ncclRecv really cross xGMI; field (SHA-256),
    sweep count and trace are the single-domain Jacobi sweep's -- with a cut across z (whole rows) and across x (3-cell rims)"""
    phi0, n, dx, h = _field((96, 80, 88))
    want = phi0.copy(order="F")
    r1 = lsf.reinit(want, None, None, *n, 11, dx, h, tol=0.0, order="jacobi", arith=arith)
    got = phi0.copy(order="F")
    r = lsf.reinit_multi(got, *n, 11, dx, h, [0, 1], dims=dims, tol=0.0, arith=arith, transport=transport, check_every=4)
    assert r.count == r1.count == 12
    assert sha(got) == sha(want), float(np.abs(got - want).max())
    assert np.allclose(r.rms, r1.rms, rtol=1e-11, atol=0)


def test_multi_on_every_device_of_the_box(lsf):
    """all the GPUs this box shows (2, 4, 8: the library's default decomposition for that count), both transports, run to a
    tolerance: same stop sweep and field as the single domain"""
    nd = 8 if NDEV >= 8 else (4 if NDEV >= 4 else 2)
    phi0, n, dx, h = _field((128, 120, 112))
    probe = phi0.copy(order="F")
    ref = lsf.reinit(probe, None, None, *n, 30, dx, h, tol=0.0, order="jacobi", arith="fast")
    tr = np.array(ref.rms)
    k = max(i for i in range(4, 28) if tr[i] < tr[:i].min())
    tol = 0.5 * (tr[k] + tr[:k].min())
    want = phi0.copy(order="F")
    r1 = lsf.reinit(want, None, None, *n, 30, dx, h, tol=tol, order="jacobi", arith="fast")
    for transport in ("peer", "rccl"):
        got = phi0.copy(order="F")
        r = lsf.reinit_multi(got, *n, 30, dx, h, list(range(nd)), tol=tol, arith="fast", transport=transport)
        assert r.converged and r.count == r1.count == k + 1, (transport, r.count)
        assert sha(got) == sha(want), transport


def test_multi_f32_on_two_devices_equals_single_domain_f32(lsf):
    phi0, n, dx, h = _field((90, 72, 64), np.float32)
    want = phi0.copy(order="F")
    r1 = lsf.reinit(want, None, None, *n, 7, dx, h, tol=0.0, order="jacobi", arith="fast")
    for transport in ("peer", "rccl"):
        got = phi0.copy(order="F")
        r = lsf.reinit_multi(got, *n, 7, dx, h, [0, 1], tol=0.0, arith="fast", transport=transport)
        assert r.count == r1.count == 8 and sha(got) == sha(want), transport


@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_exact_ordering_over_two_devices_equals_one_device_and_the_oracle(lsf, oracle, arith):
    """LSF_ORDER_GS over z slabs, one per DEVICE: cut planes, tile flags, hyperplane counters and verdicts are stored into the other
    device's memory and polled from it.  Field, count and trace are lsf_reinit's; in STRICT arithmetic the oracle's (= the reference's)"""
    phi0, n, dx, h = _field((72, 64, 90))
    want = phi0.copy(order="F")
    r1 = lsf.reinit(want, None, None, *n, 17, dx, h, tol=0.0, order="gs", arith=arith)
    got = phi0.copy(order="F")
    r = lsf.reinit_multi(got, *n, 17, dx, h, [0, 1], tol=0.0, arith=arith, order="gs")
    assert r.count == r1.count == 18 and r.rms == r1.rms
    assert sha(got) == sha(want), float(np.abs(got - want).max())
    if arith == "strict":
        ref = phi0.copy(order="F")
        _, cnt, _ = oracle.reinit(ref, *n, 17, dx, h, tol=0.0)
        assert cnt == 18 and sha(got) == sha(ref)


def test_exact_ordering_over_every_device_of_the_box_at_256(lsf):
    """256^3 (BASELINE configuration 2's size) over all the GPUs of the box: the reference's own field (tests/golden/synth_big.npz:
    SHA-256 after 8 sweeps of its reinit) from slabs that live on different devices"""
    import hashlib

    from conftest import GOLDEN
    from levelsetfortran_amd import fields

    path = os.path.join(GOLDEN, "synth_big.npz")
    if not os.path.exists(path):
        pytest.skip("synth_big.npz not generated")
    g = np.load(path)
    sweeps = int(g["n256_sweeps"])
    phi, dx = fields.two_sphere_phi0((256, 256, 256))
    rep = lsf.reinit_multi(phi, 255, 255, 255, sweeps - 1, dx, fields.reinit_step(dx), list(range(min(NDEV, 8))), tol=0.0, arith="strict",
                           order="gs")
    assert rep.count == sweeps
    assert hashlib.sha256(phi.reshape(-1, order="F").tobytes()).hexdigest() == str(g["n256_sha"])


@pytest.mark.parametrize("dims,arith,dtype", [((1, 1, 2), "strict", "f64"), ((2, 1, 1), "fast", "f64"), ((1, 2, 1), "fast", "f32")])
def test_two_nccl_ranks_equal_single_domain(lsf, tmp_path, dims, arith, dtype):
    """one process per GPU under torch.distributed.run, backend nccl (RCCL over xGMI): the path bench.py --gpus N times.  The ranks
    are fresh child processes (tests/nccl_two_rank_worker.py); the parent compares their owned points with its own single-domain
    sweep, and reads the in-run parity record the ranks computed over the same communicator."""
    npts, sweeps = (72, 60, 66), 9
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "nccl_two_rank_worker.py"), str(tmp_path), repr(dims),
           repr(npts), str(sweeps), arith, dtype]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    phi0, n, dx, h = _field(npts, np.float32 if dtype == "f32" else np.float64)
    ref = phi0.copy(order="F")
    rep = lsf.reinit(ref, None, None, *n, sweeps - 1, dx, h, tol=0.0, order="jacobi", arith=arith)
    got = np.full_like(ref, np.nan)
    seen = set()
    for rk in range(2):
        z = np.load(tmp_path / f"r{rk}.npz")
        got[tuple(slice(int(s), int(e)) for s, e in z["own"])] = z["data"]
        assert int(z["nsw"]) == sweeps and str(z["backend"]) == "nccl" and int(z["world"]) == 2
        assert np.allclose(z["rms"], rep.rms, rtol=1e-11 if dtype == "f64" else 1e-6, atol=0)
        seen.add(int(z["device"]))
    assert seen == {0, 1}  # two distinct devices
    assert sha(got) == sha(ref)
    rec = json.load(open(tmp_path / "parity.json"))
    assert rec["ok"] is True and rec["field_sha_equal"] is True and rec["rms_trace_equal"] is True and rec["ranks"] == 2


def test_bench_on_two_devices_carries_parity_records(lsf):
    """bench.py --gpus 2 as the driver launches it (RCCL, a GPU per rank), small sizes: one line, every decomposed / one-process /
    slab entry with a parity record that says yes, exit status 0"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "96", "--steps", "4",
           "--warmup", "2", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LSF_BENCH_SHARED_GPU")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stderr[-3000:]
    d = json.loads(lines[0])
    assert r.returncode == 0, json.dumps(d.get("decomposed", {}))[:3000]
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["parity"]["ok"] is True
    ent = d["decomposed"]["entries"]
    kinds = {(e.get("ordering"), e.get("transport", "slabs")) for e in ent if e.get("value")}
    assert ("jacobi", "nccl") in kinds and ("jacobi", "peer") in kinds and ("jacobi", "rccl") in kinds and ("gs", "slabs") in kinds, kinds
    for e in ent:
        assert e.get("error") is None, e
        assert e["parity"]["ok"] is True, e
