"""Decomposition / halo-plan index arithmetic of levelsetfortran_amd.distributed (pure host logic)."""
import itertools
import os

import numpy as np
import pytest

from conftest import ROOT

from levelsetfortran_amd import distributed as D


def test_split_points_covers_and_balances():
    for n, p in ((62, 2), (513, 4), (7, 3), (100, 1)):
        parts = D.split_points(n, p)
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [e - s for s, e in parts]
        assert max(sizes) - min(sizes) <= 1


def test_default_dims_follow_baseline_configs():
    # a 2x2x1 decomposition on 4 ranks, 2x2x2 on 8; the unit-stride axis x is the one cut last
    assert D.default_dims(4) == (1, 2, 2) and D.default_dims(8) == (2, 2, 2) and D.default_dims(1) == (1, 1, 1)
    assert D.default_dims(2) == (1, 1, 2) and D.default_dims(6) == (1, 3, 2) and D.default_dims(16) == (2, 2, 4)
    for w in (2, 3, 6, 12):
        assert np.prod(D.default_dims(w)) == w


@pytest.mark.parametrize("dims", [(2, 1, 1), (1, 2, 1), (1, 1, 2), (2, 2, 1), (2, 2, 2), (3, 1, 2)])
def test_blocks_tile_the_grid_and_regions_are_disjoint(dims):
    n = (40, 33, 27)
    world = int(np.prod(dims))
    owner = -np.ones(tuple(v + 1 for v in n), dtype=int)
    updated = np.zeros(tuple(v + 1 for v in n), dtype=int)
    for r in range(world):
        b = D.make_block(r, dims, n)
        sl = tuple(slice(s, e) for s, e in b.own)
        assert np.all(owner[sl] == -1)
        owner[sl] = r
        core, rims = D.sweep_regions(b)
        for reg in [core] + rims:
            g = tuple(slice(lo + g0, hi + g0) for (lo, hi), g0 in zip(reg, b.g0))
            updated[g] += 1
            # the stencil of every cell of the region stays inside the local box
            for a in range(3):
                if reg[a][1] > reg[a][0]:
                    assert reg[a][0] - 1 >= 0 and reg[a][1] <= b.ext[a] - 1
        # core never touches ghost-dependent cells
        for a in range(3):
            if b.coords[a] > 0 and core[a][1] > core[a][0]:
                assert core[a][0] - D.HALO >= b.own_local[a][0]
    assert np.all(owner >= 0)
    inner = updated[1:-1, 1:-1, 1:-1]
    assert np.all(inner == 1)  # every interior cell exactly once
    walls = updated.sum() - inner.sum()
    assert walls == 0


def test_halo_plan_is_symmetric():
    dims, n = (2, 2, 2), (40, 33, 27)
    plans = {r: D.halo_plan(D.make_block(r, dims, n)) for r in range(8)}
    blocks = {r: D.make_block(r, dims, n) for r in range(8)}
    for r, plan in plans.items():
        assert len(plan) == 3  # a corner rank of a 2x2x2 grid has 3 face neighbours
        for peer, send, recv, a, side in plan:
            back = [p for p in plans[peer] if p[0] == r and p[3] == a and p[4] == -side]
            assert len(back) == 1
            _, psend, precv, _, _ = back[0]
            to_g = lambda box, b: [(lo + g, hi + g) for (lo, hi), g in zip(box, b.g0)]
            assert to_g(send, blocks[r]) == to_g(precv, blocks[peer])  # what I send is what the peer expects
            assert to_g(recv, blocks[r]) == to_g(psend, blocks[peer])


def test_block_too_thin_is_rejected():
    with pytest.raises(ValueError):
        D.make_block(0, (8, 1, 1), (30, 30, 30))


def test_device_field_builder_matches_numpy_and_slices():
    """fields.two_sphere_phi0_device (used by bench.py to build every rank's block in HBM) on the CPU device: equal to
    the numpy formula to rounding, and a `ranges` block is exactly the slice of the whole field."""
    import torch

    from levelsetfortran_amd import fields

    npts = (21, 18, 15)
    whole, dx = fields.two_sphere_phi0_device(npts, torch.device("cpu"))
    host, dx2 = fields.two_sphere_phi0(npts)
    assert dx == dx2
    w = whole.reshape(npts[2], npts[1], npts[0])
    assert np.allclose(w.numpy().transpose(2, 1, 0), host, rtol=0, atol=1e-14)
    rng = ((3, 17), (0, 9), (4, 15))
    part, _ = fields.two_sphere_phi0_device(npts, torch.device("cpu"), ranges=rng)
    part = part.reshape(rng[2][1] - rng[2][0], rng[1][1] - rng[1][0], rng[0][1] - rng[0][0])
    assert torch.equal(part, w[rng[2][0]:rng[2][1], rng[1][0]:rng[1][1], rng[0][0]:rng[0][1]])


def test_rms_denominator_of_the_decomposed_loop():
    """fp64: the reference's INTEGER*4 product nx*ny*nz, wrapping (subs.f90:914); fp32 (no reference): the true product,
    because the wrapped one is negative on the 1536^3 grid of BASELINE configuration 5."""
    import torch

    class _Backend:
        host_staging = False

        def __init__(self, dtype):
            self.dtype = dtype

        def empty(self, n, dtype=None):
            return torch.empty(n, dtype=dtype or self.dtype)

        def zeros(self, n):
            return torch.zeros(n, dtype=torch.float64)

    n = (1535, 1535, 1535)
    b = D.make_block(0, (1, 1, 1), n)
    d64 = D.DistributedReinit(_Backend(torch.float64), b, 1e-3, 1e-5)
    d32 = D.DistributedReinit(_Backend(torch.float32), b, 1e-3, 1e-5)
    assert d64.den == float(np.int32(np.uint32((1535 ** 3) & 0xFFFFFFFF))) and d64.den < 0
    assert d32.den == 1535.0 ** 3
    small = D.make_block(0, (1, 1, 1), (61, 61, 61))
    assert D.DistributedReinit(_Backend(torch.float64), small, 1e-3, 1e-5).den == 61.0 ** 3


# ---------------------------------------------------------------------------------- the patched Fortran host (no GPU)
def _dropin_exe():
    exe = os.path.join(ROOT, "build", "dropin", "set3d_hip.exec")
    return exe if os.path.exists(exe) else None


@pytest.mark.skipif(_dropin_exe() is None, reason="drop-in executable not built")
def test_host_reads_the_namelist_and_pads_per_axis(tmp_path):
    """Host edits E4 / E4b without a GPU: the reference's main program, patched, reads &lsf_inputs,
    applies pad cells per axis and side, prints the grid it will use -- and then stops LOUDLY at the first seam, because
    the library has no CPU fallback.  (Grid: twoCube10's 12 x 1 x 1 box padded to 112 x 64 x 64 cells.)"""
    import subprocess

    import stl_io

    s = np.load(os.path.join(ROOT, "tests", "golden", "surfaces.npz"))
    stl_io.stl_write(tmp_path / "twoCube10.stl", s["twocube10_surfX"], s["twocube10_surfElem"])
    (tmp_path / "in.nml").write_text("&lsf_inputs\n  dx = 0.125\n  dd = 7\n  dd_lo = 7, 27, 27\n  dd_hi = 8, 28, 28\n"
                                     "  reinit_iter = 3\n  minmax_iter = 0\n  reinit2_iter = 0\n  arith = 'strict'\n  resident = 1\n  devices = 0, 0\n/\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("LSF_")}
    env["LSF_REINIT_ITER"] = "5"  # the environment overrides the namelist
    env["HIP_VISIBLE_DEVICES"] = "-1"
    env["ROCR_VISIBLE_DEVICES"] = "-1"
    p = subprocess.run(f"ulimit -s unlimited; cd {tmp_path}; {_dropin_exe()} twoCube10.stl in.nml", shell=True, env=env, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = p.stdout
    assert "Run parameters read from in.nml" in out
    assert "(namelist)" in out
    # x: ceil(12 / 0.125) + 1 + 7 + 8 = 112; y, z: ceil(1 / 0.125) + 1 + 27 + 28 = 64
    assert "Grid Size: nx = 112 , ny = 64 ,nz = 64" in out, out[-1500:]
    assert "no HIP device visible" in out and "has no CPU fallback" in out.replace("\n", "")
    assert p.returncode != 0

    # a namelist that is named but missing is an error, not a silent default
    p = subprocess.run(f"cd {tmp_path}; {_dropin_exe()} twoCube10.stl missing.nml", shell=True, env=env, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert "namelist file not found" in p.stdout and p.returncode != 0


# ---------------------------------------------------------------------------------- lsf_stl_read (host only)
def _stl_merge_literal(tris):
    """subs.f90:64-93 restated literally (test tooling; quadratic): tris float32 (ntri, 3, 3) -> nodes, 1-based elem."""
    ntri = tris.shape[0]
    nodes = np.full((max(ntri * 5, 3), 3), 1.0e6, dtype=np.float32)
    elem = np.zeros((ntri, 3), dtype=np.int32)
    bound, k = 3, 0
    for n in range(ntri):
        for p in range(3):
            v = tris[n, p]
            share = 0
            for kk in range(1, bound + 1):
                d = np.abs(nodes[kk - 1] - v)  # REAL*4 arithmetic
                if np.all(d.astype(np.float64) < 1.0e-13):
                    share = kk
                    break
            if share:
                elem[n, p] = share
            else:
                k += 1
                nodes[k - 1] = v
                elem[n, p] = k
        bound = k
    return nodes[:k].astype(np.float64), elem


def _read_native(path):
    import ctypes

    from levelsetfortran_amd import _lib

    lib = _lib.load()
    nt, nn = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.lsf_stl_read(str(path).encode(), ctypes.byref(nt), ctypes.byref(nn)))
    X = np.zeros((nn.value, 3), order="F")
    E = np.zeros((nt.value, 3), dtype=np.int32, order="F")
    _lib.check(lib.lsf_stl_get(X.ctypes.data, E.ctypes.data))
    return X, E


def test_native_stl_reader_reproduces_the_reference_merge(tmp_path):
    """lsf_stl_read (hash) == the literal quadratic loop of stlRead on: the two sample surfaces, near-duplicate tiny
    coordinates (the 1e-13 tolerance only bites below 2^-19), a vertex repeated INSIDE one triangle (not merged: the
    search bound moves once per triangle) except in the first triangle (bound starts at 3), signed zeros."""
    import stl_io

    s = np.load(os.path.join(ROOT, "tests", "golden", "surfaces.npz"))
    for tag in ("twocube10", "cube40"):
        X0, E0 = s[tag + "_surfX"].astype(np.float64), s[tag + "_surfElem"]
        stl_io.stl_write(tmp_path / (tag + ".stl"), X0, E0)
        X, E = _read_native(tmp_path / (tag + ".stl"))
        assert np.array_equal(X, X0) and np.array_equal(E, E0), tag
    rng = np.random.default_rng(7)
    pool = rng.integers(-3, 4, size=(40, 3)).astype(np.float32) * np.float32(0.25)
    pool[5] = (3e-14, 0.5, -0.25)      # within 1e-13 of pool[6] in x although the bits differ
    pool[6] = (-2e-14, 0.5, -0.25)
    pool[7] = (1.5e-6, 1e-7, 0.0)      # just below 2^-19: distinct floats 1e-13 apart do not exist here
    pool[8] = (np.float32(1.5e-6) + np.float32(2e-13), 1e-7, -0.0)
    pool[9] = (np.nextafter(np.float32(2.0 ** -19), np.float32(0)), 0.0, 0.0)
    pool[10] = (np.float32(2.0 ** -19), 0.0, 0.0)
    idx = rng.integers(0, 40, size=(300, 3))
    idx[0] = (3, 3, 4)       # first triangle repeats a vertex: merged (bound = 3 covers the slot just filled)
    idx[1] = (11, 11, 12)    # a later triangle repeats a NEW vertex: two nodes
    idx[2] = (11, 5, 6)
    tris = pool[idx]
    rec = np.zeros(300, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("pad", "<i2")]))
    rec["v"] = tris
    with open(tmp_path / "syn.stl", "wb") as f:
        f.write(b"x".ljust(80, b" "))
        f.write(np.int32(300).tobytes())
        f.write(rec.tobytes())
    Xl, El = _stl_merge_literal(tris)
    X, E = _read_native(tmp_path / "syn.stl")
    assert np.array_equal(E, El) and np.array_equal(X, Xl)
    assert E[0, 0] == E[0, 1] and E[1, 0] != E[1, 1]  # the two quirks are in the data


def test_wide_path_guard_follows_the_buffer_descriptor():
    """the 16-byte path of the exact-ordering tiles addresses a tile image of NZT + 6 planes through one raw
    buffer descriptor of 2^31 - 1 bytes.  The guard must refuse every plane size whose image does not fit -- the old test
    (4.0e9 bytes) let planes of about 3 500^2 to 4 650^2 points through, whose upper rows would have read zeros and dropped
    their stores.  No GPU needed: the library evaluates the kernels' expression on the host."""
    from levelsetfortran_amd import _lib

    lib = _lib.load()
    nzt = 16  # the one-lane-per-cell tile
    for n, want in ((511, 1), (1023, 1), (3000, 1), (3399, 1), (3500, 0), (4095, 0), (4650, 0), (8000, 0)):
        image_bytes = (nzt + 7) * (n + 1) ** 2 * 8
        assert (image_bytes <= 0x7FFFFFFF - 16) == bool(want), n
        assert lib.lsf_skew_wide_fits(n, n, nzt) == want, n
    # the boundary itself
    n = int(((0x7FFFFFFF - 16) / (8 * (nzt + 7))) ** 0.5) - 1
    assert lib.lsf_skew_wide_fits(n, n, nzt) == 1 and lib.lsf_skew_wide_fits(n + 2, n + 2, nzt) == 0
