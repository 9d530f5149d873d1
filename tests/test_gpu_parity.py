"""Parity of the HIP path (through the C ABI) with the reference -- via the committed golden outputs
of the reference itself and via the oracle on the same seeded inputs.  Integer/sign results and the
STRICT arithmetic are compared with `==`; the FAST arithmetic within 1e-12 RMS (north_star: 1e-10)."""
import os

import numpy as np
import pytest

from conftest import F, sha

pytestmark = pytest.mark.gpu

FAST_RMS_TOL = 1.0e-12  # north_star asks for 1e-10 RMS against the reference; measured ~1e-16


def _n(g):
    return int(g["nx"]), int(g["ny"]), int(g["nz"])


def _rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


@pytest.fixture(scope="module")
def lsf():
    import torch

    assert torch.cuda.is_available()
    import levelsetfortran_amd

    return levelsetfortran_amd


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a.ravel(order="F"))).cuda()


def _host(t, shape):
    return t.cpu().numpy().reshape(shape, order="F")


# ---------------------------------------------------------------------------------- reinit, exact order
def test_reinit_strict_matches_reference_synthetic(lsf, synth):
    nx, ny, nz = _n(synth)
    phi = F(synth["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, 15, float(synth["dx"]), float(synth["h"]), arith="strict")
    assert rep.count == 16 and not rep.converged
    assert np.array_equal(phi, synth["phi_16"])  # all 8 raster directions twice, WENO + first-order cells, BC
    assert np.allclose(rep.rms, synth["rms"], rtol=1e-11, atol=0)  # parallel sum vs sequential sum
    phi = F(synth["phi0"])
    lsf.reinit(phi, None, None, nx, ny, nz, 0, float(synth["dx"]), float(synth["h"]), arith="strict")
    assert np.array_equal(phi, synth["phi_1"])


@pytest.mark.parametrize("scale", [1e-140, 1e-60, 1e40, 1e55, 1e70, 1e140])
@pytest.mark.parametrize("order", ["gs", "jacobi"])
def test_reinit_strict_at_the_ends_of_the_exponent_range(lsf, oracle, scale, order):
    """STRICT divides by the hardware's reciprocal / Newton / correction sequence without its scale / fixup frame
    (lsf_cell.hpp: recip_refined, div_by) wherever eps + IS < 1e120, and by the plain division elsewhere.  Fields scaled
    so that the WENO divisors sit in the middle of that range, at its edge (1e55: IS ~ 1e114), beyond it (1e70), at the
    underflow end, and where the reference itself overflows to NaN (1e140): every bit as the oracle's IEEE arithmetic.
    (Below ~1e-150 the reference's own phiSign divides by an underflowed zero and fills the field with infinities; the
    sweep after that ends in its NaN STOP here as there, but the unframed sequences turn inf / dx into NaN, so a few
    wall cells of that last, never-written field differ: not part of this test.)"""
    from levelsetfortran_amd import fields

    npts = (26, 23, 21)
    phi0, dx = fields.two_sphere_phi0(npts)
    phi0 = np.asfortranarray(phi0 * scale)
    nx, ny, nz = (n - 1 for n in npts)
    h = fields.reinit_step(dx)
    want = phi0.copy(order="F")
    oracle.reinit(want, nx, ny, nz, 2, dx, h, tol=0.0, order=oracle.GS_LEX if order == "gs" else oracle.JACOBI)
    got = phi0.copy(order="F")
    try:
        lsf.reinit(got, None, None, nx, ny, nz, 2, dx, h, tol=0.0, order=order, arith="strict")
    except lsf.LsfNaNError:
        assert np.isnan(want).any()  # the reference stops on the same NaN (subs.f90:926); the field is written back
    assert np.array_equal(got, want, equal_nan=True), (scale, order, float(np.nanmax(np.abs(got - want))))


def test_reinit_fast_within_tolerance_and_sign_exact(lsf, synth):
    nx, ny, nz = _n(synth)
    phi = F(synth["phi0"])
    lsf.reinit(phi, None, None, nx, ny, nz, 15, float(synth["dx"]), float(synth["h"]), arith="fast")
    assert _rms(phi, synth["phi_16"]) < FAST_RMS_TOL
    assert np.array_equal(np.signbit(phi), np.signbit(synth["phi_16"]))


@pytest.mark.parametrize("npts", [(5, 5, 5), (9, 12, 10), (11, 11, 11), (18, 9, 10), (21, 27, 13), (70, 21, 45), (35, 35, 35)])
def test_reinit_strict_vs_oracle_ragged_sizes(lsf, oracle, npts):
    """tiny grids (no WENO cell at all), partial tiles in every axis, one-tile grids"""
    from levelsetfortran_amd import fields

    phi0, dx = fields.sphere_phi0(npts, radius=0.7, centers=((0.1, -0.2, 0.05),))
    nx, ny, nz = (v - 1 for v in npts)
    h = fields.reinit_step(dx)
    ref = phi0.copy(order="F")
    rc, n, tr = oracle.reinit(ref, nx, ny, nz, 9, dx, h, tol=0.0)
    for seam in ("host", "device"):
        if seam == "host":
            got = phi0.copy(order="F")
            rep = lsf.reinit(got, None, None, nx, ny, nz, 9, dx, h, tol=0.0, arith="strict")
        else:
            t = _dev(phi0)
            rep = lsf.reinit(t, None, None, nx, ny, nz, 9, dx, h, tol=0.0, arith="strict")
            got = _host(t, phi0.shape)
        assert rep.count == n == 10
        assert np.array_equal(got, ref), (seam, np.abs(got - ref).max())


@pytest.mark.parametrize("sweeps", [1, 8, 64])
def test_reinit_cube40_intermediate(lsf, cube40, sweeps):
    nx, ny, nz = _n(cube40)
    phi = F(cube40["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, sweeps - 1, float(cube40["dx"]), float(cube40["h"]), arith="strict")
    assert rep.count == sweeps
    assert sha(phi) == str(cube40[f"re{sweeps}_sha"])


def test_reinit_cube40_to_convergence(lsf, cube40):
    """BASELINE config 1 (as shipped, 62^3): same sweep count, bit-identical field (STRICT); FAST within tol."""
    nx, ny, nz = _n(cube40)
    dx, h, it = float(cube40["dx"]), float(cube40["h"]), int(cube40["iter_reinit"])
    phi = F(cube40["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, it, dx, h, arith="strict")
    assert rep.count == int(cube40["sweeps_reinit"]) == 2155 and rep.converged
    assert np.array_equal(phi, cube40["phi_reinit"])
    assert np.allclose(rep.rms[:-1], cube40["rms_reinit"], rtol=1e-10, atol=0)
    lines = rep.lines(0, "steady")
    assert len(lines) == 2155 and lines[-1] == "steady"  # 2154 "Iteration" lines + the steady-state line
    phi = F(cube40["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, it, dx, h, arith="fast")
    assert rep.count == 2155
    assert _rms(phi, cube40["phi_reinit"]) < FAST_RMS_TOL
    assert np.array_equal(phi < 0, cube40["phi_reinit"] < 0)  # inside/outside bit-exact


def test_reinit_twocube10(lsf, twocube):
    """BASELINE config 3's surface: pre-divergence state and the NaN stop of the reference."""
    nx, ny, nz = _n(twocube)
    dx, h = float(twocube["dx"]), float(twocube["h"])
    phi = F(twocube["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, 63, dx, h, arith="strict")
    assert rep.count == 64 and sha(phi) == str(twocube["re64_sha"])
    phi = F(twocube["phi0"])
    with pytest.raises(lsf.LsfNaNError):
        lsf.reinit(phi, None, None, nx, ny, nz, 10000, dx, h, arith="strict")
    # ... at the reference's own sweep: its NaN is born in phiSign (0 / 0, subs.f90:169) at sweep index 272, and every
    # RMS value it printed before that is reproduced (STRICT divides by dx through a reciprocal sequence that must round
    # like the IEEE division -- lsf_cell.hpp div_dx -- an error there would move this trace)
    import ctypes

    from levelsetfortran_amd import _lib

    k = int(twocube["nan_sweep_index"])
    phi = F(twocube["phi0"])
    trace = np.zeros(k + 8)
    done = ctypes.c_int(0)
    rc = _lib.load().lsf_reinit(phi.ctypes.data, nx, ny, nz, k + 5, dx, h, 1.0e-5, _lib.LSF_ARITH_STRICT, ctypes.byref(done),
                                trace.ctypes.data, k + 8)
    assert rc == _lib.LSF_ERR_NAN and done.value == k + 1
    assert np.isnan(trace[k]) and np.allclose(trace[:k], twocube["rms"][:k], rtol=1e-10, atol=0)


def test_reinit_device_seam_resume_with_phiS(lsf, oracle, synth):
    """16 sweeps == 5 sweeps + 11 more resumed at raster 5 with the original sign field."""
    nx, ny, nz = _n(synth)
    dx, h = float(synth["dx"]), float(synth["h"])
    t, s = _dev(synth["phi0"]), _dev(synth["phi0"])
    lsf.reinit(t, None, None, nx, ny, nz, 4, dx, h, tol=0.0, arith="strict", phiS=s)
    lsf.reinit(t, None, None, nx, ny, nz, 10, dx, h, tol=0.0, arith="strict", phiS=s, first_raster=5)
    assert np.array_equal(_host(t, synth["phi0"].shape), synth["phi_16"])


# ---------------------------------------------------------------------------------- reinit, Jacobi order
def test_reinit_jacobi_matches_oracle_jacobi(lsf, oracle, synth):
    nx, ny, nz = _n(synth)
    dx, h = float(synth["dx"]), float(synth["h"])
    ref = F(synth["phi0"])
    oracle.reinit(ref, nx, ny, nz, 15, dx, h, tol=0.0, order=oracle.JACOBI)
    phi = F(synth["phi0"])
    lsf.reinit(phi, None, None, nx, ny, nz, 15, dx, h, tol=0.0, order="jacobi", arith="strict")
    assert np.array_equal(phi, ref)
    phi = F(synth["phi0"])
    lsf.reinit(phi, None, None, nx, ny, nz, 15, dx, h, tol=0.0, order="jacobi", arith="fast")
    assert _rms(phi, ref) < FAST_RMS_TOL


@pytest.mark.parametrize("shape", ["1x4", "2x2", "4x1", "4x2", "8x1", None])
def test_jacobi_shared_interface_kernel_equals_per_cell_kernel(lsf, oracle, monkeypatch, shape):
    """FAST Jacobi sweeps run k_reinit_jacobi_sh (WENO interfaces shared along x and z, lsf_cell.hpp) in one of five
    block shapes (auto-selected from the row length unless LSF_JAC_SH says otherwise).  Every shape must return the bits
    of the per-cell kernel (LSF_JAC_SH=0, which evaluates the same interfaces twice per cell) -- on grids whose rows
    need helper lanes in the middle (130 and 300 cells per row), that end exactly at a block boundary (64, 128), that
    are shorter than one block, with partial rows of blocks in y and partial chunks in z -- and stay within the FAST
    tolerance of the oracle's Jacobi sweep."""
    from levelsetfortran_amd import fields

    for npts in ((24, 24, 24), (66, 21, 37), (130, 11, 9), (132, 18, 35), (302, 9, 12), (35, 70, 40)):
        phi0, dx = fields.two_sphere_phi0(npts)
        nx, ny, nz = (v - 1 for v in npts)
        h = fields.reinit_step(dx)
        monkeypatch.setenv("LSF_JAC_SH", "0")
        base = _dev(phi0)
        r0 = lsf.reinit(base, None, None, nx, ny, nz, 4, dx, h, tol=0.0, order="jacobi", arith="fast")
        if shape:
            monkeypatch.setenv("LSF_JAC_SH", shape)
        else:
            monkeypatch.delenv("LSF_JAC_SH")
        t = _dev(phi0)
        r1 = lsf.reinit(t, None, None, nx, ny, nz, 4, dx, h, tol=0.0, order="jacobi", arith="fast")
        assert r0.count == r1.count == 5
        assert bool((t == base).all()), (shape, npts, float((t - base).abs().max()))
        assert np.allclose(r0.rms, r1.rms, rtol=1e-12, atol=0)
        ref = phi0.copy(order="F")
        oracle.reinit(ref, nx, ny, nz, 4, dx, h, tol=0.0, order=oracle.JACOBI)
        assert _rms(_host(t, phi0.shape), ref) < FAST_RMS_TOL, (shape, npts)


def test_jacobi_strict_shared_difference_kernel_equals_per_cell_kernel_and_oracle(lsf, oracle, monkeypatch):
    """STRICT Jacobi sweeps run k_reinit_jacobi_strict_sh: the first and second differences of subs.f90:509-513 / :525-530
    evaluated once per point (carried along z, exchanged inside a wavefront's 16 x 4 patch of columns along x and y, the
    points beyond the patch by edge jobs).  Bits of the per-cell kernel (LSF_JAC_SH=0) and of the oracle's Jacobi sweep on
    grids with partial patches in x and y, rows shorter than a patch, fewer rows than a patch, z chunks that are not a
    multiple of the march loop's six-fold unrolling, and grids where every cell takes the first-order branch."""
    from levelsetfortran_amd import fields

    for npts in ((24, 24, 24), (66, 21, 37), (130, 11, 9), (19, 70, 40), (9, 8, 50), (8, 9, 7), (35, 37, 71), (49, 6, 13)):
        phi0, dx = fields.two_sphere_phi0(npts)
        nx, ny, nz = (v - 1 for v in npts)
        h = fields.reinit_step(dx)
        monkeypatch.setenv("LSF_JAC_SH", "0")
        base = _dev(phi0)
        r0 = lsf.reinit(base, None, None, nx, ny, nz, 5, dx, h, tol=0.0, order="jacobi", arith="strict")
        monkeypatch.delenv("LSF_JAC_SH")
        t = _dev(phi0)
        r1 = lsf.reinit(t, None, None, nx, ny, nz, 5, dx, h, tol=0.0, order="jacobi", arith="strict")
        assert r0.count == r1.count == 6
        assert bool((t == base).all()), (npts, float((t - base).abs().max()))
        assert np.allclose(r0.rms, r1.rms, rtol=1e-12, atol=0)
        ref = phi0.copy(order="F")
        oracle.reinit(ref, nx, ny, nz, 5, dx, h, tol=0.0, order=oracle.JACOBI)
        assert np.array_equal(_host(t, phi0.shape), ref), npts


def test_jacobi_fast_decomposed_equals_single_domain_bitwise(lsf, monkeypatch):
    """The block-decomposed sweep (core through the shared-interface kernel, 3-cell x rims through the per-cell THINX
    kernel, y and z rims as thin shared-interface launches) returns the bits of the single-domain FAST sweep."""
    import torch

    monkeypatch.setenv("LSF_MULTI_SMALL", "0")  # this test is about the core + rims split, which small blocks skip by default

    from levelsetfortran_amd import distributed as D
    from levelsetfortran_amd import fields

    npts = (140, 45, 38)
    phi0, dx = fields.two_sphere_phi0(npts)
    nx, ny, nz = (v - 1 for v in npts)
    h = fields.reinit_step(dx)
    whole = _dev(phi0)
    lsf.reinit(whole, None, None, nx, ny, nz, 2, dx, h, tol=0.0, order="jacobi", arith="fast")
    be = D.HipBackend(torch.device("cuda", 0), arith="fast")
    b = D.make_block(0, (1, 1, 1), (nx, ny, nz))
    dr = D.DistributedReinit(be, b, dx, h)
    fake = D.Block((3, 3, 3), (1, 1, 1), b.n, b.own, b.g0, b.ext)
    dr.core, dr.rims = D.sweep_regions(fake)
    assert len(dr.rims) == 6
    out, nsw, _ = dr.run(be.from_numpy(phi0), 2, tol=0.0)
    assert nsw == 3 and np.array_equal(be.to_numpy(out, b.ext), _host(whole, phi0.shape))


def test_box_building_blocks_single_rank(lsf, oracle, synth, monkeypatch):
    """lsf_jacobi_sweep_box + lsf_bc_box + pack/unpack (the multi-GPU pieces) on one GPU, split into
    core + rims exactly as a rank of a 2x2x2 decomposition would."""
    import torch

    monkeypatch.setenv("LSF_MULTI_SMALL", "0")  # the core + rims split of a large block, on a small fixture

    from levelsetfortran_amd import distributed as D

    nx, ny, nz = _n(synth)
    dx, h = float(synth["dx"]), float(synth["h"])
    ref = F(synth["phi0"])
    rc, n, tr = oracle.reinit(ref, nx, ny, nz, 2, dx, h, tol=0.0, order=oracle.JACOBI)
    be = D.HipBackend(torch.device("cuda", 0), arith="strict")
    b = D.make_block(0, (1, 1, 1), (nx, ny, nz))
    dr = D.DistributedReinit(be, b, dx, h)
    # force a multi-region sweep: pretend the block had neighbours on every side
    cells = D.interior_cells_local(b)
    core = [(lo + 3, hi - 3) for lo, hi in cells]
    fake = D.Block((3, 3, 3), (1, 1, 1), b.n, b.own, b.g0, b.ext)
    c2, rims = D.sweep_regions(fake)
    assert c2 == core and len(rims) == 6
    dr.core, dr.rims = c2, rims
    out, nsw, rms = dr.run(be.from_numpy(synth["phi0"]), 2, tol=0.0)
    assert nsw == 3 and np.array_equal(be.to_numpy(out, b.ext), ref)
    assert np.allclose(rms, tr, rtol=1e-11, atol=0)
    # pack / unpack round trip of a ragged sub-box
    f = be.from_numpy(synth["phi0"])
    reg = [(2, 7), (1, 9), (3, 6)]
    buf = be.empty(5 * 8 * 3)
    be.pack(f, b, reg, buf, be.compute)
    g = be.zeros(f.numel())
    be.unpack(g, b, reg, buf, be.compute)
    be.synchronize()
    a3 = synth["phi0"]
    z = np.zeros_like(a3)
    z[2:7, 1:9, 3:6] = a3[2:7, 1:9, 3:6]
    assert np.array_equal(be.to_numpy(g, b.ext), z)


# ---------------------------------------------------------------------------------- narrowBand, min/max
def test_narrowband(lsf, cube40):
    nx, ny, nz = _n(cube40)
    phi = F(cube40["phi_reinit"])
    nb = np.zeros(phi.shape, dtype=np.int32, order="F")
    sb = np.full(phi.shape, 7, dtype=np.int32, order="F")
    lsf.narrowBand(nx, ny, nz, float(cube40["dx"]), phi, nb, sb)
    assert np.array_equal(nb, cube40["NB0"]) and np.array_equal(sb, cube40["SB0"])


@pytest.fixture(params=["default", "dense"])
def mm_executor(request, monkeypatch):
    """cube40 as shipped has 32 % of its 62^3 points in the narrow band.  Since round 6 the band executor takes it by default (every
    list up to 3.5 M cells + 30 % of the grid: on a grid this small the dense executor's launches are the time); "dense" makes the dense
    executor take it (rounds 1-5's default for bands above a quarter of the grid), so that both run every cube40 case."""
    if request.param == "dense":
        monkeypatch.setenv("LSF_MINMAX_DENSE", "1")
    return request.param


@pytest.mark.parametrize("its", [1, 2, 10, 200])
def test_minmax_cube40_intermediate(lsf, cube40, its, mm_executor):
    nx, ny, nz = _n(cube40)
    phi, nb, sb = F(cube40["phi_reinit"]), F(cube40["NB0"].astype(np.int32)), F(cube40["SB0"].astype(np.int32))
    rep = lsf.minmaxFlow(phi, nb, sb, nx, ny, nz, its, float(cube40["dx"]), float(cube40["h1"]))
    assert rep.count == its and sha(phi) == str(cube40[f"mm{its}_sha"])


def test_minmax_cube40_to_convergence(lsf, cube40, mm_executor):
    nx, ny, nz = _n(cube40)
    for seam in ("host", "device"):
        phi, nb, sb = F(cube40["phi_reinit"]), F(cube40["NB0"].astype(np.int32)), F(cube40["SB0"].astype(np.int32))
        if seam == "device":
            import torch

            tp, tn, ts = _dev(phi), _dev(nb), _dev(sb)
            rep = lsf.minmaxFlow(tp, tn, ts, nx, ny, nz, 10000, float(cube40["dx"]), float(cube40["h1"]))
            phi, nb, sb = _host(tp, phi.shape), _host(tn, phi.shape), _host(ts, phi.shape)
        else:
            rep = lsf.minmaxFlow(phi, nb, sb, nx, ny, nz, 10000, float(cube40["dx"]), float(cube40["h1"]))
        assert rep.count == int(cube40["iters_minmax"]) == 406 and rep.converged
        assert np.array_equal(phi, cube40["phi_minmax"])
        # EXIT happens before narrowBand (set3d.f90:448-460): masks describe the previous iteration
        assert np.array_equal(nb, cube40["NBfinal"]) and np.array_equal(sb, cube40["SBfinal"])
        assert np.allclose(rep.rms[:-1], cube40["rms_minmax"], rtol=1e-9, atol=0)


def test_minmax_vs_oracle_other_shapes(lsf, oracle):
    from levelsetfortran_amd import fields

    for npts, order, oo in (((40, 33, 27), "gs", None), ((24, 50, 31), "gs", None), ((40, 33, 27), "jacobi", None)):
        phi0, dx = fields.sphere_phi0(npts, radius=0.45, centers=((0.0, 0.0, -0.1),), lo=-1.0, hi=1.0)
        nx, ny, nz = (v - 1 for v in npts)
        # a rough signed-distance-like field so that the band is a thin shell away from the walls
        x, y, z, _ = fields.grid_axes(npts, -1.0, 1.0)
        d = np.sqrt(x[:, None, None] ** 2 + y[None, :, None] ** 2 + (z[None, None, :] + 0.1) ** 2) - 0.45
        phi0 = np.asfortranarray(d + 0.02 * np.sin(9 * x)[:, None, None] * np.cos(7 * y)[None, :, None])
        nb, sb = oracle.narrowband(nx, ny, nz, dx, phi0)
        assert 0 < nb.sum() < nb.size // 2
        a, na, sa = phi0.copy(order="F"), nb.copy(order="F"), sb.copy(order="F")
        oracle.minmax(a, na, sa, nx, ny, nz, 12, dx, 1e-4, tol=0.0, order=oracle.GS_LEX if order == "gs" else oracle.JACOBI)
        b, nb2, sb2 = phi0.copy(order="F"), nb.copy(order="F"), sb.copy(order="F")
        rep = lsf.minmaxFlow(b, nb2, sb2, nx, ny, nz, 12, dx, 1e-4, tol=0.0, order=order)
        assert rep.count == 12
        assert np.array_equal(a, b) and np.array_equal(na, nb2) and np.array_equal(sa, sb2)


def test_zero_iterations_and_errors(lsf):
    phi = np.ones((8, 8, 8), order="F")
    nb = np.zeros((8, 8, 8), dtype=np.int32, order="F")
    rep = lsf.minmaxFlow(phi, nb, nb.copy(order="F"), 7, 7, 7, 0, 0.1, 0.01)
    assert rep.count == 0 and np.all(phi == 1.0)
    with pytest.raises(lsf.LsfError):
        lsf.reinit(np.ones((2, 2, 2), order="F"), None, None, 1, 1, 1, 0, 0.1, 0.01)  # nx must be >= 2


# ---------------------------------------------------------------------------------- phi0
def test_phi0_matches_reference(lsf, cube40, twocube):
    """set3d.f90:196-268 on the GPU: bit-identical to the phi0 the reference's main program computed."""
    import os
    import time

    import stl_io
    from conftest import GOLDEN

    s = np.load(os.path.join(GOLDEN, "surfaces.npz"))
    for tag, gold in (("cube40", cube40), ("twocube10", twocube)):
        X, E = s[tag + "_surfX"].astype(np.float64), s[tag + "_surfElem"]
        n, xLo, mn, mx = stl_io.grid_from_surface(X)
        phi = np.ones(tuple(v + 1 for v in n), order="F")
        t0 = time.perf_counter()
        lsf.phi0Init(phi, n[0], n[1], n[2], 0.05, xLo, mn, mx, X, E)
        print(tag, "phi0 on GPU", time.perf_counter() - t0, "s")
        assert np.array_equal(phi, gold["phi0"])
        t = _dev(np.zeros_like(phi))
        lsf.phi0Init(t, n[0], n[1], n[2], 0.05, xLo, mn, mx, X, E)
        assert np.array_equal(_host(t, phi.shape), gold["phi0"])


def test_phi0_first_minimum_of_rounded_distances(lsf, oracle):
    """The reference compares ROUNDED distances, `dis < minD` with dis = sqrt(d2), and keeps the first minimum (set3d.f90:231-235).
    k_phi0 takes the square root only where d2 falls below the smallest d2 seen so far -- exact, because the square root is monotone --
    and this surface is built to hit the seam of that argument: centroids on spheres of equal radius around grid points, in
    directions that are permutations and sign flips of integer triples of equal norm, so that d2 = x^2 + y^2 + z^2 differs from
    centroid to centroid in its last bits only (many different d2 with the same rounded distance, in every order)."""
    import itertools

    dirs = []
    for tri, sc in (((1, 2, 2), 0.21), ((2, 3, 6), 0.09), ((1, 4, 8), 0.07), ((4, 4, 7), 0.07)):  # all of length 0.63
        for perm in set(itertools.permutations(tri)):
            for sg in itertools.product((1, -1), repeat=3):
                dirs.append(tuple(sc * c * g for c, g in zip(perm, sg)))
    rng = np.random.default_rng(12)
    rng.shuffle(dirs)
    dx = 0.1
    centres = [(20 * dx, 20 * dx, 20 * dx), (23 * dx, 19 * dx, 21 * dx), (18 * dx, 22 * dx, 20 * dx)]  # grid points (xLo = 0)
    nodes, elems = [], []
    d = 0.015625  # a dyadic offset: the three vertices average to the centroid up to the division's rounding
    for c in centres:
        for u in dirs:
            ctr = np.array(c) + np.array(u)
            base = len(nodes)
            nodes += [ctr + (d, 0, 0), ctr + (-d, d, 0), ctr + (0, -d, 0)]
            elems.append((base + 1, base + 2, base + 3))
    X, E = np.array(nodes, dtype=np.float64), np.array(elems, dtype=np.int32)
    n = (34, 33, 35)  # the surface's bounding box and three cells around it lie inside the grid (the search box, set3d.f90:180-186)
    xLo = np.zeros(3)
    mn, mx = X.min(axis=0), X.max(axis=0)
    want = oracle.phi0(n[0], n[1], n[2], dx, xLo, mn, mx, X, E)
    got = np.ones(tuple(v + 1 for v in n), order="F")
    lsf.phi0Init(got, n[0], n[1], n[2], dx, xLo, mn, mx, X, E)
    assert np.array_equal(got, want)
    assert len(dirs) == 144 and np.unique(want).size > 1000


def test_minmax_tile_wavefront_path(lsf, oracle, cube40, monkeypatch):
    """The exact ordering has two implementations: fixed-point passes (default) and the tile-hyperplane
    wavefront (fallback when a fixed point is not certified).  Force the fallback and re-check parity."""
    monkeypatch.setenv("LSF_MINMAX_TILES", "1")
    nx, ny, nz = _n(cube40)
    phi, nb, sb = F(cube40["phi_reinit"]), F(cube40["NB0"].astype(np.int32)), F(cube40["SB0"].astype(np.int32))
    rep = lsf.minmaxFlow(phi, nb, sb, nx, ny, nz, 10000, float(cube40["dx"]), float(cube40["h1"]))
    assert rep.count == 406 and np.array_equal(phi, cube40["phi_minmax"])
    assert np.array_equal(nb, cube40["NBfinal"]) and np.array_equal(sb, cube40["SBfinal"])
    test_minmax_vs_oracle_other_shapes(lsf, oracle)


def test_minmax_uncertified_iteration_falls_back_to_the_dense_executor(lsf, oracle, cube40, monkeypatch, capfd):
    """The band executor runs every fix pass after the first inside one resident launch that loops until a pass changes nothing (up
    to 62 passes).  LSF_MINMAX_TAIL_MAX=0 leaves it none: an iteration it cannot certify ends the attempt with nothing written,
    and the call is run by the dense executor; the result must not change."""
    monkeypatch.setenv("LSF_MINMAX_TAIL_MAX", "0")
    monkeypatch.setenv("LSF_MINMAX_BAND_MAX", "100")
    monkeypatch.setenv("LSF_TRACE", "1")
    nx, ny, nz = _n(cube40)
    phi, nb, sb = F(cube40["phi_reinit"]), F(cube40["NB0"].astype(np.int32)), F(cube40["SB0"].astype(np.int32))
    rep = lsf.minmaxFlow(phi, nb, sb, nx, ny, nz, 200, float(cube40["dx"]), float(cube40["h1"]))
    assert rep.count == 200 and sha(phi) == str(cube40["mm200_sha"])
    assert "NOT certified -> dense executor" in capfd.readouterr().err
    monkeypatch.delenv("LSF_TRACE")
    monkeypatch.delenv("LSF_MINMAX_TAIL_MAX")
    test_minmax_vs_oracle_other_shapes(lsf, oracle)


def test_minmax_dense_executor(lsf, oracle, cube40, monkeypatch, capfd):
    """LSF_MINMAX_DENSE=1: the executor that streams the whole grid every iteration (the path of bands above a quarter of the grid
    and of fields beyond 32-bit point indices): cube40 to its stop, the other shapes, and its own ladder -- an uncertified
    iteration there repeats the call with the full pass count."""
    monkeypatch.setenv("LSF_MINMAX_DENSE", "1")
    nx, ny, nz = _n(cube40)
    phi, nb, sb = F(cube40["phi_reinit"]), F(cube40["NB0"].astype(np.int32)), F(cube40["SB0"].astype(np.int32))
    rep = lsf.minmaxFlow(phi, nb, sb, nx, ny, nz, 10000, float(cube40["dx"]), float(cube40["h1"]))
    assert rep.count == 406 and rep.converged and np.array_equal(phi, cube40["phi_minmax"])
    assert np.array_equal(nb, cube40["NBfinal"]) and np.array_equal(sb, cube40["SBfinal"])
    assert np.allclose(rep.rms[:-1], cube40["rms_minmax"], rtol=1e-9, atol=0)
    test_minmax_vs_oracle_other_shapes(lsf, oracle)
    monkeypatch.setenv("LSF_MINMAX_FIX_START", "1")
    monkeypatch.setenv("LSF_TRACE", "1")
    phi, nb, sb = F(cube40["phi_reinit"]), F(cube40["NB0"].astype(np.int32)), F(cube40["SB0"].astype(np.int32))
    rep = lsf.minmaxFlow(phi, nb, sb, nx, ny, nz, 10, float(cube40["dx"]), float(cube40["h1"]))
    assert rep.count == 10 and sha(phi) == str(cube40["mm10_sha"])
    assert "NOT certified -> rerun" in capfd.readouterr().err


def test_minmax_band_executor_with_a_mask_that_is_not_the_band(lsf, oracle, monkeypatch):
    """The C ABI takes ANY mask for the first iteration (set3d.f90:399 reads phiNB as the host left it).  The band executor's list is
    mask == 1 OR |phi| < 4.1 dx: a mask with cells far outside the band (they move in iteration 1 only) and WITHOUT some band cells
    (they start to move in iteration 2) -- against the oracle, both orderings, and against the dense executor."""
    from levelsetfortran_amd import fields

    npts = (36, 30, 41)
    nx, ny, nz = (v - 1 for v in npts)
    x, y, z, dx = fields.grid_axes(npts, -1.0, 1.0)
    d = np.sqrt(x[:, None, None] ** 2 + y[None, :, None] ** 2 + (z[None, None, :] + 0.1) ** 2) - 0.45
    phi0 = np.asfortranarray(d + 0.02 * np.sin(9 * x)[:, None, None] * np.cos(7 * y)[None, :, None])
    nb, sb = oracle.narrowband(nx, ny, nz, dx, phi0)
    rng = np.random.default_rng(5)
    odd = nb.copy(order="F")
    odd[rng.random(odd.shape) < 0.02] = 1          # cells anywhere in the grid, walls included (never updated there)
    odd[(rng.random(odd.shape) < 0.3) & (nb == 1)] = 0  # band cells the first iteration must skip
    odd[rng.random(odd.shape) < 0.01] = 7          # neither 0 nor 1: not in the band
    for order in ("gs", "jacobi"):
        a, na, sa = phi0.copy(order="F"), odd.copy(order="F"), sb.copy(order="F")
        oracle.minmax(a, na, sa, nx, ny, nz, 9, dx, 1e-4, tol=0.0, order=oracle.GS_LEX if order == "gs" else oracle.JACOBI)
        res = {}
        for dense in ("0", "1"):
            monkeypatch.setenv("LSF_MINMAX_DENSE", dense)
            b, nb2, sb2 = phi0.copy(order="F"), odd.copy(order="F"), sb.copy(order="F")
            rep = lsf.minmaxFlow(b, nb2, sb2, nx, ny, nz, 9, dx, 1e-4, tol=0.0, order=order)
            assert rep.count == 9
            assert np.array_equal(a, b) and np.array_equal(na, nb2) and np.array_equal(sa, sb2), (order, dense)
            res[dense] = np.array(rep.rms)
        assert np.allclose(res["0"], res["1"], rtol=1e-11, atol=0)


@pytest.mark.parametrize("geometry", [("4", "16"), ("5", "32"), ("4", "32")])
def test_reinit_alternative_tile_geometries(lsf, oracle, synth, monkeypatch, geometry):
    """Default tiles are 16 x 5 x 4 cells with three lanes per cell; 4-lane cells and 32-long tiles are kept."""
    from levelsetfortran_amd import fields

    monkeypatch.setenv("LSF_GS_NY", geometry[0])
    monkeypatch.setenv("LSF_GS_TA", geometry[1])
    nx, ny, nz = _n(synth)
    phi = F(synth["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, 15, float(synth["dx"]), float(synth["h"]), arith="strict")
    assert rep.count == 16 and np.array_equal(phi, synth["phi_16"])
    npts = (70, 21, 45)
    phi0, dx = fields.sphere_phi0(npts, radius=0.7, centers=((0.1, -0.2, 0.05),))
    h = fields.reinit_step(dx)
    ref = phi0.copy(order="F")
    oracle.reinit(ref, 69, 20, 44, 9, dx, h, tol=0.0)
    got = phi0.copy(order="F")
    lsf.reinit(got, None, None, 69, 20, 44, 9, dx, h, tol=0.0, arith="strict")
    assert np.array_equal(got, ref)


def test_dataflow_timeout_falls_back_to_slot_launches(lsf, synth, monkeypatch, capfd):
    """Every spin of the dataflow launch is bounded; if one ever runs out the library repeats the call with slot
    launches from the copy of the input it keeps as phiS.  A one-tick bound forces that path."""
    monkeypatch.setenv("LSF_GS_TIMEOUT_TICKS", "1")
    nx, ny, nz = _n(synth)
    phi = F(synth["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, 15, float(synth["dx"]), float(synth["h"]), arith="strict")
    assert rep.count == 16 and np.array_equal(phi, synth["phi_16"])
    assert "repeating the call with slot launches" in capfd.readouterr().err


@pytest.mark.parametrize("waves,schedule", [("2x2", None), ("2x2", "skew"), ("1", None), ("4x2", None), ("c1x4", None),
                                            ("c1x4", "skew"), ("c1x1", None), ("c1x2", None), ("c1x3", "skew")])
def test_reinit_odd_grid_shapes(lsf, oracle, monkeypatch, waves, schedule):
    """Grids whose extents are not multiples of the tile size (partial tiles at either end, extents below one tile,
    two interior cells per axis): every sweep direction once, bit-identical to the oracle.  2x2 tiles run the
    dataflow launch by default and slot launches with LSF_GS_SCHEDULE=skew; other shapes always use slot launches."""
    from levelsetfortran_amd import fields

    monkeypatch.setenv("LSF_GS_SKEW_W", waves)
    if schedule:
        monkeypatch.setenv("LSF_GS_SCHEDULE", schedule)
    for npts in ((4, 4, 4), (6, 10, 8), (18, 7, 6), (34, 13, 11), (10, 42, 14), (65, 8, 30), (23, 23, 5), (5, 5, 47)):
        phi0, dx = fields.two_sphere_phi0(npts)
        nx, ny, nz = (v - 1 for v in npts)
        h = fields.reinit_step(dx)
        ref = phi0.copy(order="F")
        _, n_ref, tr_ref = oracle.reinit(ref, nx, ny, nz, 8, dx, h, tol=0.0)
        got = phi0.copy(order="F")
        rep = lsf.reinit(got, None, None, nx, ny, nz, 8, dx, h, tol=0.0, arith="strict")
        assert rep.count == n_ref == 9, npts
        assert np.array_equal(got, ref), npts
        assert np.allclose(rep.rms, tr_ref[:9], rtol=1e-9, atol=0), npts


@pytest.mark.parametrize("schedule", ["planes", "slots", "skew", "dataflow", "dataflow:1", "dataflow:4x2", "skew:1", "skew:2", "skew:4", "skew:1x2", "skew:2x2", "skew:4x2", "skew:2x4", "dataflow:c1x4", "skew:c1x4",
                                      "dataflow:c1x2", "skew:c1x1", "skew:c1x3"])
def test_reinit_alternative_schedules(lsf, synth, cube40, monkeypatch, schedule):
    """The exact-GS tile graph has several executors (LSF_GS_SCHEDULE): the dataflow launch on skewed 2x2-wavefront
    tiles (one launch per batch of sweeps, dependencies resolved in the kernel; the default and what every other test
    runs), slot launches on skewed tiles of WY x WZ wavefronts (LSF_GS_SKEW_W), slot launches on box tiles, one launch
    per box-tile hyperplane, and the experimental persistent kernel on box tiles.  All must be bit-identical to the
    reference."""
    schedule, _, waves = schedule.partition(":")
    monkeypatch.setenv("LSF_GS_SCHEDULE", schedule)
    if waves:
        monkeypatch.setenv("LSF_GS_SKEW_W", waves)
    nx, ny, nz = _n(synth)
    phi = F(synth["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, 15, float(synth["dx"]), float(synth["h"]), arith="strict")
    assert rep.count == 16 and np.array_equal(phi, synth["phi_16"])
    nx, ny, nz = _n(cube40)
    phi = F(cube40["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, 63, float(cube40["dx"]), float(cube40["h"]), arith="strict")
    assert rep.count == 64 and sha(phi) == str(cube40["re64_sha"])
    # stop test: converge on a loose tolerance and compare the stop sweep and field with the default schedule
    phi = F(cube40["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, 200, float(cube40["dx"]), float(cube40["h"]), tol=3.0e-3, arith="strict")
    monkeypatch.delenv("LSF_GS_SCHEDULE")
    phi2 = F(cube40["phi0"])
    rep2 = lsf.reinit(phi2, None, None, nx, ny, nz, 200, float(cube40["dx"]), float(cube40["h"]), tol=3.0e-3, arith="strict")
    assert rep.count == rep2.count and 1 < rep.count < 200 and rep.converged and np.array_equal(phi, phi2)


def test_baseline_config1_literal_64_cubed_device_seam(lsf):
    """BASELINE config 1 at its literal size (cube40.stl, dx = 2/42: nx = ny = nz = 63), stage by stage through the device
    seam in the reference's arithmetic: phi0 (lsf_phi0) == the fixture's, reinit to the 1e-5 stop == the reference's own
    `reinit` (2 066 sweeps, field, every printed RMS), narrowBand counts, the min/max flow to its 1e-7 stop (384 iterations)
    == the pinned oracle (tests/golden/make_golden_c1.py)."""
    import hashlib
    import os

    from conftest import GOLDEN

    g = np.load(os.path.join(GOLDEN, "cube40_64.npz"))
    s = np.load(os.path.join(GOLDEN, "surfaces.npz"))
    nx = int(g["nx"])
    assert nx == 63
    dx, h, h1 = float(g["dx"]), float(g["h"]), float(g["h1"])
    phi = np.ones((nx + 1,) * 3, order="F")
    lsf.phi0Init(phi, nx, nx, nx, dx, g["xLo"], g["xMin"], g["xMax"], s["cube40_surfX"].astype(np.float64), s["cube40_surfElem"])
    assert np.array_equal(phi, g["phi0"])
    rep = lsf.reinit(phi, None, None, nx, nx, nx, 10000, dx, h, arith="strict")
    assert rep.count == int(g["sweeps"]) and rep.converged
    assert np.array_equal(phi, g["phi_re"])
    assert np.allclose(rep.rms[:-1], g["rms"], rtol=1e-9, atol=0) and np.isclose(rep.rms[-1], float(g["rms_stop"]), rtol=1e-9)
    nb = np.zeros(phi.shape, dtype=np.int32, order="F")
    sb = np.zeros(phi.shape, dtype=np.int32, order="F")
    lsf.narrowBand(nx, nx, nx, dx, phi, nb, sb)
    assert int(nb.sum()) == int(g["nb_count"]) and int(sb.sum()) == int(g["sb_count"])
    rep = lsf.minmaxFlow(phi, nb, sb, nx, nx, nx, 10000, dx, h1)
    assert rep.count == int(g["mm_iters"]) and rep.converged
    assert np.array_equal(phi[::3, ::3, ::3], g["phi_mm_sample"])
    sha_ = lambda a: hashlib.sha256(np.ascontiguousarray(a.ravel(order="F")).tobytes()).hexdigest()
    assert sha_(phi) == str(g["phi_mm_sha"]) and sha_(nb) == str(g["nb_sha"]) and sha_(sb) == str(g["sb_sha"])


def test_peer_selftest_on_one_device(lsf):
    """lsf_peer_selftest: the message-passing and atomic-max litmus of the slab launches' device-to-device hand-offs
    (DESIGN.md section 6.1), here with both kernels on device 0 -- the form a one-GPU box can run.  A pair of kernels that
    cannot see each other ends with LSF_ERR_HIP naming assumption (4), not with a hang (time-out hook)."""
    import levelsetfortran_amd as pkg

    assert pkg.peer_selftest(0, 0) == 0
    with pytest.raises(pkg.LsfError, match="out of range"):
        pkg.peer_selftest(0, 99)


def test_peer_selftest_time_out_names_assumption_4(lsf, monkeypatch):
    """its spins are bounded by a hook of their own (LSF_PEER_TIMEOUT_TICKS): with 0 ticks neither side sees the other in time and the
    call ends with LSF_ERR_HIP naming assumption (4) -- not a hang, and the next call works"""
    import levelsetfortran_amd as pkg

    monkeypatch.setenv("LSF_PEER_TIMEOUT_TICKS", "0")
    with pytest.raises(pkg.LsfError, match=r"\(4\)"):
        pkg.peer_selftest(0, 0)
    monkeypatch.delenv("LSF_PEER_TIMEOUT_TICKS")
    assert pkg.peer_selftest(0, 0) == 0


@pytest.mark.parametrize("march,nbuf", [("x", "3"), ("x", "4"), ("y", "3"), ("y", "4")])
def test_dataflow_march_axis_and_buffer_count(lsf, oracle, synth, cube40, monkeypatch, march, nbuf):
    """The dataflow launch marches its tiles along the reference's y axis by default (the kernel runs on the x <-> y
    transposed field: the p5 = 0 quirk of subs.f90:576 moves to the kernel's x lanes, raster signs and extents swap) with
    four field buffers in rotation; LSF_GS_MARCH=x / LSF_GS_NBUF=3 are the round-1 configuration.  All bit-identical to the
    reference, including non-cubic grids with partial tiles and the stop sweep."""
    from levelsetfortran_amd import fields

    monkeypatch.setenv("LSF_GS_MARCH", march)
    monkeypatch.setenv("LSF_GS_NBUF", nbuf)
    nx, ny, nz = _n(synth)
    phi = F(synth["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, 15, float(synth["dx"]), float(synth["h"]), arith="strict")
    assert rep.count == 16 and np.array_equal(phi, synth["phi_16"])
    assert np.allclose(rep.rms, synth["rms"], rtol=1e-11, atol=0)
    for npts in ((6, 10, 8), (34, 13, 11), (10, 42, 14), (65, 8, 30), (23, 23, 5), (70, 21, 45)):
        phi0, dx = fields.two_sphere_phi0(npts)
        nx, ny, nz = (v - 1 for v in npts)
        h = fields.reinit_step(dx)
        ref = phi0.copy(order="F")
        _, n_ref, tr_ref = oracle.reinit(ref, nx, ny, nz, 10, dx, h, tol=0.0)
        t, sgn = _dev(phi0), _dev(phi0)
        rep = lsf.reinit(t, None, None, nx, ny, nz, 10, dx, h, tol=0.0, arith="strict", phiS=sgn)  # caller's phiS
        assert rep.count == n_ref == 11, npts
        assert np.array_equal(_host(t, phi0.shape), ref), npts
        assert np.array_equal(_host(sgn, phi0.shape), phi0), npts  # the caller's sign field is never written
        assert np.allclose(rep.rms, tr_ref[:11], rtol=1e-9, atol=0), npts
    nx, ny, nz = _n(cube40)
    phi = F(cube40["phi0"])
    rep = lsf.reinit(phi, None, None, nx, ny, nz, 63, float(cube40["dx"]), float(cube40["h"]), arith="strict")
    assert rep.count == 64 and sha(phi) == str(cube40["re64_sha"])


def test_stop_verdict_at_many_different_sweeps(lsf):
    """A converged sweep's result must survive the sweeps still in flight behind it: the sweep that would overwrite its
    buffer learns the verdict from the very word that releases it (planes_done = np + 2), so a stop at ANY sweep index --
    every position in the buffer rotation, every raster phase -- leaves exactly the field of that sweep.  The tolerance
    is placed between consecutive running minima of the RMS trace to choose the stop sweep; the same field must come out
    of a run with that fixed sweep count (same arithmetic, deterministic), for both buffer counts."""
    import os

    from levelsetfortran_amd import fields

    npts = (88, 75, 66)
    phi0, dx = fields.two_sphere_phi0(npts)
    nx, ny, nz = (v - 1 for v in npts)
    h = fields.reinit_step(dx)
    t = _dev(phi0)
    rep = lsf.reinit(t, None, None, nx, ny, nz, 99, dx, h, tol=0.0, arith="fast")
    tr = np.asarray(rep.rms)
    assert rep.count == 100
    ks = [k for k in range(2, 100) if tr[k] < tr[:k].min()]  # sweeps that set a new minimum of the trace
    assert len(ks) >= 12, len(ks)  # sweeps 2..17 on this field: every residue mod 3, 4 and 8
    tested = 0
    for nbuf in ("4", "3"):
        os.environ["LSF_GS_NBUF"] = nbuf
        try:
            for k in ks:
                tol = 0.5 * (tr[k] + tr[:k].min())
                a = _dev(phi0)
                ra = lsf.reinit(a, None, None, nx, ny, nz, 99, dx, h, tol=tol, arith="fast")
                b = _dev(phi0)
                rb = lsf.reinit(b, None, None, nx, ny, nz, k, dx, h, tol=0.0, arith="fast")
                assert ra.converged and ra.count == k + 1 == rb.count, (nbuf, k, ra.count)
                assert bool((a == b).all()), (nbuf, k)
                tested += 1
        finally:
            del os.environ["LSF_GS_NBUF"]
    assert tested >= 24


# ---------------------------------------------------------------------------------- device-resident chain
def test_host_seams_with_device_twins(lsf, cube40, tmp_path):
    """lsf_mirror (include/lsf.h): the host-pointer seams keep phi / phiNB / phiSB on the device
    between calls.  TRUST skips the copy in of an array the previous seam left there; LAZY also leaves the results
    there until lsf_mirror_sync.  The chain reinit -> snapshot -> narrowBand -> min/max -> sum((phi - phiO)^2) -> .vti
    gives the reference's fields in every mode, and in LAZY mode the host arrays stay untouched until they are synced."""
    import ctypes

    import stl_io
    from levelsetfortran_amd import _lib

    lib = _lib.load()
    nx, ny, nz = _n(cube40)
    dx, h = float(cube40["dx"]), float(cube40["h"])
    shape = (nx + 1, ny + 1, nz + 1)
    want_re, want_mm = cube40["phi_reinit"], cube40["phi_minmax"]
    want_sum = float(np.sum((want_mm - want_re) ** 2))
    xlo = np.array([-1.5, -1.5, -1.5])
    try:
        for flags in (0, _lib.LSF_MIRROR_TRUST, _lib.LSF_MIRROR_TRUST | _lib.LSF_MIRROR_LAZY):
            _lib.check(lib.lsf_mirror(flags))
            lazy = bool(flags & _lib.LSF_MIRROR_LAZY)
            phi = F(cube40["phi0"])
            phiO = np.zeros(shape, order="F")
            nb = np.zeros(shape, dtype=np.int32, order="F")
            sb = np.zeros(shape, dtype=np.int32, order="F")
            rep = lsf.reinit(phi, None, None, nx, ny, nz, int(cube40["iter_reinit"]), dx, h, arith="strict")
            assert rep.count == 2155
            if lazy:
                assert np.array_equal(phi, cube40["phi0"])  # the result is on the device only
            else:
                assert np.array_equal(phi, want_re)
            _lib.check(lib.lsf_snapshot(phi.ctypes.data, phiO.ctypes.data, nx, ny, nz))
            f1 = str(tmp_path / f"signed_{flags}.vti").encode()
            _lib.check(lib.lsf_write_vti(f1, phi.ctypes.data, nx, ny, nz, dx, xlo.ctypes.data))
            assert np.array_equal(stl_io.vti_read_phi(f1.decode(), shape), want_re)
            assert stl_io.vti_header_count(f1.decode()) == (8 * phi.size, False)
            if flags == 0:  # the UInt64 header of fields above 4 GB, forced onto this small one: same payload, 8-byte count
                os.environ["LSF_VTI_WIDE"] = "1"
                try:
                    f1w = str(tmp_path / "signed_wide.vti").encode()
                    _lib.check(lib.lsf_write_vti(f1w, phi.ctypes.data, nx, ny, nz, dx, xlo.ctypes.data))
                finally:
                    del os.environ["LSF_VTI_WIDE"]
                assert np.array_equal(stl_io.vti_read_phi(f1w.decode(), shape), want_re)
                assert stl_io.vti_header_count(f1w.decode()) == (8 * phi.size, True)
                assert os.path.getsize(f1w.decode()) == os.path.getsize(f1.decode()) + 4 + len(' header_type="UInt64"')
            lsf.narrowBand(nx, ny, nz, dx, phi, nb, sb)
            rm = lsf.minmaxFlow(phi, nb, sb, nx, ny, nz, 10000, dx, float(cube40["h1"]))
            assert rm.count == 406
            tot = ctypes.c_double(0.0)
            _lib.check(lib.lsf_sumsq_diff(phi.ctypes.data, phiO.ctypes.data, nx, ny, nz, ctypes.byref(tot)))
            assert abs(tot.value - want_sum) <= 1e-12 * want_sum
            if lazy:
                assert np.array_equal(phi, cube40["phi0"]) and not nb.any() and not phiO.any()
                for a in (phi, nb, sb, phiO):
                    _lib.check(lib.lsf_mirror_sync(a.ctypes.data))
            assert np.array_equal(phi, want_mm) and np.array_equal(phiO, want_re)
            assert np.array_equal(nb, cube40["NBfinal"].astype(np.int32)) and np.array_equal(sb, cube40["SBfinal"].astype(np.int32))
            # a host array the caller DID change is copied in again when its twin cannot be trusted ...
            if not flags:
                phi[...] = cube40["phi0"]
                rep = lsf.reinit(phi, None, None, nx, ny, nz, 7, dx, h, arith="strict")
                assert sha(phi) == str(cube40["re8_sha"])
    finally:
        _lib.check(lib.lsf_mirror(0))
        _lib.check(lib.lsf_release_workspace())


@pytest.mark.parametrize("flags", ["trust", "lazy"])
def test_device_twins_survive_interleaved_arrays_sizes_and_precisions(lsf, flags):
    """The twin slots are shared by every host array that comes through a seam.  An un-synced (LAZY) result of array A must
    reach A when array B -- larger, smaller, the same address with another size, or the float field of the fp32 seam --
    takes the slot; a float field must never be trusted as the doubles of an earlier array; a FAILED call leaves nothing
    behind that a later sync could copy; a forgotten twin is never written through."""
    from levelsetfortran_amd import _lib, fields

    lib = _lib.load()
    fl = _lib.LSF_MIRROR_TRUST | (_lib.LSF_MIRROR_LAZY if flags == "lazy" else 0)
    lazy = flags == "lazy"

    def field(npts):
        phi0, dx = fields.two_sphere_phi0(npts)
        n = tuple(v - 1 for v in npts)
        return phi0, n, dx, fields.reinit_step(dx)

    def want(phi0, n, dx, h, sweeps, order="jacobi"):
        w = phi0.copy(order="F")
        _lib.check(lib.lsf_mirror(0))
        lsf.reinit(w, None, None, n[0], n[1], n[2], sweeps - 1, dx, h, tol=0.0, order=order, arith="strict")
        _lib.check(lib.lsf_mirror(fl))
        return w

    A0, nA, dxA, hA = field((40, 36, 30))
    B0, nB, dxB, hB = field((52, 44, 38))  # larger: the slot is reallocated
    C0, nC, dxC, hC = field((24, 20, 22))  # smaller
    try:
        wA, wB, wC = want(A0, nA, dxA, hA, 3), want(B0, nB, dxB, hB, 2), want(C0, nC, dxC, hC, 4)
        _lib.check(lib.lsf_mirror(fl))
        A, B, C = A0.copy(order="F"), B0.copy(order="F"), C0.copy(order="F")
        lsf.reinit(A, None, None, *nA, 2, dxA, hA, tol=0.0, order="jacobi", arith="strict")
        assert np.array_equal(A, A0 if lazy else wA)
        lsf.reinit(B, None, None, *nB, 1, dxB, hB, tol=0.0, order="jacobi", arith="strict")  # takes A's slot: A goes home first
        assert np.array_equal(A, wA)
        lsf.reinit(C, None, None, *nC, 3, dxC, hC, tol=0.0, order="jacobi", arith="strict")
        assert np.array_equal(B, wB)
        # the fp32 seam borrows the same slot: C goes home, and afterwards C's twin is gone (a trusted hit would read floats)
        F32 = np.asfortranarray(C0.astype(np.float32))
        lsf.reinit(F32, None, None, *nC, 1, dxC, hC, tol=0.0, order="jacobi", arith="fast")
        assert np.array_equal(C, wC)
        nb = np.zeros(C.shape, dtype=np.int32, order="F")
        sb = np.zeros(C.shape, dtype=np.int32, order="F")
        lsf.narrowBand(*nC, dxC, C, nb, sb)  # must upload C again
        if lazy:
            for a_ in (nb, sb):
                _lib.check(lib.lsf_mirror_sync(a_.ctypes.data))
        assert np.array_equal(nb != 0, np.abs(C) < 4.1 * dxC) and np.array_equal(sb != 0, np.abs(C) < 8.1 * dxC)
        # the same address with another size is another array
        big = np.zeros(60 * 50 * 44, dtype=np.float64)
        v1 = big[:A0.size].reshape(A0.shape, order="F")
        v1[...] = A0
        lsf.reinit(v1, None, None, *nA, 2, dxA, hA, tol=0.0, order="jacobi", arith="strict")
        v2 = big[:C0.size].reshape(C0.shape, order="F")  # same address, smaller
        keep = v1.copy(order="F") if not lazy else None
        if lazy:
            _lib.check(lib.lsf_mirror_sync(v1.ctypes.data))
            assert np.array_equal(v1, wA)
        v2[...] = C0
        lsf.reinit(v2, None, None, *nC, 3, dxC, hC, tol=0.0, order="jacobi", arith="strict")
        if lazy:
            _lib.check(lib.lsf_mirror_sync(v2.ctypes.data))
        assert np.array_equal(v2, wC) and (keep is None or np.array_equal(keep, wA))
        # a failed call: the twin it touched is dropped, a later sync copies nothing
        D = A0.copy(order="F")
        lsf.reinit(D, None, None, *nA, 2, dxA, hA, tol=0.0, order="jacobi", arith="strict")
        if lazy:
            assert np.array_equal(D, A0)
        with pytest.raises(lsf.LsfError):
            lsf.reinit(D, None, None, *nA, -1, dxA, hA, tol=0.0, order="jacobi", arith="strict")  # iter < 0
        before = D.copy(order="F")
        _lib.check(lib.lsf_mirror_sync(D.ctypes.data))
        assert np.array_equal(D, before)
        # forget: the twin of an array that is about to be freed is never written through
        E = A0.copy(order="F")
        lsf.reinit(E, None, None, *nA, 2, dxA, hA, tol=0.0, order="jacobi", arith="strict")
        _lib.check(lib.lsf_mirror_forget(E.ctypes.data))
        E[...] = 7.0
        _lib.check(lib.lsf_mirror_sync(E.ctypes.data))
        lsf.reinit(B, None, None, *nB, 0, dxB, hB, tol=0.0, order="jacobi", arith="strict")
        assert float(E.min()) == float(E.max()) == 7.0
    finally:
        _lib.check(lib.lsf_mirror(0))
        _lib.check(lib.lsf_release_workspace())


# ---------------------------------------------------------------------------------- advection
def test_node_advection_matches_reference(lsf, cube40):
    """set3d.f90:464-501 on the GPU: bit-identical advected nodes (host and device seam)."""
    import os

    from conftest import GOLDEN

    s = np.load(os.path.join(GOLDEN, "surfaces.npz"))
    adv = np.load(os.path.join(GOLDEN, "cube40_advect.npz"))
    nx, ny, nz = _n(cube40)
    phi, sb = F(cube40["phi_minmax"]), F(cube40["SBfinal"].astype(np.int32))
    XX = np.array(s["cube40_surfX"], dtype=np.float64, order="F")
    lsf.advectNodes(phi, sb, nx, ny, nz, float(cube40["dx"]), adv["xLo"], XX)
    assert np.array_equal(XX, adv["surfXX"])
    XX = np.array(s["cube40_surfX"], dtype=np.float64, order="F")
    lsf.advectNodes(_dev(phi), _dev(sb), nx, ny, nz, float(cube40["dx"]), adv["xLo"], XX)
    assert np.array_equal(XX, adv["surfXX"])
    with pytest.raises(lsf.LsfError):
        bad = XX.copy(order="F")
        bad[0, 0] = 99.0  # a node outside the grid would make the reference read outside phi
        lsf.advectNodes(phi, sb, nx, ny, nz, float(cube40["dx"]), adv["xLo"], bad)


def test_library_before_torch_in_a_fresh_process():
    """One HIP runtime per process whatever the import order: a process that uses the library through host arrays FIRST and touches
    torch.cuda afterwards (torch ships its own libamdhip64) once found "No HIP GPUs are available" -- _lib.load() imports torch first."""
    import subprocess
    import sys

    from conftest import ROOT

    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, levelsetfortran_amd as lsf\n"
            "from levelsetfortran_amd import fields\n"
            "phi, dx = fields.two_sphere_phi0((24, 24, 24))\n"
            "r = lsf.reinit(phi, None, None, 23, 23, 23, 3, dx, fields.reinit_step(dx), tol=0.0)\n"
            "import torch\n"
            "t = torch.ones(8, device='cuda:0', dtype=torch.float64)\n"
            "print('ok', r.count, float(t.sum().item()))\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok 4 8.0" in out.stdout, out.stderr[-1500:]
