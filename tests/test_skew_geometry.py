"""Host-side restatement of the skewed-tile geometry of the exact Gauss-Seidel reinit executor
(levelsetfortran_amd/csrc/lsf_skew.hpp, get_skew_tiles / skew_spacing in lsf_api.hip).

The reference sweeps the grid in place in raster order (subs.f90:743-852, eight cyclic directions): a cell needs
this sweep's values of the three cells before it on every axis and last sweep's values of the three cells after it.
The GPU executor cuts every (5 WY x 4 WZ)-row bundle (WY x WZ wavefronts per tile) into tiles along the skew
coordinate s = Fx + Fy + Fz and launches them by hyperplanes m + fB + fC.  These tests check, on small grids and without a GPU, the properties the
bit-for-bit GPU parity tests rely on:
  * every interior cell belongs to exactly one tile of a sweep and every address the loader forms lies in the array,
  * a cell's upstream stencil cells run in an earlier launch or on an earlier step of the same tile,
  * the closed-form spacing of consecutive sweeps is never smaller than the exact (brute force) requirement.
The arithmetic below mirrors the kernel line by line (row table, clamps, frame coordinates).
"""
import itertools

import numpy as np
import pytest

TA = 16


def cdiv(a, b):
    return -(-a // b)


# (WY, WZ[, BY]) shapes: bundles of BY x 4 rows per wavefront, BY = 5 (three lanes per cell, the default) or 16 (one lane
# per cell: LSF_GS_SKEW_W=c1x4 ..., chosen by the library itself on grids of 900 cells and more across)
WAVES = ((1, 1), (2, 1), (4, 1), (8, 1), (2, 2), (4, 2), (4, 4), (1, 1, 16), (1, 2, 16), (1, 3, 16), (1, 4, 16))


def layout(w):
    """SkTile<16, WY, WZ, BY>: rows of a tile in y and z, first row of each LDS row group, number of rows"""
    wy, wz = w[:2]
    by = w[2] if len(w) > 2 else 5
    nyt, nzt = by * wy, 4 * wz
    ncore, yh, zp = nzt * nyt, 3 * nzt, (3 * nyt + 3) // 4 * 4
    yu0, zu0 = ncore, ncore + yh
    yd0 = zu0 + zp
    zd0 = yd0 + yh
    nr = cdiv(zd0 + zp, 4 * wy * wz) * (4 * wy * wz)
    return nyt, nzt, yu0, zu0, yd0, zd0, nr


RASTER = [(1, 1, 1), (1, 1, -1), (1, -1, -1), (-1, -1, -1), (-1, 1, -1), (-1, -1, 1), (-1, 1, 1), (1, -1, 1)]
GRIDS = [(24, 24, 24), (39, 32, 26), (7, 9, 5), (3, 3, 3), (16, 5, 4), (17, 6, 5), (33, 12, 10), (4, 20, 3)]


def frame(g, n, ts, sgn):
    """interior coordinate g (1..n-1) -> (frame index F, bundle index fT) for sweep direction sgn"""
    if ts == 0:
        return (g - 1 if sgn > 0 else n - 1 - g), 0
    n_t = cdiv(n - 1, ts)
    t = (g - 1) // ts
    y = (g - 1) - t * ts
    cnt = min(ts, n - 1 - t * ts)
    f_t = t if sgn > 0 else n_t - 1 - t
    b = y if sgn > 0 else cnt - 1 - y
    return ts * f_t + b, f_t


def row_of(r, nj, nk, w):
    """LDS row r -> frame-local (b', c') exactly as the kernel's row table decodes it"""
    NY, _, YU0, ZU0, YD0, ZD0, _ = layout(w)
    if r < YU0:
        cq, bq = divmod(r, NY)
        return min(bq, nj - 1), min(cq, nk - 1)
    if r < ZU0:
        cq, hy = divmod(r - YU0, 3)
        return hy - 3, min(cq, nk - 1)
    if r < YD0:
        hz, bq = divmod(min(r - ZU0, 3 * NY - 1), NY)
        return min(bq, nj - 1), hz - 3
    if r < ZD0:
        cq, hy = divmod(r - YD0, 3)
        return nj + hy, min(cq, nk - 1)
    hz, bq = divmod(min(r - ZD0, 3 * NY - 1), NY)
    return min(bq, nj - 1), nk + hz


@pytest.mark.parametrize("w", WAVES)
@pytest.mark.parametrize("dims", GRIDS)
def test_tiles_cover_every_cell_once_and_addresses_stay_inside(dims, w):
    nx, ny, nz = dims
    NY, NZ, _, _, _, _, NR = layout(w)
    sx, sxy = nx + 1, (nx + 1) * (ny + 1)
    n = sxy * (nz + 1)
    n_tj, n_tk, nxi = cdiv(ny - 1, NY), cdiv(nz - 1, NZ), nx - 1
    for si, sj, sk in itertools.product((1, -1), repeat=3):
        cover = np.zeros(n, dtype=np.int32)
        for f_c, f_b in itertools.product(range(n_tk), range(n_tj)):
            m_lo = (NY * f_b + NZ * f_c) // TA
            m_hi = (NY * f_b + NY - 1 + NZ * f_c + NZ - 1 + nxi - 1) // TA
            tj = f_b if sj > 0 else n_tj - 1 - f_b
            tk = f_c if sk > 0 else n_tk - 1 - f_c
            j_lo, k_lo = 1 + tj * NY, 1 + tk * NZ
            nj, nk = min(NY, ny - j_lo), min(NZ, nz - k_lo)
            org_j, org_k = max(j_lo - 3, 0), max(k_lo - 3, 0)
            org = sx * org_j + sxy * org_k
            for m in range(m_lo, m_hi + 1):
                x0 = TA * m - NY * f_b - NZ * f_c
                gi0 = 1 + x0 if si > 0 else nx - 1 - x0
                table = []
                for r in range(NR):
                    bq, cq = row_of(r, nj, nk, w)
                    gj = min(max(j_lo + (bq if sj > 0 else nj - 1 - bq), 0), ny)
                    gk = min(max(k_lo + (cq if sk > 0 else nk - 1 - cq), 0), nz)
                    o = (gj - org_j) * sx + (gk - org_k) * sxy
                    assert o >= 0
                    table.append((o, gi0 - (bq + cq) if si > 0 else gi0 + (bq + cq)))
                for o, y in table:  # every entry the loader touches
                    for k in range(TA + 6):
                        gi = min(max(y + (k - 3 if si > 0 else 3 - k), 0), nx)
                        assert 0 <= org + o + gi < n
                for r in range(NY * NZ):  # write-back of the bundle rows
                    cq, bq = divmod(r, NY)
                    o, y = table[r]
                    for t in range(TA):
                        gi = y + (t if si > 0 else -t)
                        if bq < nj and cq < nk and 1 <= gi <= nx - 1:
                            cover[org + o + gi] += 1
        c3 = cover.reshape(nz + 1, ny + 1, nx + 1)
        assert (c3[1:nz, 1:ny, 1:nx] == 1).all()
        c3[1:nz, 1:ny, 1:nx] = 0
        assert not c3.any()


def plane_maps(nx, ny, nz, direction, NY, NZ):
    si, sj, sk = direction
    fx = np.array([frame(g, nx, 0, si)[0] for g in range(1, nx)])
    fy = np.array([frame(g, ny, NY, sj) for g in range(1, ny)])
    fz = np.array([frame(g, nz, NZ, sk) for g in range(1, nz)])
    s = fx[None, None, :] + fy[None, :, 0, None] + fz[:, 0, None, None]
    m = s // TA
    plane = m + fy[None, :, 1, None] + fz[:, 1, None, None]
    tile = (m * 4096 + fy[None, :, 1, None]) * 4096 + fz[:, 1, None, None]
    return plane, s - TA * m, tile


def shifted(a, axis, d, fill):
    """b[u] = a[u + d] along axis, `fill` outside"""
    b = np.full_like(a, fill)
    dst, src = [slice(None)] * 3, [slice(None)] * 3
    if d > 0:
        dst[axis], src[axis] = slice(0, -d), slice(d, None)
    else:
        dst[axis], src[axis] = slice(-d, None), slice(0, d)
    b[tuple(dst)] = a[tuple(src)]
    return b


def spacing_closed_form(da, db, nx, ny, nz, NY, NZ):
    """skew_spacing() of lsf_api.hip"""
    m0, md = [0] * 3, [0] * 3
    for ax, (n, ts) in enumerate(((nx, 0), (ny, NY), (nz, NZ))):
        def w(g, sgn):
            f, f_t = frame(g, n, ts, sgn)
            return f + TA * f_t

        a0 = ad = -(10 ** 12)
        for g in range(1, n):
            wu = w(g, db[ax])
            for d in range(-3, 4):
                if 1 <= g + d <= n - 1:
                    v = w(g + d, da[ax]) - wu
                    if d == 0:
                        a0 = max(a0, v)
                    else:
                        ad = max(ad, v)
        m0[ax], md[ax] = a0, max(ad, a0)
    tot = sum(m0)
    for ax in range(3):
        tot = max(tot, sum(m0) - m0[ax] + md[ax])
    return tot // TA + 2


@pytest.mark.parametrize("w", WAVES)
@pytest.mark.parametrize("dims", GRIDS + [(64, 64, 64), (100, 37, 51)])
def test_launch_order_respects_the_in_place_sweep(dims, w):
    nx, ny, nz = dims
    NY, NZ = layout(w)[:2]
    maps = [plane_maps(nx, ny, nz, d, NY, NZ) for d in RASTER]
    for direction, (plane, step, tile) in zip(RASTER, maps):
        for axis, sgn in zip((2, 1, 0), direction):  # arrays are [k, j, i]
            for d in (1, 2, 3):
                pv, tv, sv = (shifted(a, axis, -sgn * d, f) for a, f in ((plane, -10 ** 6), (tile, -1), (step, -1)))
                # the cell d places before this one: an earlier launch, or an earlier step of the same tile
                assert ((pv < plane) | ((tv == tile) & (sv < step)) | (pv == -10 ** 6)).all()
                tw, sw = shifted(tile, axis, sgn * d, -1), shifted(step, axis, sgn * d, -1)
                # the cell d places after it still holds last sweep's value when this one is updated
                assert ((tw != tile) | (sw > step)).all()
    for q in range(8):
        pa, pb = maps[q][0], maps[(q + 1) & 7][0]
        need = (pa - pb).max()
        for axis in range(3):
            for d in (-3, -2, -1, 1, 2, 3):
                need = max(need, (shifted(pa, axis, d, -10 ** 6) - pb).max())
        got = spacing_closed_form(RASTER[q], RASTER[(q + 1) & 7], nx, ny, nz, NY, NZ)
        assert need + 1 <= got <= need + 2
