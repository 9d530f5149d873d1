"""randomised soak of the min/max flow on the narrow band (round 5, lsf_minmax_band.hpp) on one GPU: random grids, surfaces (one or
two spheres, smooth or rippled, shifted towards a wall), step sizes, iteration counts, tolerances that stop the flow early, and masks
that are NOT the band (cells anywhere, band cells missing, values other than 0 / 1) -- the band executor (any band: the 25 % limit
lifted), the dense fixed-point executor and the tile wavefront against the CPU oracle, bit for bit: field, both masks, iteration
count; the Jacobi flow against the oracle's; RMS traces to rounding.
python3 profiles/micro/soak_minmax.py [cases=80] [seed=1] [min points=8] [max points=70]"""
import os, random, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
import levelsetfortran_amd as L
from levelsetfortran_amd import fields
import oracle_lib

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo_n, hi_n = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8, 70)
rng = random.Random(seed)
bad = 0
for case in range(n_cases):
    for k in ("LSF_MINMAX_TILES", "LSF_MINMAX_DENSE", "LSF_MINMAX_BAND_MAX"):
        os.environ.pop(k, None)
    npts = tuple(rng.randint(lo_n, hi_n) for _ in range(3))
    n = tuple(v - 1 for v in npts)
    nr = np.random.default_rng(rng.randint(0, 1 << 30))
    x, y, z, dx = fields.grid_axes(npts, -1.0, 1.0)
    cs = [(rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5)) for _ in range(rng.randint(1, 2))]
    d = None
    for c in cs:
        r = np.sqrt((x[:, None, None] - c[0]) ** 2 + (y[None, :, None] - c[1]) ** 2 + (z[None, None, :] - c[2]) ** 2) - rng.uniform(0.2, 0.7)
        d = r if d is None else np.minimum(d, r)
    if rng.random() < 0.5:
        d = d + 0.05 * dx * nr.standard_normal(npts)
    phi0 = np.asfortranarray(d)
    nb, sb = oracle_lib.narrowband(*n, dx, phi0)
    mask = nb.copy(order="F")
    kind = rng.choice(["band", "band", "odd"])
    if kind == "odd":
        mask[nr.random(npts) < 0.03] = 1
        mask[(nr.random(npts) < 0.3) & (nb == 1)] = 0
        mask[nr.random(npts) < 0.01] = rng.choice([2, -1, 7])
    its = rng.randint(1, 40)
    h1 = rng.choice([1e-4, 1e-3, 0.01 * dx * dx, 0.1 * dx * dx])
    msgs, tag = [], [kind]
    for order in ("gs", "jacobi"):
        oo = oracle_lib.GS_LEX if order == "gs" else oracle_lib.JACOBI
        a, na, sa = phi0.copy(order="F"), mask.copy(order="F"), sb.copy(order="F")
        _, cnt_o, tr_o = oracle_lib.minmax(a, na, sa, *n, its, dx, h1, tol=0.0, order=oo)
        tol = 0.0
        if cnt_o >= 3 and rng.random() < 0.5:  # a tolerance between two residuals: the flow stops early (EXIT before narrowBand)
            k = rng.randint(1, cnt_o - 1)
            tol = float(np.sqrt(tr_o[k] * min(tr_o[:k]))) if tr_o[k] > 0 and tr_o[k] < min(tr_o[:k]) else 0.0
            if tol > 0.0:
                a, na, sa = phi0.copy(order="F"), mask.copy(order="F"), sb.copy(order="F")
                _, cnt_o, tr_o = oracle_lib.minmax(a, na, sa, *n, its, dx, h1, tol=tol, order=oo)
                tag.append(f"{order}:stop@{cnt_o}")
        for ex in (("band", "dense", "tiles") if order == "gs" else ("band", "dense")):
            for k in ("LSF_MINMAX_TILES", "LSF_MINMAX_DENSE", "LSF_MINMAX_BAND_MAX"):
                os.environ.pop(k, None)
            if ex == "band":
                os.environ["LSF_MINMAX_BAND_MAX"] = "100"
            elif ex == "dense":
                os.environ["LSF_MINMAX_DENSE"] = "1"
            else:
                os.environ["LSF_MINMAX_TILES"] = "1"
            b, nb2, sb2 = phi0.copy(order="F"), mask.copy(order="F"), sb.copy(order="F")
            try:
                rep = L.minmaxFlow(b, nb2, sb2, *n, its, dx, h1, tol=tol, order=order)
            except Exception as e:  # noqa: BLE001
                msgs.append(f"{order} {ex}: {e!r}"[:160])
                continue
            if not (rep.count == cnt_o and np.array_equal(a, b) and np.array_equal(na, nb2) and np.array_equal(sa, sb2)):
                msgs.append(f"{order} {ex}: count {rep.count} / {cnt_o}, field {'==' if np.array_equal(a, b) else '!='}, masks "
                            f"{'==' if np.array_equal(na, nb2) and np.array_equal(sa, sb2) else '!='}")
            elif not np.allclose(rep.rms, tr_o[:cnt_o], rtol=1e-9, atol=1e-300):
                msgs.append(f"{order} {ex}: RMS trace differs")
    bad += bool(msgs)
    print(f"case {case}: {npts} band {100.0 * float(nb.sum()) / nb.size:.0f} % its {its} h1 {h1:.1e} {' '.join(tag)}: {'OK' if not msgs else '; '.join(msgs)}", flush=True)
print(f"{n_cases} cases, {bad} failed")
sys.exit(1 if bad else 0)
