"""randomised soak of the paths the exact-ordering soak (slab_soak.py) does not touch, on one GPU:
  J  Jacobi ordering: STRICT against the CPU oracle (bitwise, where small enough), FAST against STRICT (RMS), lsf_reinit_multi on a
     random block decomposition of one device (peer copies or the RCCL schedule's stand-in, small-block mode on / off, random
     check_every) against lsf_reinit bit for bit -- fp64 FAST / STRICT and fp32;
  M  min/max flow: narrowBand and the exact flow in both executors (fixed-point passes, tile wavefront) against the oracle, bit for
     bit: field, masks, iteration count; the Jacobi flow against the oracle's;
  S  the stop test of the exact ordering (grids of up to 40 000 cells): a tolerance placed between two residuals of the oracle's trace
     must stop every executor (dataflow launch with random batch length and buffer count, slot launches on skewed and on box tiles, plane launches, resident blocks, two
     slabs) at the oracle's sweep with the oracle's field, STRICT; FAST: the executors among themselves.
python3 profiles/micro/soak_other.py [cases=60] [seed=1] [min points=8] [max points=80]"""
import os, random, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
import levelsetfortran_amd as L
from levelsetfortran_amd import fields
import oracle_lib

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo_n, hi_n = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8, 80)
rng = random.Random(seed)
bad = 0


def field(npts, rough):
    phi0, dx = fields.two_sphere_phi0(npts)
    if rough:  # not only the smooth distance field: a ripple of a few per cent of a cell
        r = np.random.default_rng(rng.randint(0, 1 << 30)).standard_normal(npts)
        phi0 = np.asfortranarray(phi0 + 0.03 * dx * r)
    return phi0, dx


for case in range(n_cases):
    for k in ("LSF_MULTI_SMALL", "LSF_MINMAX_TILES"):
        os.environ.pop(k, None)
    npts = tuple(rng.randint(lo_n, hi_n) for _ in range(3))
    n = tuple(v - 1 for v in npts)
    rough = rng.random() < 0.5
    phi0, dx = field(npts, rough)
    h = fields.reinit_step(dx)
    msgs = []
    tag = []
    # ---- J: Jacobi ordering ------------------------------------------------------------------------------------------------
    sweeps = rng.randint(1, 12)
    strict = phi0.copy(order="F")
    rs = L.reinit(strict, None, None, *n, sweeps - 1, dx, h, tol=0.0, order="jacobi", arith="strict")
    fast = phi0.copy(order="F")
    rf = L.reinit(fast, None, None, *n, sweeps - 1, dx, h, tol=0.0, order="jacobi", arith="fast")
    d = fast - strict
    rms = float(np.sqrt(np.mean(d * d)))
    if not (rs.count == rf.count == sweeps and rms <= 1e-10):  # north_star's tolerance for the FAST arithmetic (largest seen in 2 000 cases: 1.4e-12)
        msgs.append(f"Jacobi FAST - STRICT rms {rms:.2e}")
    if float(np.prod(n)) * sweeps <= 4e6:
        ref = phi0.copy(order="F")
        oracle_lib.reinit(ref, *n, sweeps - 1, dx, h, tol=0.0, order=oracle_lib.JACOBI)
        tag.append("J=oracle")
        if not np.array_equal(ref, strict):
            msgs.append("Jacobi STRICT differs from the oracle")
    # block decomposition on one device: every axis needs >= 7 owned points per block (3 ghost layers each side)
    dims = [rng.randint(1, 3) for _ in range(3)]
    while dims[0] * dims[1] * dims[2] > 6:
        dims[rng.randrange(3)] = 1
    dims = [dv if n[ax] + 1 >= 8 * dv else 1 for ax, dv in enumerate(dims)]
    nb = dims[0] * dims[1] * dims[2]
    arith = rng.choice(["fast", "strict"])
    small = rng.choice(["0", "192"])
    os.environ["LSF_MULTI_SMALL"] = small
    ce = rng.choice([1, 3, 8])
    transport = rng.choice(["peer", "mock"]) if nb > 1 else "peer"
    want = fast if arith == "fast" else strict
    got = phi0.copy(order="F")
    try:
        r = L.reinit_multi(got, *n, sweeps - 1, dx, h, [0] * nb, dims=dims, tol=0.0, arith=arith, check_every=ce, transport=transport)
        if not (np.array_equal(got, want) and r.count == sweeps):
            msgs.append(f"blocks {dims} {arith} small={small} ce={ce} {transport} differ")
    except Exception as e:  # noqa: BLE001
        msgs.append(f"blocks {dims} {arith} small={small}: {e!r}"[:200])
    # fp32 (FAST only), single launch against blocks
    p32 = phi0.astype(np.float32, order="F")
    one = p32.copy(order="F")
    L.reinit(one, None, None, *n, sweeps - 1, dx, h, tol=0.0, order="jacobi", arith="fast")
    blk = p32.copy(order="F")
    try:
        L.reinit_multi(blk, *n, sweeps - 1, dx, h, [0] * nb, dims=dims, tol=0.0, arith="fast", check_every=ce, transport=transport)
        if not np.array_equal(one, blk):
            msgs.append(f"fp32 blocks {dims} small={small} differ")
    except Exception as e:  # noqa: BLE001
        msgs.append(f"fp32 blocks {dims}: {e!r}"[:200])
    os.environ.pop("LSF_MULTI_SMALL")
    # ---- M: min/max flow -----------------------------------------------------------------------------------------------------
    its = rng.randint(1, 10)
    h1 = rng.choice([1e-4, 0.01 * dx / 3.5])
    src = strict  # a field a few Jacobi sweeps old: a band around both spheres
    nb_o, sb_o = oracle_lib.narrowband(*n, dx, src)
    nb_g = np.zeros(src.shape, dtype=np.int32, order="F"); sb_g = nb_g.copy(order="F")
    L.narrowBand(*n, dx, src, nb_g, sb_g)
    if not (np.array_equal(nb_o, nb_g) and np.array_equal(sb_o, sb_g)):
        msgs.append("narrowBand differs")
    if float(np.prod(n)) * its <= 6e6:
        for order_, oo in (("gs", oracle_lib.GS_LEX), ("jacobi", oracle_lib.JACOBI)):
            a, na, sa = src.copy(order="F"), nb_o.copy(order="F"), sb_o.copy(order="F")
            _, cnt_o, tr_o = oracle_lib.minmax(a, na, sa, *n, its, dx, h1, tol=0.0, order=oo)
            for forced in (("", "1") if order_ == "gs" else ("",)):
                if forced:
                    os.environ["LSF_MINMAX_TILES"] = forced
                b, nb2, sb2 = src.copy(order="F"), nb_o.copy(order="F"), sb_o.copy(order="F")
                rep = L.minmaxFlow(b, nb2, sb2, *n, its, dx, h1, tol=0.0, order=order_)
                os.environ.pop("LSF_MINMAX_TILES", None)
                if not (rep.count == cnt_o and np.array_equal(a, b) and np.array_equal(na, nb2) and np.array_equal(sa, sb2)):
                    msgs.append(f"min/max {order_}{' tiles' if forced else ''} differs from the oracle")
        tag.append("M=oracle")
    # ---- S: stop test ------------------------------------------------------------------------------------------------------
    if float(np.prod(n)) <= 4e4 and min(n) >= 3:
        K = rng.randint(6, 40)
        ref = phi0.copy(order="F")
        _, _, tr = oracle_lib.reinit(ref, *n, K - 1, dx, h, tol=0.0)
        cand = [k for k in range(1, K) if tr[k] < min(tr[:k]) and tr[k] > 0]
        if cand:
            k = rng.choice(cand)
            tol = float(np.sqrt(tr[k] * min(tr[:k])))
            ref = phi0.copy(order="F")
            _, cnt_o, tr_o = oracle_lib.reinit(ref, *n, K - 1, dx, h, tol=tol)
            assert cnt_o == k + 1
            for arith_ in ("strict", "fast"):
                first = None
                for ex in ("dataflow", "skew", "slots", "planes", "slabs"):
                    for kk in ("LSF_GS_SCHEDULE", "LSF_DF_BATCH", "LSF_GS_NBUF"):
                        os.environ.pop(kk, None)
                    if rng.random() < 0.5:
                        os.environ["LSF_DF_BATCH"] = "8"
                    if rng.random() < 0.5:
                        os.environ["LSF_GS_NBUF"] = "3"
                    if ex in ("skew", "slots", "planes"):
                        os.environ["LSF_GS_SCHEDULE"] = ex
                    g = phi0.copy(order="F")
                    try:
                        if ex == "slabs":
                            if n[2] - 1 < 16:
                                continue
                            rr = L.reinit_multi(g, *n, K - 1, dx, h, [0, 0], tol=tol, arith=arith_, order="gs")
                        else:
                            rr = L.reinit(g, None, None, *n, K - 1, dx, h, tol=tol, order="gs", arith=arith_)
                    except Exception as e:  # noqa: BLE001
                        msgs.append(f"stop {arith_} {ex}: {e!r}"[:200])
                        continue
                    if arith_ == "strict":
                        if not (rr.count == cnt_o and np.array_equal(g, ref) and np.allclose(rr.rms, tr_o, rtol=1e-9, atol=0)):
                            msgs.append(f"stop strict {ex}: sweep {rr.count} against {cnt_o}" + ("" if np.array_equal(g, ref) else ", field differs"))
                    elif first is None:
                        first = (rr.count, g, rr.rms)
                    elif not (rr.count == first[0] and np.array_equal(g, first[1]) and
                              (np.allclose(rr.rms, first[2], rtol=1e-12, atol=0) if ex in ("skew", "slots", "planes") else rr.rms == first[2])):  # (slot launches sum the RMS in another order)
                        msgs.append(f"stop fast {ex} differs from the dataflow launch")
                for kk in ("LSF_GS_SCHEDULE", "LSF_DF_BATCH", "LSF_GS_NBUF"):
                    os.environ.pop(kk, None)
            tag.append(f"S=oracle@{cnt_o}")
    bad += bool(msgs)
    print(case, npts, "rough" if rough else "smooth", f"sweeps {sweeps} blocks {dims} {arith} small={small} ce={ce} {transport} its {its}", " ".join(tag),
          "ok" if not msgs else "FAIL " + "; ".join(msgs), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
