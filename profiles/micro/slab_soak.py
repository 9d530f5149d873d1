"""randomised soak of the exact ordering: odd grids, 1-3 slabs on one device, tile shapes, both arithmetics, against lsf_reinit;
lsf_reinit (dataflow) against the slot launches (the launch with column continuation of round 4 is on the branch r04-column-continuation);
STRICT cases of up to 6e6 cell updates also against the CPU oracle (tests/oracle_lib.py: the reference's bits).
python3 profiles/micro/slab_soak.py [cases=40] [seed=1] [min points] [max points]"""
import os, sys, random
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib
import levelsetfortran_amd as L
from levelsetfortran_amd import fields
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lo_n, hi_n = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (18, 110)  # points per axis
shapes = [None, "1x1", "2x1", "4x1", "1x2", "4x2", "2x4", "c1x1", "c1x2", "c1x3", "c1x4"]
bad = 0
for case in range(n_cases):
    npts = tuple(rng.randint(lo_n, hi_n) for _ in range(3))
    arith = rng.choice(["fast", "strict"])
    shape = rng.choice(shapes)
    sweeps = rng.randint(1, 19)
    for k in ("LSF_GS_SKEW_W", "LSF_GS_SCHEDULE", "LSF_DF_BATCH"):
        os.environ.pop(k, None)
    if shape:
        os.environ["LSF_GS_SKEW_W"] = shape
    if rng.random() < 0.3:
        os.environ["LSF_DF_BATCH"] = "8"
    phi0, dx = fields.two_sphere_phi0(npts)
    n = tuple(v - 1 for v in npts); h = fields.reinit_step(dx)
    want = phi0.copy(order="F")
    r1 = L.reinit(want, None, None, *n, sweeps - 1, dx, h, tol=0.0, order="gs", arith=arith)
    os.environ["LSF_GS_SCHEDULE"] = "skew"
    slot = phi0.copy(order="F")
    r0 = L.reinit(slot, None, None, *n, sweeps - 1, dx, h, tol=0.0, order="gs", arith=arith)
    os.environ.pop("LSF_GS_SCHEDULE")
    ok = np.array_equal(slot, want) and r0.count == r1.count
    msgs = []
    if not ok:
        msgs.append("slot launches differ")
    checked_oracle = False
    if arith == "strict" and float(np.prod(n)) * sweeps <= 6e6:
        ref = phi0.copy(order="F")
        oracle_lib.reinit(ref, *n, sweeps - 1, dx, h, tol=0.0)
        checked_oracle = True
        if not np.array_equal(ref, want):
            ok = False; msgs.append("differs from the oracle")
    nzc = 16 if (shape or "").startswith("c") else 4 * int((shape or "2x2").split("x")[-1] if "x" in (shape or "2x2") else 1)
    layers = -(-(n[2] - 1) // nzc)
    for slabs in (1, 2, 3):
        if layers < slabs:
            continue
        got = phi0.copy(order="F")
        try:
            r = L.reinit_multi(got, *n, sweeps - 1, dx, h, [0] * slabs, tol=0.0, arith=arith, order="gs")
            good = np.array_equal(got, want) and r.rms == r1.rms and r.count == r1.count
        except Exception as e:  # noqa: BLE001
            good = False; msgs.append(repr(e)[:120])
        ok = ok and good
        if not good:
            msgs.append(f"slabs={slabs} differs")
    bad += not ok
    print(case, npts, arith, shape, sweeps, os.environ.get("LSF_DF_BATCH"), ("ok+oracle" if checked_oracle else "ok") if ok else ("FAIL " + "; ".join(msgs)), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
