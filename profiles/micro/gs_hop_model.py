"""python3 profiles/micro/gs_hop_model.py 62 128 256   (host only, numpy)
Unit-hop simulation of the dataflow schedule of the exact ordering: sweeps per hop in steady state under
  (i)  the shipped condition (b): sweep q waits for the LEADING COMPLETE HYPERPLANES of sweep q-1 (planes_done >= P + H), and
  (ii) exact tile-level dependencies on sweep q-1 (the tiles that hold a cell within stencil reach of the tile).
Infinite parallelism, every tile takes one time unit after its dependencies: the length of the dependency chains only."""
import sys, numpy as np
RASTER = [(1,1,1),(1,1,-1),(1,-1,-1),(-1,-1,-1),(-1,1,-1),(-1,-1,1),(-1,1,1),(1,-1,1)]
TA = 16

def geom(n, sgn, TS):
    """per interior coordinate g = 1..n-1: (w contribution incl. TA*fT, frame tile index)"""
    g = np.arange(1, n)
    if TS == 0:
        F = g - 1 if sgn > 0 else n - 1 - g
        return F.astype(np.int64), np.zeros_like(F)
    nT = -(-(n - 1) // TS)
    t = (g - 1) // TS
    y = (g - 1) - t * TS
    cnt = np.minimum(TS, n - 1 - t * TS)
    fT = t if sgn > 0 else nT - 1 - t
    b = y if sgn > 0 else cnt - 1 - y
    return (TS * fT + b).astype(np.int64), fT.astype(np.int64)

def sweep_tiles(nx, ny, nz, signs, NYT, NZT):
    """tile id per interior cell: (m, B, C) packed; P per cell"""
    wx, _ = geom(nx, signs[0], 0)
    wy, fB = geom(ny, signs[1], NYT)
    wz, fC = geom(nz, signs[2], NZT)
    S = wx[:, None, None] + wy[None, :, None] + wz[None, None, :]
    m = S // TA
    B = np.broadcast_to(fB[None, :, None], S.shape)
    C = np.broadcast_to(fC[None, None, :], S.shape)
    tid = (m * 1024 + B) * 1024 + C
    P = m + B + C
    return tid, P

def run(N, nsweeps=17, NYT=10, NZT=8, nbuf=4, transposed=True, early=0.0, cost=0.0, early_m=0.0):
    nx = ny = nz = N - 1
    fin_prev = None
    ends = []
    hist = []  # per sweep: dict tid -> finish time, for both models
    res = {}
    for model in (("planes", "tiles") if N <= 128 and not (early or cost or early_m) else ("planes",)):
        done = []  # list of dict tid->finish
        tids_prev = P_prev = None
        for q in range(nsweeps):
            r = RASTER[q % 8]
            signs = (r[1], r[0], r[2]) if transposed else r   # kernel x = reference y
            tid, P = sweep_tiles(nx, ny, nz, signs, NYT, NZT)
            ut, inv = np.unique(tid, return_inverse=True)
            inv = inv.reshape(tid.shape)
            Pt = np.zeros(len(ut), dtype=np.int64); Pt[inv.ravel()] = P.ravel()
            np_planes = Pt.max() + 1
            # previous-sweep dependencies
            dep_prev = [set() for _ in ut]
            need_plane = np.zeros(len(ut), dtype=np.int64)
            if q > 0:
                if model == "tiles":
                    pairs = set()
                    for ax in range(3):
                        for d in range(-3, 4):
                            sl_u = [slice(None)] * 3; sl_v = [slice(None)] * 3
                            if d > 0: sl_u[ax] = slice(0, -d); sl_v[ax] = slice(d, None)
                            elif d < 0: sl_u[ax] = slice(-d, None); sl_v[ax] = slice(0, d)
                            elif ax > 0: continue
                            a = inv[tuple(sl_u)].ravel().astype(np.int64); b = inv_prev[tuple(sl_v)].ravel().astype(np.int64)
                            pr = np.unique(a * (len(ut_prev) + 1) + b)
                            for x in pr: dep_prev[x // (len(ut_prev) + 1)].add(int(x % (len(ut_prev) + 1)))
                else:
                    # H = max over (u, v in N(u)) of P_prev(v) - P_cur(u), + 1
                    Hm = -10**9
                    for ax in range(3):
                        for d in range(-3, 4):
                            sl_u = [slice(None)] * 3; sl_v = [slice(None)] * 3
                            if d > 0: sl_u[ax] = slice(0, -d); sl_v[ax] = slice(d, None)
                            elif d < 0: sl_u[ax] = slice(-d, None); sl_v[ax] = slice(0, d)
                            elif ax > 0: continue
                            Hm = max(Hm, int((P_prev[tuple(sl_v)] - P[tuple(sl_u)]).max()))
                    H = Hm + 1
                    need_plane = np.minimum(Pt + H, np_prev)  # planes_done[q-1] >= this
            # finish times
            order = np.argsort(Pt, kind="stable")
            fin = np.zeros(len(ut))
            idx = {int(t): i for i, t in enumerate(ut)}
            if q > 0:
                fin_prev_arr = np.array([done[-1][int(t)] for t in ut_prev])
                # time at which planes_done[q-1] reaches k: max finish over tiles with P < k (leading complete planes)
                plane_done_time = np.zeros(np_prev + 2)
                for k in range(1, np_prev + 1):
                    sel = Pt_prev == k - 1
                    plane_done_time[k] = max(plane_done_time[k - 1], fin_prev_arr[sel].max() if sel.any() else 0.0)
            verdict = done[q - nbuf] if q >= nbuf else None
            t_verdict = max(verdict.values()) if verdict else 0.0
            for i in order:
                t = int(ut[i]); m_, B_, C_ = t // (1024 * 1024), (t // 1024) % 1024, t % 1024
                ready = t_verdict
                for n_up, up in enumerate(((m_ - 1, B_, C_), (m_, B_ - 1, C_), (m_, B_, C_ - 1))):
                    k = (up[0] * 1024 + up[1]) * 1024 + up[2]
                    # a hand-off finer than a tile: the y / z neighbours (and, in round 5's probe only, the next tile of the row bundle)
                    # may start `early` hop units before the upstream tile is complete
                    if k in idx: ready = max(ready, fin[idx[k]] - (early_m if n_up == 0 else early))
                if q > 0:
                    if model == "tiles":
                        for j in dep_prev[i]: ready = max(ready, fin_prev_arr[j])
                    else:
                        ready = max(ready, plane_done_time[int(need_plane[i])])
                fin[i] = ready + 1.0 + cost
            done.append({int(t): float(f) for t, f in zip(ut, fin)})
            ut_prev, inv_prev, P_prev, Pt_prev, np_prev = ut, inv, P, Pt, np_planes
        ends = [max(d.values()) for d in done]
        res[model] = (ends[-1] - ends[-1 - 8]) / 8.0
    return res

for N in [int(a) for a in sys.argv[1:]] or [62, 128]:
    r = run(N)
    tl = f", tile-level dependencies {r['tiles']:.2f} ({100*(r['tiles']/r['planes']-1):+.1f} %)" if "tiles" in r else ""
    print(f"N={N}: hops per sweep in steady state: hyperplane condition {r['planes']:.2f}{tl}", flush=True)
    # the hand-off finer than a tile in the same model: half a march = 0.18 of a hop earlier for the y / z neighbours
    a = run(N, early=0.18)['planes']; b = run(N, early=0.18, cost=0.07)['planes']; c = run(N, early=0.18, early_m=0.18)['planes']
    print(f"      y / z neighbours released 0.18 hop early: {a:.2f} ({100*(a/r['planes']-1):+.1f} %); the same with every tile 0.07 hop longer: {b:.2f} "
          f"({100*(b/r['planes']-1):+.1f} %); round 5's probe (the next tile of the row bundle released early too): {c:.2f} ({100*(c/r['planes']-1):+.1f} %)", flush=True)
