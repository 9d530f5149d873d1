"""Run ON THE GPU BOX: band executor against dense executor of the exact min/max flow on SMALL grids with a WIDE band (the reference's
default run: 62^3, band 35 % -- above the 25 % up to which the band executor is selected on bandwidth grounds).  The band is widened by
handing the flow a larger dx (timing only).  python3 profiles/micro/mm_band_small.py"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import levelsetfortran_amd as lsf
from levelsetfortran_amd import fields

dev = torch.device("cuda", 0)
for N in (48, 62, 96, 128, 160, 200, 256, 384):
    x, y, z, dx = fields.grid_axes((N, N, N))
    d = None
    for c in ((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)):
        r = np.sqrt((x[:, None, None] - c[0]) ** 2 + (y[None, :, None] - c[1]) ** 2 + (z[None, None, :] - c[2]) ** 2) - 0.5
        d = r if d is None else np.minimum(d, r)
    sdf = torch.from_numpy(np.asfortranarray(d).reshape(-1, order="F")).to(dev)
    n = N - 1
    for dxe in (dx, 0.10, 0.21):  # the true spacing; ~20 % and ~45 % of the grid in the band
        h1 = 0.01 * dxe / np.sqrt(27.0)
        res = {}
        for bm in ("25", "100", "rule"):
            if bm == "rule":
                os.environ.pop("LSF_MINMAX_BAND_MAX", None)  # the library's own rule (round 6: list <= 3.5 M cells + 30 % of the grid)
            else:
                os.environ["LSF_MINMAX_BAND_MAX"] = bm
            out = None
            for rep in range(2):
                f = sdf.clone()
                nb = torch.zeros(f.numel(), dtype=torch.int32, device=dev); sb = torch.zeros_like(nb)
                lsf.narrowBand(n, n, n, dxe, f, nb, sb)
                band = float((nb == 1).sum().item()) / nb.numel()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rep_ = lsf.minmaxFlow(f, nb, sb, n, n, n, 100, dxe, h1, tol=0.0, order="gs")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            res[bm] = (dt, f.clone(), rep_.count)
        same = torch.equal(res["25"][1], res["100"][1]) and torch.equal(res["25"][1], res["rule"][1])
        print(f"N={N} band {100 * band:.1f} %: dense above 25 % (rounds 1-5) {res['25'][0] * 10:.3f} ms per iteration, band executor forced {res['100'][0] * 10:.3f}, "
              f"the library's rule {res['rule'][0] * 10:.3f}; fields equal: {same}", flush=True)
