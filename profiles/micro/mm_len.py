"""min/max flow on the band at 512^3 (bench.py's field): wall time of calls of K iterations, exact and Jacobi ordering.
python3 profiles/micro/mm_len.py [N=512] [K ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import levelsetfortran_amd as lsf
from levelsetfortran_amd import fields
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Ks = [int(v) for v in sys.argv[2:]] or [10, 50, 100, 200, 400]
dev = torch.device("cuda", 0)
x, y, z, dx = fields.grid_axes((N, N, N))
h = fields.reinit_step(dx)
d = None
for c in ((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)):
    r = np.sqrt((x[:, None, None] - c[0]) ** 2 + (y[None, :, None] - c[1]) ** 2 + (z[None, None, :] - c[2]) ** 2) - 0.5
    d = r if d is None else np.minimum(d, r)
sdf = torch.from_numpy(np.asfortranarray(d).reshape(-1, order="F")).to(dev)
for order in ("gs", "jacobi"):
    for K in Ks:
        f = sdf.clone(); nb = torch.zeros(f.numel(), dtype=torch.int32, device=dev); sb = torch.zeros_like(nb)
        lsf.narrowBand(N - 1, N - 1, N - 1, dx, f, nb, sb)
        lsf.minmaxFlow(f, nb, sb, N - 1, N - 1, N - 1, 2, dx, 0.1 * h, tol=0.0, order=order)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rep = lsf.minmaxFlow(f, nb, sb, N - 1, N - 1, N - 1, K, dx, 0.1 * h, tol=0.0, order=order)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{order} N={N} K={K}: {dt * 1e3:.3f} ms per call, {dt / K * 1e3:.4f} ms per iteration, last rms {rep.rms[-1]:.3e}", flush=True)
