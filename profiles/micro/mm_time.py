"""Time the exact (fixed point) and Jacobi min/max iterations on an exact two-sphere distance field."""
import sys, time, torch
sys.path.insert(0, '.')
import levelsetfortran_amd as L
from levelsetfortran_amd import fields
for N in (int(a) for a in sys.argv[1:]):
    dx = 3.0 / (N - 1)
    x = -1.5 + dx * torch.arange(N, dtype=torch.float64, device='cuda')
    d = None
    for c in ((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)):
        r = ((x[:, None, None] - c[2]) ** 2 + (x[None, :, None] - c[1]) ** 2 + (x[None, None, :] - c[0]) ** 2).sqrt_().sub_(0.5)
        d = r if d is None else torch.minimum(d, r)
        del r
    phi0 = d.reshape(-1)
    n = N - 1
    h1 = 0.1 * fields.reinit_step(dx)
    for order in ('jacobi', 'gs'):
        for K in (2, 12, 24):
            phi = phi0.clone()
            nb = torch.zeros(phi.numel(), dtype=torch.int32, device='cuda'); sb = torch.zeros_like(nb)
            L.narrowBand(n, n, n, dx, phi, nb, sb)
            torch.cuda.synchronize(); t = time.perf_counter()
            L.minmaxFlow(phi, nb, sb, n, n, n, K, dx, h1, tol=0.0, order=order)
            torch.cuda.synchronize()
            print(N, order, 'K', K, 'ms/iter', round((time.perf_counter() - t) * 1e3 / K, 3), flush=True)
            del phi, nb, sb
    del phi0, d
