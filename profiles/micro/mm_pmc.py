"""One warm-up iteration + K exact (fixed-point) min/max iterations on an exact two-sphere distance field at N^3, for
the kernel-trace / PMC passes of profiles/collect.sh:  python3 profiles/micro/mm_pmc.py 512 gs 16"""
import sys
import time

import torch

sys.path.insert(0, '.')
import levelsetfortran_amd as L
from levelsetfortran_amd import fields

N, order, K = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
dx = 3.0 / (N - 1)
x = -1.5 + dx * torch.arange(N, dtype=torch.float64, device='cuda')
d = None
for c in ((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)):
    r = ((x[:, None, None] - c[2]) ** 2 + (x[None, :, None] - c[1]) ** 2 + (x[None, None, :] - c[0]) ** 2).sqrt_().sub_(0.5)
    d = r if d is None else torch.minimum(d, r)
    del r
phi = d.reshape(-1)
n = N - 1
h1 = 0.1 * fields.reinit_step(dx)
nb = torch.zeros(phi.numel(), dtype=torch.int32, device='cuda')
sb = torch.zeros_like(nb)
L.narrowBand(n, n, n, dx, phi, nb, sb)
L.minmaxFlow(phi, nb, sb, n, n, n, 1, dx, h1, tol=0.0, order=order)
torch.cuda.synchronize()
t = time.perf_counter()
L.minmaxFlow(phi, nb, sb, n, n, n, K, dx, h1, tol=0.0, order=order)
torch.cuda.synchronize()
print(N, order, 'K', K, 'ms/iter', round((time.perf_counter() - t) * 1e3 / K, 4), flush=True)
