"""Cost of the core and rim regions of one rank of a 2x2x2 block decomposition (local 512^3 + ghosts) on one GPU."""
import sys, time, torch
sys.path.insert(0, '.')
from levelsetfortran_amd import distributed as lsd, fields
dev = torch.device('cuda', 0)
N = 1024  # global points per axis; rank (0,0,0) of 2x2x2 owns 512^3
b = lsd.make_block(0, (2, 2, 2), (N - 1, N - 1, N - 1))
be = lsd.HipBackend(dev)
n = b.npoints_local()
a = torch.rand(n, dtype=torch.float64, device=dev) * 0.1
out = torch.empty_like(a); ps = a.clone(); ss = torch.zeros(8, dtype=torch.float64, device=dev)
core, rims = lsd.sweep_regions(b)
dx = 3.0 / (N - 1); h = fields.reinit_step(dx)
def t(region, reps=5):
    be.sweep(a, out, ps, b, region, dx, h, ss, be.compute); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): be.sweep(a, out, ps, b, region, dx, h, ss, be.compute)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print('ext', b.ext, 'core', core, 'ms', round(t(core), 3))
for r in rims: print('rim', r, 'cells', lsd._vol(r), 'ms', round(t(r), 3))
