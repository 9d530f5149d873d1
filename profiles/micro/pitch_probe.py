import sys, time, torch
sys.path.insert(0, '.')
from levelsetfortran_amd import distributed as lsd, fields
dev = torch.device('cuda', 0)
N = 1024
b0 = lsd.make_block(0, (2, 2, 2), (N - 1, N - 1, N - 1))
be = lsd.HipBackend(dev)
dx = 3.0 / (N - 1); h = fields.reinit_step(dx)
for pitch in (515, 516, 520, 528, 512+16*2):
    b = lsd.Block(b0.dims, b0.coords, b0.n, b0.own, b0.g0, (pitch, b0.ext[1], b0.ext[2]))
    n = b.npoints_local()
    a = torch.rand(n, dtype=torch.float64, device=dev) * 0.1
    out = torch.empty_like(a); ps = a.clone(); ss = torch.zeros(8, dtype=torch.float64, device=dev)
    core, rims = lsd.sweep_regions(b)
    be.sweep(a, out, ps, b, core, dx, h, ss, be.compute); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): be.sweep(a, out, ps, b, core, dx, h, ss, be.compute)
    torch.cuda.synchronize(); print('pitch', pitch, 'core ms', round((time.perf_counter() - t0) / 5 * 1e3, 3))
