"""Compute side of a decomposed sweep, piece by piece: full | nopack | coreonly | single (see the code)."""
import sys, time, torch, os
sys.path.insert(0, '.')
from levelsetfortran_amd import distributed as lsd, fields
import levelsetfortran_amd as lsf
dev = torch.device('cuda', 0)
which = sys.argv[1]
for dtype in ('f64', 'f32'):
    N = 1024
    b = lsd.make_block(0, (2, 2, 2), (N - 1, N - 1, N - 1))
    if which == 'single':  # one block = the whole 512^3 grid: the same box call on a box without ghost layers
        N = 512
        b = lsd.make_block(0, (1, 1, 1), (N - 1, N - 1, N - 1))
    be = lsd.HipBackend(dev, dtype=dtype)
    dx = 3.0 / (N - 1); h = fields.reinit_step(dx)
    dr = lsd.DistributedReinit(be, b, dx, h)
    def fake_exchange(f):
        if which == 'nopack': return
        be.wait(be.comm, be.compute)
        with be.stream_ctx(be.comm):
            for (peer, s_box, _r, _a, _s), sb in zip(dr.plan, dr.send_bufs): be.pack(f, b, s_box, sb, be.comm)
            for (peer, _s, r_box, _a, _sd), rb in zip(dr.plan, dr.recv_bufs): be.unpack(f, b, r_box, rb, be.comm)
    dr.exchange = fake_exchange
    n = b.npoints_local()
    a = (torch.rand(n, dtype=torch.float64, device=dev) * 0.1).to(be.dtype)
    if os.environ.get('PROBE_SMOOTH'):  # a smooth field instead of noise (same arithmetic, fewer toggling bits)
        rng = tuple((g, g + e) for g, e in zip(b.g0, b.ext))
        a = fields.two_sphere_phi0_device((N, N, N), dev, ranges=rng)[0].to(be.dtype)
    bufs = [a, a.clone()]; ps = a.clone()
    ss = be.zeros(1)
    def steps(k):
        for s in range(k):
            if which in ('coreonly', 'single'):
                be.sweep(bufs[s & 1], bufs[(s + 1) & 1], ps, b, dr.core, dx, h, ss, be.compute)
            else:
                dr.sweep(bufs[s & 1], bufs[(s + 1) & 1], ps); dr.rms_async()
    steps(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); steps(16); torch.cuda.synchronize()
    print(which, dtype, os.environ.get('LSF_JAC_SH'), round((time.perf_counter() - t0) / 16 * 1e3, 3), 'ms', 'core', dr.core)
