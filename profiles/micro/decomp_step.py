"""Compute-side cost of ONE decomposed sweep of rank (0,0,0) of a 2x2x2 decomposition (512^3 owned + ghosts) on one GPU:
the real DistributedReinit.sweep with the message exchange replaced by its pack / unpack kernels only (no peers here).
Tells how far the block-decomposed sweep is from the single-domain sweep before any network time."""
import sys, time, torch
sys.path.insert(0, '.')
from levelsetfortran_amd import distributed as lsd, fields
dev = torch.device('cuda', 0)
for dtype in ('f64', 'f32'):
    N = 1024
    b = lsd.make_block(0, (2, 2, 2), (N - 1, N - 1, N - 1))
    be = lsd.HipBackend(dev, dtype=dtype)
    dx = 3.0 / (N - 1); h = fields.reinit_step(dx)
    dr = lsd.DistributedReinit(be, b, dx, h)
    def fake_exchange(f):  # pack + unpack on the comm stream, no transport
        be.wait(be.comm, be.compute)
        with be.stream_ctx(be.comm):  # as DistributedReinit.exchange does: all face slabs in one launch each way
            be.pack_all(f, b, [p[1] for p in dr.plan], dr.send_bufs, be.comm)
            be.unpack_all(f, b, [p[2] for p in dr.plan], dr.recv_bufs, be.comm)
    dr.exchange = fake_exchange
    n = b.npoints_local()
    # the bench's own field (smooth two-sphere distance), not noise: these kernels run power-limited, and a field of random
    # mantissas costs 5 % (fp64) to 13 % (fp32) more time for the same arithmetic (profiles/micro/core_probe.py)
    rng = tuple((g, g + e) for g, e in zip(b.g0, b.ext))
    a = fields.two_sphere_phi0_device((N, N, N), dev, ranges=rng)[0].to(be.dtype)
    bufs = [a, a.clone()]; ps = a.clone()
    def steps(k):
        for s in range(k):
            dr.sweep(bufs[s & 1], bufs[(s + 1) & 1], ps); dr.rms_async()
    steps(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); steps(16); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 16 * 1e3
    # the single-domain sweep of the same number of owned cells
    import levelsetfortran_amd as lsf
    M = 512
    f = fields.two_sphere_phi0_device((M, M, M), dev)[0].to(be.dtype)
    lsf.reinit(f, None, None, M - 1, M - 1, M - 1, 3, dx, h, tol=0.0, order='jacobi'); torch.cuda.synchronize()
    t0 = time.perf_counter(); lsf.reinit(f, None, None, M - 1, M - 1, M - 1, 15, dx, h, tol=0.0, order='jacobi'); torch.cuda.synchronize()
    ms1 = (time.perf_counter() - t0) / 16 * 1e3
    print(dtype, 'decomposed step (rank 0 of 2x2x2, 512^3 owned):', round(ms, 3), 'ms; single-domain 512^3 step:', round(ms1, 3), 'ms')
