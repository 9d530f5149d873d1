"""Per-rank model of north_star's table for the block-decomposed Jacobi sweep: fixed global grids 256^3, 512^3, 1024^3 on 2, 4
and 8 ranks, from what ONE GPU can measure.

For every (G, R): the decomposition bench.py uses (levelsetfortran_amd.distributed.default_dims), the rank with the most
neighbours, and
  device   the compute side of one of ITS sweeps measured here: the real DistributedReinit.sweep with the message exchange
           replaced by its pack and unpack kernels (core on the compute stream || pack / unpack on the communication stream,
           then rims, BC, block sum), 32 sweeps after 8;
  host     the wall time of ENQUEUEING a sweep (Python + ctypes + HIP launches) when the device is not waited for;
  xfer     the largest face slab of the rank / 153 GB/s (one xGMI link per neighbour, MI355X_MICROARCH / the task brief);
  T1       the single-domain Jacobi sweep of the whole G^3 grid on one GPU (same run).
Predicted ms per sweep: max(device, host) if the transfer hides behind the core kernel, + xfer if it does not (both are
printed); efficiency = T1 / (R x predicted).  The binding term is named.  No transport runs here: RCCL and peer copies
between distinct devices stay unmeasured (one GPU per lease).

  python profiles/micro/jacobi_rank_model.py [grids=256,512,1024] [ranks=2,4,8] [dtype=f64]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
import levelsetfortran_amd as lsf  # noqa: E402
from levelsetfortran_amd import distributed as lsd, fields  # noqa: E402

kv = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
grids = [int(v) for v in kv.get("grids", "256,512,1024").split(",")]
ranks = [int(v) for v in kv.get("ranks", "2,4,8").split(",")]
dtype = kv.get("dtype", "f64")
LINK_GBS = 153.0
dev = torch.device("cuda", 0)
esz = 8 if dtype == "f64" else 4
print(f"# dtype {dtype}; xGMI link {LINK_GBS} GB/s; times in ms per sweep")
print("# G  R  dims   local block        T1(G)   device  host    xfer   predicted(hidden / exposed)  efficiency   binding term")
for G in grids:
    dxg = 3.0 / (G - 1)
    h = fields.reinit_step(dxg)
    f = fields.two_sphere_phi0_device((G, G, G), dev)[0]
    if dtype == "f32":
        f = f.to(torch.float32)
    lsf.reinit(f, None, None, G - 1, G - 1, G - 1, 7, dxg, h, tol=0.0, order="jacobi")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lsf.reinit(f, None, None, G - 1, G - 1, G - 1, 31, dxg, h, tol=0.0, order="jacobi")
    torch.cuda.synchronize()
    T1 = (time.perf_counter() - t0) / 32 * 1e3
    del f
    for R in ranks:
        dims = lsd.default_dims(R)
        # the rank with the most neighbours (an interior one if there is one)
        best, bb = -1, None
        for r in range(R):
            b = lsd.make_block(r, dims, (G - 1, G - 1, G - 1))
            nn = sum(1 for ax in range(3) for sd in (0, 1) if b.neighbour(ax, sd) is not None)
            if nn > best:
                best, bb = nn, b
        b = bb
        be = lsd.HipBackend(dev, dtype=dtype)
        dr = lsd.DistributedReinit(be, b, dxg, h)

        def fake_exchange(fld, dr=dr, be=be, b=b):  # pack + unpack on the comm stream, no transport
            be.wait(be.comm, be.compute)
            with be.stream_ctx(be.comm):
                be.pack_all(fld, b, [p[1] for p in dr.plan], dr.send_bufs, be.comm)
                be.unpack_all(fld, b, [p[2] for p in dr.plan], dr.recv_bufs, be.comm)

        dr.exchange = fake_exchange
        rng = tuple((g, g + e) for g, e in zip(b.g0, b.ext))
        a = fields.two_sphere_phi0_device((G, G, G), dev, ranges=rng)[0].to(be.dtype)
        bufs = [a, a.clone()]
        ps = a.clone()

        def steps(k):
            for s in range(k):
                dr.sweep(bufs[s & 1], bufs[(s + 1) & 1], ps)
                dr.rms_async()

        steps(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps(32)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        device_ms = t_all / 32 * 1e3
        # host: enqueue time when the device is far behind is the host's own cost; when the queue fills the runtime blocks, so
        # measure it on a short burst as well and take the smaller
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps(4)
        host_ms = min(t_enq / 32, (time.perf_counter() - t0) / 4) * 1e3
        torch.cuda.synchronize()
        face = max((sb.numel() * esz for sb in dr.send_bufs), default=0)
        xfer_ms = face / (LINK_GBS * 1e9) * 1e3
        hidden, exposed = max(device_ms, host_ms), max(device_ms, host_ms) + xfer_ms
        term = "host enqueue" if host_ms > device_ms else ("device: compute side of the rank" if xfer_ms < 0.25 * device_ms else "device + transfer")
        print(f"{G:5d} {R}  {'x'.join(map(str, dims))}  {'x'.join(map(str, b.ext)):>16s}  {T1:7.3f} {device_ms:7.3f} {host_ms:6.3f} {xfer_ms:6.3f}   "
              f"{hidden:6.3f} / {exposed:6.3f}          {T1 / (R * hidden) * 100:4.0f} % / {T1 / (R * exposed) * 100:4.0f} %   {term}", flush=True)
        del a, bufs, ps, dr, be
        torch.cuda.empty_cache()
