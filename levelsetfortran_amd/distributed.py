"""Block-decomposed Jacobi reinitialisation: one process per GPU, 3-cell face halos over
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The reference is serial (README.md:17): nothing here mirrors reference code.  What must hold is
that the decomposed sweep produces exactly the field of the single-domain Jacobi sweep
(include/lsf.h LSF_ORDER_JACOBI), because every cell update reads the same 19 values.  The exact
Gauss-Seidel ordering does not shard and is not offered here.

Per sweep (DESIGN.md "multi-GPU"):
  comm stream   : pack owned face slabs (3 thick) -> isend/irecv with up to 6 face neighbours ->
                  unpack into the ghost layers.  Star stencil: faces only, no edges or corners.
  compute stream: interior cells (>= 3 from every cut face) concurrently with the exchange,
                  then the rim slabs, then the extrapolation BC on the owned wall points.
  RMS           : device partial sums -> one all_reduce(SUM) of a double per sweep.

All arithmetic goes through a backend object:
  HipBackend   -> liblsf_hip.so (lsf_jacobi_sweep_box / lsf_bc_box / lsf_pack_box / lsf_unpack_box)
The CPU tests inject an oracle-backed backend from tests/ (the package itself never imports the
oracle), which exercises this file's decomposition, halo schedule and reduction unchanged.
"""
from __future__ import annotations

import contextlib
import ctypes
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

HALO = 3  # WENO5 reaches +-3 (subs.f90:509-530)


# ------------------------------------------------------------------------------------------------
# decomposition (pure index arithmetic)
# ------------------------------------------------------------------------------------------------
def default_dims(world: int) -> Tuple[int, int, int]:
    """(Px, Py, Pz).  BASELINE.json configs: 4 GPUs -> a 2x2x1 decomposition, 8 GPUs -> 2x2x2.  The unit-stride axis x is
    cut LAST (2 ranks: z; 4 ranks: y and z): a cut across z or y leaves face slabs made of whole rows (contiguous packs,
    rims the sweep kernel covers with full wavefronts), a cut across x makes 3-cell-wide rims and 24-byte pack runs.
    Otherwise the prime factors are dealt to z, y, x in turn."""
    table = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}
    if world in table:
        return table[world]
    dims = [1, 1, 1]
    n, a = world, 0
    for p in range(2, world + 1):
        while n % p == 0:
            dims[2 - a % 3] *= p
            n //= p
            a += 1
    return tuple(dims)


def split_points(npoints: int, parts: int) -> List[Tuple[int, int]]:
    """Contiguous balanced split of point indices 0..npoints-1 into `parts` ranges [s,e)."""
    base, extra = divmod(npoints, parts)
    out, s = [], 0
    for p in range(parts):
        e = s + base + (1 if p < extra else 0)
        out.append((s, e))
        s = e
    return out


@dataclass
class Block:
    """What one rank holds of the global field (0:nx,0:ny,0:nz)."""

    dims: Tuple[int, int, int]  # process grid
    coords: Tuple[int, int, int]  # this rank's position
    n: Tuple[int, int, int]  # global (nx,ny,nz)
    own: Tuple[Tuple[int, int], ...]  # owned global point range per axis [s,e)
    g0: Tuple[int, int, int]  # global index of local point 0 (own start minus ghost)
    ext: Tuple[int, int, int]  # local extents including ghosts

    @property
    def own_local(self):
        return tuple((s - g, e - g) for (s, e), g in zip(self.own, self.g0))

    def neighbour(self, axis: int, side: int) -> Optional[int]:
        c = list(self.coords)
        c[axis] += side
        if c[axis] < 0 or c[axis] >= self.dims[axis]:
            return None
        return rank_of(tuple(c), self.dims)

    def npoints_local(self) -> int:
        return self.ext[0] * self.ext[1] * self.ext[2]


def rank_of(coords, dims) -> int:
    return coords[0] + dims[0] * (coords[1] + dims[1] * coords[2])


def coords_of(rank, dims):
    return (rank % dims[0], (rank // dims[0]) % dims[1], rank // (dims[0] * dims[1]))


def make_block(rank: int, dims, n) -> Block:
    coords = coords_of(rank, dims)
    own, g0, ext = [], [], []
    for a in range(3):
        s, e = split_points(n[a] + 1, dims[a])[coords[a]]
        if e - s < 2 * HALO and dims[a] > 1:
            raise ValueError(f"axis {a}: {e - s} owned points per rank is fewer than 2*HALO")
        lo = s - HALO if coords[a] > 0 else s
        hi = e + HALO if coords[a] < dims[a] - 1 else e
        own.append((s, e))
        g0.append(lo)
        ext.append(hi - lo)
    return Block(tuple(dims), coords, tuple(n), tuple(own), tuple(g0), tuple(ext))


def interior_cells_local(b: Block):
    """Owned cells that are interior cells of the global grid (1..n-1), as local [lo,hi) per axis."""
    out = []
    for a in range(3):
        s, e = b.own[a]
        out.append((max(s, 1) - b.g0[a], min(e, b.n[a]) - b.g0[a]))
    return out


def small_block(b: Block) -> bool:
    """Blocks below LSF_MULTI_SMALL (default 192, 0 = never) owned points along their longest axis run a sweep as ONE launch
    after the exchange instead of core || exchange + rims: at 131^3 per rank (north_star's 256^3 on eight GPUs) a sweep is
    0.03 ms of arithmetic under launches whose fixed costs are the time (profiles/r04_jacobi_rank_model.txt).  Same rule as
    csrc/lsf_multi.hpp."""
    import os

    thr = int(os.environ.get("LSF_MULTI_SMALL", "192"))
    return max(e - s for s, e in b.own) < thr


def sweep_regions(b: Block):
    """(core, rims): core needs no ghost data; the rims (disjoint boxes) need the halo exchange."""
    cells = interior_cells_local(b)
    if small_block(b):
        return [(c[0], c[0]) for c in cells], [[tuple(c) for c in cells]]
    core = []
    for a in range(3):
        lo, hi = cells[a]
        if b.coords[a] > 0:
            lo += HALO
        if b.coords[a] < b.dims[a] - 1:
            hi -= HALO
        core.append((lo, max(hi, lo)))
    rims = []
    cur = [list(c) for c in cells]  # shrinking box; peel one axis at a time so rims are disjoint
    for a in range(3):
        if core[a][0] > cur[a][0]:
            r = [tuple(c) for c in cur]
            r[a] = (cur[a][0], core[a][0])
            rims.append(r)
            cur[a][0] = core[a][0]
        if core[a][1] < cur[a][1]:
            r = [tuple(c) for c in cur]
            r[a] = (core[a][1], cur[a][1])
            rims.append(r)
            cur[a][1] = core[a][1]
    return core, rims


def halo_plan(b: Block):
    """[(peer, send_box, recv_box)] in local indices; cross-sections are the owned ranges."""
    plan = []
    ol = b.own_local
    for a in range(3):
        for side in (-1, +1):
            peer = b.neighbour(a, side)
            if peer is None:
                continue
            send = [tuple(r) for r in ol]
            recv = [tuple(r) for r in ol]
            if side < 0:
                send[a] = (ol[a][0], ol[a][0] + HALO)
                recv[a] = (ol[a][0] - HALO, ol[a][0])
            else:
                send[a] = (ol[a][1] - HALO, ol[a][1])
                recv[a] = (ol[a][1], ol[a][1] + HALO)
            plan.append((peer, send, recv, a, side))
    return plan


# ------------------------------------------------------------------------------------------------
# HIP backend
# ------------------------------------------------------------------------------------------------
class HipBackend:
    """Device arithmetic through the C ABI; tensors are 1-D torch CUDA float64, i fastest."""

    def __init__(self, device, arith: str = "fast", host_staging: bool = False, dtype: str = "f64"):
        import torch

        from . import _lib

        # host_staging: route the halo messages through pinned host buffers (for transports that cannot
        # read device memory, e.g. gloo in the single-GPU tests); RCCL moves device buffers directly
        self.host_staging = host_staging
        self.torch = torch
        self.L = _lib
        self.lib = _lib.load()
        self.device = device
        self.mode = _lib.LSF_ORDER_JACOBI | (_lib.LSF_ARITH_STRICT if arith == "strict" else _lib.LSF_ARITH_FAST)
        # dtype "f32": the single-precision twins of the four building blocks (BASELINE configuration 5)
        if dtype not in ("f64", "f32"):
            raise ValueError("dtype must be 'f64' or 'f32'")
        self.dtype = torch.float32 if dtype == "f32" else torch.float64
        sfx = "_f32" if dtype == "f32" else ""
        self._sweep = getattr(self.lib, "lsf_jacobi_sweep_box" + sfx)
        self._bc = getattr(self.lib, "lsf_bc_box" + sfx)
        self._pack = getattr(self.lib, "lsf_pack_box" + sfx)
        self._unpack = getattr(self.lib, "lsf_unpack_box" + sfx)
        self._pack_all = getattr(self.lib, "lsf_pack_boxes" + sfx)
        self._unpack_all = getattr(self.lib, "lsf_unpack_boxes" + sfx)
        _lib.check(self.lib.lsf_set_device(device.index or 0))
        self.compute = torch.cuda.current_stream(device)
        self.comm = torch.cuda.Stream(device)

    def empty(self, n, dtype=None):
        return self.torch.empty(n, dtype=dtype or self.dtype, device=self.device)

    def zeros(self, n):
        # sums of squares are accumulated in double whatever the field type
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.device)

    def from_numpy(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a.ravel(order="F"))).to(self.device).to(self.dtype)

    def to_numpy(self, t, shape):
        return t.cpu().numpy().reshape(shape, order="F")

    def _box(self, b: Block):
        return self.L.LsfBox(b.ext[0], b.ext[1], b.ext[2], b.g0[0], b.g0[1], b.g0[2], b.n[0], b.n[1], b.n[2])

    @staticmethod
    def _lohi(region):
        from ._lib import int3

        return int3([r[0] for r in region]), int3([r[1] for r in region])

    def sweep(self, a_in, a_out, phiS, b, region, dx, h, sumsq, stream):
        lo, hi = self._lohi(region)
        self.L.check(self._sweep(a_in.data_ptr(), a_out.data_ptr(), phiS.data_ptr(),
                                                   ctypes.byref(self._box(b)), lo, hi, dx, h, self.mode,
                                                   sumsq.data_ptr(), stream.cuda_stream))

    def bc(self, a_in, a_out, b, region, dx, sumsq, stream):
        lo, hi = self._lohi(region)
        self.L.check(self._bc(a_in.data_ptr(), a_out.data_ptr(), ctypes.byref(self._box(b)), lo, hi, dx,
                                         sumsq.data_ptr(), stream.cuda_stream))

    def pack(self, f, b, region, buf, stream):
        lo, hi = self._lohi(region)
        self.L.check(self._pack(f.data_ptr(), ctypes.byref(self._box(b)), lo, hi, buf.data_ptr(),
                                           stream.cuda_stream))

    def unpack(self, f, b, region, buf, stream):
        lo, hi = self._lohi(region)
        self.L.check(self._unpack(f.data_ptr(), ctypes.byref(self._box(b)), lo, hi, buf.data_ptr(),
                                             stream.cuda_stream))

    def _multi(self, fn, f, b, regions, bufs, stream):
        n = len(regions)
        lo = (ctypes.c_int * (3 * n))(*[r[a][0] for r in regions for a in range(3)])
        hi = (ctypes.c_int * (3 * n))(*[r[a][1] for r in regions for a in range(3)])
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in bufs])
        self.L.check(fn(f.data_ptr(), ctypes.byref(self._box(b)), n, lo, hi, ptrs, stream.cuda_stream))

    def pack_all(self, f, b, regions, bufs, stream):
        """all face slabs of a block in one launch (include/lsf.h: lsf_pack_boxes)"""
        self._multi(self._pack_all, f, b, regions, bufs, stream)

    def unpack_all(self, f, b, regions, bufs, stream):
        self._multi(self._unpack_all, f, b, regions, bufs, stream)

    def sumsq_begin(self, stream):
        """bracket the box calls of a sweep: their partial sums are reduced once, by sumsq_end (include/lsf.h)"""
        self.L.check(self.lib.lsf_sumsq_begin(stream.cuda_stream))

    def sumsq_end(self, stream):
        self.L.check(self.lib.lsf_sumsq_end(stream.cuda_stream))

    def stream_ctx(self, stream):
        return self.torch.cuda.stream(stream)

    def wait(self, waiter, waited):
        waiter.wait_stream(waited)

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)


# ------------------------------------------------------------------------------------------------
# the distributed sweep loop
# ------------------------------------------------------------------------------------------------
def _vol(region):
    return max(region[0][1] - region[0][0], 0) * max(region[1][1] - region[1][0], 0) * max(region[2][1] - region[2][0], 0)


class DistributedReinit:
    """Jacobi reinit of a block-decomposed field.  `backend` supplies the arithmetic and streams."""

    def __init__(self, backend, block: Block, dx: float, h: float, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.be = backend
        self.b = block
        self.dx, self.h = float(dx), float(h)
        self.group = group
        self.core, self.rims = sweep_regions(block)
        self.plan = halo_plan(block)
        self.send_bufs = [backend.empty(_vol(s)) for (_, s, _, _, _) in self.plan]
        self.recv_bufs = [backend.empty(_vol(r)) for (_, _, r, _, _) in self.plan]
        self.staging = bool(getattr(backend, "host_staging", False))
        if self.staging:
            import torch

            self.send_host = [torch.empty(b.numel(), dtype=b.dtype).pin_memory() for b in self.send_bufs]
            self.recv_host = [torch.empty(b.numel(), dtype=b.dtype).pin_memory() for b in self.recv_bufs]
        self.sumsq = backend.zeros(1)
        # INTEGER*4 product nx*ny*nz of the GLOBAL grid (subs.f90:914), wrapping like the reference
        nx, ny, nz = block.n
        self.den = float(np.int32(np.uint32((nx * ny * nz) & 0xFFFFFFFF)))
        if str(getattr(backend, "dtype", "")).endswith("float32"):
            # fp32 path (BASELINE configuration 5, 1536^3): the wrapped product is negative there; no reference to mirror
            self.den = float(nx) * float(ny) * float(nz)

    # -- halo exchange of field f on the comm stream ------------------------------------------------
    def exchange(self, f):
        be, dist = self.be, self.dist
        if not self.plan:
            return
        be.wait(be.comm, be.compute)  # f was produced on the compute stream
        with be.stream_ctx(be.comm):
            ops = []
            if hasattr(be, "pack_all"):  # the three to six face slabs in one launch
                be.pack_all(f, self.b, [p[1] for p in self.plan], self.send_bufs, be.comm)
            else:
                for (peer, s_box, _r, _a, _s), sb in zip(self.plan, self.send_bufs):
                    be.pack(f, self.b, s_box, sb, be.comm)
            send_bufs, recv_bufs = self.send_bufs, self.recv_bufs
            if self.staging:
                for hb, sb in zip(self.send_host, self.send_bufs):
                    hb.copy_(sb, non_blocking=True)
                be.comm.synchronize()
                send_bufs, recv_bufs = self.send_host, self.recv_host
            for (peer, _s, _r, a, side), sb, rb in zip(self.plan, send_bufs, recv_bufs):
                # tag by (axis, direction of travel) so the two messages between a pair of ranks that are
                # neighbours on both sides of a periodic-free 2-rank axis cannot be confused
                ops.append(dist.P2POp(dist.isend, sb, peer, group=self.group, tag=2 * a + (0 if side < 0 else 1)))
                ops.append(dist.P2POp(dist.irecv, rb, peer, group=self.group, tag=2 * a + (1 if side < 0 else 0)))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if self.staging:
                for hb, rb in zip(self.recv_host, self.recv_bufs):
                    rb.copy_(hb, non_blocking=True)
            if hasattr(be, "unpack_all"):
                be.unpack_all(f, self.b, [p[2] for p in self.plan], self.recv_bufs, be.comm)
            else:
                for (peer, _s, r_box, _a, _sd), rb in zip(self.plan, self.recv_bufs):
                    be.unpack(f, self.b, r_box, rb, be.comm)

    # -- one sweep: a_in -> a_out; returns nothing; self.sumsq accumulates the local sum of squares ---
    def sweep(self, a_in, a_out, phiS):
        be, b = self.be, self.b
        self.sumsq.zero_()
        self.exchange(a_in)  # ghosts of a_in (comm stream)
        bracket = hasattr(be, "sumsq_begin")  # one reduction of the partial sums per sweep instead of one per region
        if bracket:
            be.sumsq_begin(be.compute)
        try:
            if _vol(self.core) > 0:
                be.sweep(a_in, a_out, phiS, b, self.core, self.dx, self.h, self.sumsq, be.compute)  # overlaps the exchange
            be.wait(be.compute, be.comm)
            for r in self.rims:
                if _vol(r) > 0:
                    be.sweep(a_in, a_out, phiS, b, r, self.dx, self.h, self.sumsq, be.compute)
            be.bc(a_in, a_out, b, [tuple(r) for r in b.own_local], self.dx, self.sumsq, be.compute)
        finally:
            if bracket:
                be.sumsq_end(be.compute)

    def rms_async(self):
        """all_reduce of the sum of squares; returns a 1-element tensor holding the global sum."""
        if self.dist.is_initialized() and self.dist.get_world_size(self.group) > 1:
            self.dist.all_reduce(self.sumsq, op=self.dist.ReduceOp.SUM, group=self.group)
        return self.sumsq

    def run(self, phi, iter: int, tol: float = 1.0e-5, check_every: int = 8):
        """reinit semantics (subs.f90:735-928) on the decomposed field: at most iter+1 sweeps, stop when
        RMS < tol.  phi is this rank's local box (ghost layers included); returns (result, sweeps, rms list).

        The host looks at the RMS once per window of `check_every` sweeps, one window late ("may be
        checked ... late"): inside a window it only enqueues, the block sums of the window's sweeps go into one device
        vector, ONE all_reduce and ONE device-to-host read per window.  A run that can stop (tol > 0) keeps the field at the
        start of the last two windows; when a window turns out to hold the stop sweep, the sweeps from its start to that
        sweep are run again: result, sweep count and trace are those of a driver that looks after every sweep (the same
        scheme as lsf_multi_run, csrc/lsf_multi.hpp).
        """
        from ._lib import LsfNaNError

        K = max(1, min(int(check_every), 64))
        phiS = phi.clone()
        bufs = [phi, phi.clone()]
        keep = tol > 0.0
        snaps = [None, None]
        rms_hist = []
        live = self.dist.is_initialized() and self.dist.get_world_size(self.group) > 1

        def enqueue_window(c0, n, w):
            sums = self.be.zeros(K)
            if keep:
                if snaps[w & 1] is None:
                    snaps[w & 1] = bufs[c0 & 1].clone()
                else:
                    snaps[w & 1].copy_(bufs[c0 & 1])
            for j in range(n):
                s = c0 + j
                self.sweep(bufs[s & 1], bufs[(s + 1) & 1], phiS)
                sums[j:j + 1].copy_(self.sumsq)  # compute stream: behind the sweep's reduction
            if live:
                self.dist.all_reduce(sums, op=self.dist.ReduceOp.SUM, group=self.group)
            return sums

        def judge(sums, c0, n):
            """first sweep of the window that ends the run, or None"""
            vals = sums[:n].tolist()  # the one host read of the window
            for j, v in enumerate(vals):
                q = float(v) / self.den  # the wrapped INTEGER*4 product is negative for some grids:
                rms = math.sqrt(q) if q >= 0 else float("nan")  # NaN like the single-domain path (and the reference)
                rms_hist.append(rms)
                if rms < tol or rms != rms:
                    return c0 + j
            return None

        s, w, stop_at = 0, 0, None
        pending = None  # (sums, c0, n) of the window enqueued last
        max_sweeps = iter + 1
        while s < max_sweeps and stop_at is None:
            c0, n = s, min(K, max_sweeps - s)
            cur = (enqueue_window(c0, n, w), c0, n)
            s += n
            if pending is not None:
                stop_at = judge(*pending)  # one window late: window w is already in the queues
            pending = cur
            w += 1
        if stop_at is None and pending is not None:
            stop_at = judge(*pending)
        self.be.synchronize()
        nsw = stop_at + 1 if stop_at is not None else len(rms_hist)
        if stop_at is not None and nsw < s and keep:
            # sweeps beyond the stop sweep have overwritten both buffers: back to the start of its window, repeat up to it
            wv, c0 = stop_at // K, (stop_at // K) * K
            bufs[c0 & 1].copy_(snaps[wv & 1])
            for t in range(c0, stop_at + 1):
                self.sweep(bufs[t & 1], bufs[(t + 1) & 1], phiS)
            self.be.synchronize()
        if nsw and rms_hist[nsw - 1] != rms_hist[nsw - 1]:
            raise LsfNaNError(1, "RMS became NaN (the reference STOPs here, subs.f90:926)")
        return bufs[nsw & 1], nsw, rms_hist[:nsw]


# ------------------------------------------------------------------------------------------------
def bench_decomposed(global_pts, K: int, W: int, device, arith: str = "fast", dtype: str = "f64", shared_gpu: bool = False,
                     dims=None):
    """K timed sweeps (after W) of the block-decomposed Jacobi sweep on a grid of global_pts = (Nx, Ny, Nz) POINTS split
    over the ranks of the job (default_dims: 2x2x1 on 4 ranks with x the uncut axis, 2x2x2 on 8; blocks need not be cubic:
    BASELINE configuration 4 is 1024^3 on 4 ranks = 1024 x 512 x 512 per rank).  Barrier + synchronize on both sides; the caller takes
    the max over ranks."""
    import time

    import torch
    import torch.distributed as dist

    from . import fields

    live = dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if live else (1, 0)

    def barrier():
        if live:
            if shared_gpu:  # rehearsal on one GPU over gloo (bench.py LSF_BENCH_SHARED_GPU)
                dist.barrier()
            else:
                dist.barrier(device_ids=[device.index])
        torch.cuda.synchronize(device)

    dims = tuple(dims) if dims is not None else default_dims(world)
    gpts = tuple(int(g) for g in global_pts)
    n = tuple(g - 1 for g in gpts)
    b = make_block(rank, dims, n)
    be = HipBackend(device, arith, host_staging=shared_gpu, dtype=dtype)
    rng = tuple((g, g + e) for g, e in zip(b.g0, b.ext))
    phi, dx = fields.two_sphere_phi0_device(gpts, device, ranges=rng)  # built in HBM: no host temporaries
    phi = phi.to(be.dtype)
    h = fields.reinit_step(dx)
    dr = DistributedReinit(be, b, dx, h)
    phiS = phi.clone()
    bufs = [phi, phi.clone()]

    def steps(k, s0):
        for s in range(s0, s0 + k):
            dr.sweep(bufs[s & 1], bufs[(s + 1) & 1], phiS)
            dr.rms_async()
        return s0 + k

    s0 = steps(W, 0)
    barrier()
    t0 = time.perf_counter()
    steps(K, s0)
    barrier()
    dt = time.perf_counter() - t0
    cells = float(n[0] - 1) * (n[1] - 1) * (n[2] - 1) * K
    return {"cells_total": cells, "seconds": dt, "prof": None, "global_grid": list(gpts), "dims": list(dims),
            "local_block": list(b.ext),
            "parallelism": f"{dims[0]}x{dims[1]}x{dims[2]} block decomposition, 3-cell face halos over RCCL (xGMI), "
                           f"halo exchange overlapped with interior cells on a second HIP stream"}


def bench_weak_scaling(N: int, K: int, W: int, device, arith: str = "fast", dtype: str = "f64", shared_gpu: bool = False):
    """every rank owns an N^3-point block of a (Px N, Py N, Pz N) grid"""
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    dims = default_dims(world)
    return bench_decomposed(tuple(d * N for d in dims), K, W, device, arith=arith, dtype=dtype, shared_gpu=shared_gpu, dims=dims)


def parity_decomposed(global_pts, sweeps: int, device, arith: str = "fast", dtype: str = "f64", shared_gpu: bool = False, dims=None,
                      rtol: float = 1.0e-11, sabotage: bool = False):
    """Evidence, inside the job that is being timed, that the decomposed sweep IS the single-domain sweep (the call site the
    decomposition stands in for: set3d.f90:308).  Every rank runs `sweeps` sweeps of its block of a global_pts grid through
    DistributedReinit.run (halo exchange, rims, BC, one reduction per window: the timed path), rank 0 runs the same sweeps on the
    whole grid on its own GPU (lsf_reinit_device, Jacobi ordering, same arithmetic) and compares
      * the field: SHA-256 of every rank's OWNED points against the same points of the single-domain field, and
      * the RMS trace (block sums added in rank order against one fixed-order sum: relative tolerance `rtol`).
    Never inside a timed region.  Returns the record on rank 0, None elsewhere; collective (every rank must call it).
    sabotage (test aid, bench.py LSF_BENCH_PARITY_SABOTAGE=1): the last rank moves ONE owned value by one unit in the last place
    before the comparison -- the record must say so and the job must fail."""
    import hashlib

    import torch
    import torch.distributed as dist

    from . import fields
    from .levelset import reinit

    live = dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if live else (1, 0)
    dims = tuple(dims) if dims is not None else default_dims(world)
    gpts = tuple(int(g) for g in global_pts)
    n = tuple(g - 1 for g in gpts)
    b = make_block(rank, dims, n)
    be = HipBackend(device, arith, host_staging=shared_gpu, dtype=dtype)
    rng = tuple((g, g + e) for g, e in zip(b.g0, b.ext))
    phi, dx = fields.two_sphere_phi0_device(gpts, device, ranges=rng)
    phi = phi.to(be.dtype)
    h = fields.reinit_step(dx)
    out, nsw, rms = DistributedReinit(be, b, dx, h).run(phi, sweeps - 1, tol=0.0)

    def digest(t, ext, box):
        v = t.reshape(ext[2], ext[1], ext[0])[box[2][0]:box[2][1], box[1][0]:box[1][1], box[0][0]:box[0][1]]
        return hashlib.sha256(v.contiguous().cpu().numpy().tobytes()).hexdigest()

    if sabotage and rank == world - 1:
        ol = b.own_local
        at = (ol[0][0] + 1) + b.ext[0] * ((ol[1][0] + 1) + b.ext[1] * (ol[2][0] + 1))
        out[at:at + 1] = torch.nextafter(out[at:at + 1], out[at:at + 1] + 1)
    mine = digest(out, b.ext, b.own_local)
    got = [mine]
    if live and world > 1:
        got = [None] * world
        dist.all_gather_object(got, mine)
    if rank != 0:
        return None
    whole, _ = fields.two_sphere_phi0_device(gpts, device)
    whole = whole.to(be.dtype)
    rep = reinit(whole, None, None, n[0], n[1], n[2], sweeps - 1, dx, h, tol=0.0, order="jacobi", arith=arith)
    want = [digest(whole, gpts, make_block(r, dims, n).own) for r in range(world)]
    field_ok = got == want
    worst = max((abs(a - c) / abs(c) if c else abs(a - c)) for a, c in zip(rms, rep.rms)) if rms else 0.0
    trace_ok = nsw == rep.count == sweeps and len(rms) == len(rep.rms) and worst <= rtol
    return {"ok": bool(field_ok and trace_ok), "field_sha_equal": bool(field_ok), "rms_trace_equal": bool(trace_ok),
            "rms_trace_max_rel_diff": worst, "rms_trace_rtol": rtol, "grid": list(gpts), "dims": list(dims), "sweeps": sweeps,
            "ranks": world, "arith": arith, "dtype": dtype, "blocks_differing": [r for r in range(world) if got[r] != want[r]],
            "against": "rank 0's single-domain Jacobi sweep of the same grid and sweeps (lsf_reinit_device), outside the timed region"}
