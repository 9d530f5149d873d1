"""Worker of tests/test_gpu_two_devices.py::test_two_nccl_ranks_equal_single_domain: run under torch.distributed.run, ONE RANK PER
GPU, backend nccl (= RCCL over xGMI on MI355X) -- the transport bench.py --gpus N uses, which a one-GPU box cannot start (RCCL
refuses two ranks on one device).  Every rank runs DistributedReinit on its block of the global field with device-resident halo
messages (no host staging) and writes its owned points; the parent compares them with the single-domain sweep."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    outdir, dims, npts, sweeps, arith, dtype = sys.argv[1], eval(sys.argv[2]), eval(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    import torch
    import torch.distributed as dist

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    from levelsetfortran_amd import distributed as D, fields

    n = tuple(v - 1 for v in npts)
    b = D.make_block(rank, dims, n)
    rng = tuple((g, g + e) for g, e in zip(b.g0, b.ext))
    phi, dx = fields.two_sphere_phi0_device(npts, dev, ranges=rng)
    be = D.HipBackend(dev, arith=arith, dtype=dtype)  # RCCL moves device buffers: no host staging
    phi = phi.to(be.dtype)
    h = fields.reinit_step(dx)
    out, nsw, rms = D.DistributedReinit(be, b, dx, h).run(phi, sweeps - 1, tol=0.0, check_every=3)
    own = tuple(slice(lo, hi) for lo, hi in b.own_local)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), own=np.array(b.own), data=be.to_numpy(out, b.ext)[own], nsw=nsw, rms=np.array(rms),
             device=local, backend=dist.get_backend(), world=world)
    # the in-run parity record bench.py attaches to its decomposed entries, over the same communicator
    rec = D.parity_decomposed((64, 64, 64), 8, dev, arith=arith, dtype=dtype, dims=dims)
    if rank == 0:
        import json

        json.dump(rec, open(os.path.join(outdir, "parity.json"), "w"))
    dist.barrier(device_ids=[local])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
