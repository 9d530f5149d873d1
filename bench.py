#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X:

    cell-updates/s (WENO5 reinit, 512^3 fp64); achieved HBM GB/s vs peak

A "step" is one reinitialisation sweep (raster sweep of the WENO5/Godunov/Euler update over all
(N-2)^3 interior cells + extrapolation BC + RMS/stop test; subs.f90:735-928) of a 512^3 fp64 field
that is already resident in HBM.  The default ordering is the reference's own in-place Gauss-Seidel
ordering, reproduced exactly on the GPU (LSF_ORDER_GS): that is the path whose output matches the
reference to the last bit (STRICT arithmetic) / to ~1e-16 (FAST arithmetic, used here).  The
double-buffered Jacobi ordering (does not reproduce the reference field; shards across GPUs) is
measured in the same run and reported under "jacobi"; the reference's own arithmetic (LSF_ARITH_STRICT, bit-identical
results, 2 x slower) under "strict_arithmetic".

    python bench.py --gpus N --steps K --warmup W [--mode gs|jacobi] [--size 512] [--dtype f64|f32]

--dtype f32 (BASELINE configuration 5; Jacobi ordering only, 12 B per cell-update) is a secondary measurement:
the headline line is the default fp64 run.

N > 1 is launched by torch.distributed.run, one rank per GPU:
  mode gs      (default) `value` = the path that COMMUNICATES: the block-decomposed Jacobi sweep on N blocks of --size^3, 3-cell
               halos over RCCL ("scaling": "weak"; `metric`, `config` and "headline_path" say so; "same_path_one_gpu" is the
               single-domain Jacobi sweep of one block on one GPU of the job).  N independent replicas of the exact ordering
               -- linear by construction -- are kept as "replicas_gs".  In the same job, under "decomposed": FIXED global
               grids (256^3, 512^3, 1024^3 fp64: "strong", north_star's table), the one-process C-ABI driver (lsf_multi_*)
               run by rank 0 on 1, 2, 4, ... N devices, and the exact ordering over z slabs (reference-equal, sharded),
               each entry stating its own scaling.  If the decomposed measurement fails the replicas stay the headline.
  mode jacobi  the decomposed sweep is the headline: --global G fixes the global grid at G^3 points ("strong": BASELINE
               configuration 4 = --gpus 4 --global 1024, configuration 5 = --gpus 8 --global 1536 --dtype f32); without
               --global every rank owns a --size^3 block ("weak")

Prints ONE JSON line on rank 0.  Exit status is non-zero when a requested decomposed measurement failed or timed out
(the headline line is still printed first).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # vendor vector fp64 (FMA counted as 2)
FP32_VALU_PEAK_TFLOPS = 157.3  # vendor vector fp32 (packed FMA counted as 4 per lane-issue)
BYTES_PER_CELL_UPDATE = 24.0  # read phi, read phiS, write phi


# ------------------------------------------------------------------------------------------------
# CPU baseline worker (separate process: the Fortran runtime of the reference prints one line per
# sweep on stdout, which must not reach the JSON line)
# ------------------------------------------------------------------------------------------------
# the device code of the sweep kernels whose counters profiles/traffic.json holds (host-side files do not change a kernel instance)
KERNEL_SOURCES = ("lsf_cell.hpp", "lsf_kernels.hpp", "lsf_boxtile.hpp", "lsf_skew.hpp", "lsf_f32.hpp")


def kernel_sources_fingerprint():
    """sha256 (16 hex digits) of the sweep kernels' sources as they lie in the tree: profiles/ships_summarize.py stamps the counters it
    condenses with it, and the counters are attached to a bench line only while those sources are the ones they were measured on"""
    import hashlib

    hsh = hashlib.sha256()
    for name in KERNEL_SOURCES:
        hsh.update(name.encode())
        hsh.update(open(os.path.join(ROOT, "levelsetfortran_amd", "csrc", name), "rb").read())
    return hsh.hexdigest()[:16]


def _cpu_worker(out_path: str, n: int, slab: int, sweeps: int) -> None:
    import threading

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from levelsetfortran_amd import fields

    # the sample: a z-slab of `slab` points through the centre of the same n^3 two-sphere phi0
    phi_full_x, dx = None, 3.0 / (n - 1)
    x = -1.5 + dx * np.arange(n)
    k0 = n // 2 - slab // 2
    z = x[k0:k0 + slab]
    d = None
    for c in ((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)):
        r = np.sqrt((x[:, None, None] - c[0]) ** 2 + (x[None, :, None] - c[1]) ** 2 + (z[None, None, :] - c[2]) ** 2)
        d = r - 0.5 if d is None else np.minimum(d, r - 0.5)
    phi = np.asfortranarray(d / np.sqrt(d * d + dx * dx))
    del phi_full_x
    nx, ny, nz = n - 1, n - 1, slab - 1
    h = fields.reinit_step(dx)
    res = {}

    ref_so = os.path.join(ROOT, "oracle", "_ref", "libref_subs.so")

    def run():
        t0 = time.perf_counter()
        if os.path.exists(ref_so):
            L = ctypes.CDLL(ref_so)
            f = L._QMset_subsPreinit  # SUBROUTINE reinit, subs.f90:717, as compiled by amdflang
            f.restype = None
            f.argtypes = [ctypes.c_void_p] * 9
            gp = np.zeros(phi.shape + (3,), order="F")
            gm = np.zeros(phi.shape, order="F")
            keep = [ctypes.c_int(nx), ctypes.c_int(ny), ctypes.c_int(nz), ctypes.c_int(sweeps - 1), ctypes.c_double(dx),
                    ctypes.c_double(h)]
            t0 = time.perf_counter()
            f(phi.ctypes.data, gp.ctypes.data, gm.ctypes.data, *[ctypes.addressof(v) for v in keep])
            res["kind"] = "reference"
        else:
            import oracle_lib

            t0 = time.perf_counter()
            oracle_lib.reinit(phi, nx, ny, nz, sweeps - 1, dx, h, tol=0.0, order=oracle_lib.GS_LEX)
            res["kind"] = "port"
        res["seconds"] = time.perf_counter() - t0

    threading.stack_size(1 << 30)  # reinit keeps two automatic arrays on the stack (subs.f90:724)
    th = threading.Thread(target=run)
    th.start()
    th.join()
    res["cells"] = (nx - 1) * (ny - 1) * (nz - 1) * sweeps
    with open(out_path, "w") as fh:
        json.dump(res, fh)


def cpu_baseline(n: int, slab: int = 48, sweeps: int = 8):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "cpu.json")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", out, "--size", str(n), "--cpu-slab", str(slab),
               "--cpu-sweeps", str(sweeps)]
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=900)
            r = json.load(open(out))
        except Exception as e:  # noqa: BLE001
            return {"value": None, "unit": "cell-updates/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
    what = ("reference subs.f90 reinit built by amdflang -O3 -fdefault-real-8 (oracle/_ref)" if r["kind"] == "reference"
            else "C restatement oracle/lsf_oracle.c (bit-identical to the reference)")
    return {
        "value": r["cells"] / r["seconds"],
        "unit": "cell-updates/s",
        "cores": 1,
        "kind": r["kind"],
        "sample": f"{n}x{n}x{slab}-point z-slab through the centre of the same {n}^3 phi0, {sweeps} sweeps "
                  f"(one cycle of the 8 raster directions), {r['seconds']:.1f} s; {what}; serial like the reference",
    }


# ------------------------------------------------------------------------------------------------
def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--mode", choices=("gs", "jacobi"), default="gs")
    ap.add_argument("--arith", choices=("fast", "strict"), default="fast")
    ap.add_argument("--size", type=int, default=512, help="points per axis (per GPU)")
    ap.add_argument("--global", dest="global_size", type=int, default=0,
                    help="N > 1, mode jacobi: points per axis of the FIXED global grid (strong scaling); 0 = weak scaling, "
                         "every rank owns a --size^3 block")
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64",
                    help="f32 = single-precision Jacobi path (BASELINE configuration 5); implies --mode jacobi")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary-ordering measurement")
    ap.add_argument("--no-sizes", action="store_true", help="skip the 256^3 / 1024^3 entries of a 512^3 run")
    ap.add_argument("--no-decomposed", action="store_true",
                    help="N > 1, mode gs: skip the block-decomposed Jacobi sweep (RCCL halo exchange) that is otherwise "
                         "timed after the headline measurement and attached as \"decomposed\"")
    ap.add_argument("--force-decomposed", action="store_true", help="run that measurement at N = 1 too (test aid)")
    ap.add_argument("--decomposed-timeout", type=float, default=480.0,
                    help="seconds the decomposed measurement may take before the headline line is printed without it")
    ap.add_argument("--cpu-worker", default=None)
    ap.add_argument("--multi-worker", default=None,
                    help="internal: measure the one-process driver with the RCCL transport on --gpus devices and write the "
                         "entries to this file (a child process of rank 0: a transport that has never run on more than one "
                         "device where it was written must not be able to take the job with it)")
    ap.add_argument("--slab-worker", default=None,
                    help="internal: measure the reference's ordering over z slabs (lsf_reinit_multi, LSF_ORDER_GS) on 1, 2, 4 ... "
                         "--gpus devices and write the entries to this file (a child process, for the same reason)")
    ap.add_argument("--multi-grid", type=int, default=512)
    ap.add_argument("--cpu-slab", type=int, default=48)
    ap.add_argument("--cpu-sweeps", type=int, default=8)
    args = ap.parse_args()
    if args.cpu_worker:
        _cpu_worker(args.cpu_worker, args.size, args.cpu_slab, args.cpu_sweeps)
        return
    if args.multi_worker:
        from levelsetfortran_amd import _lib

        ent = _single_process_entries(_lib.load(), args.gpus, args.multi_grid, args.steps, args.warmup, args.arith, transports=("rccl",))
        with open(args.multi_worker, "w") as fh:
            json.dump(ent, fh)
        return
    if args.slab_worker:
        ent = _slab_entries(args.gpus, args.multi_grid, args.steps, args.warmup, args.arith)
        with open(args.slab_worker, "w") as fh:
            json.dump(ent, fh)
        return
    f32 = args.dtype == "f32"
    if f32:
        args.mode, args.no_secondary, args.no_cpu_baseline = "jacobi", True, True
    bytes_per_cell = 12.0 if f32 else BYTES_PER_CELL_UPDATE  # read phi, read phiS, write phi

    import numpy as np
    import torch
    import torch.distributed as dist

    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import _lib, fields

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # test aid (LSF_BENCH_SHARED_GPU=1): all ranks share GPU 0 and talk over gloo, so that the N > 1 control flow of this
    # file can be rehearsed on a one-GPU box (RCCL refuses two ranks on one device); never a measurement
    shared_gpu = os.environ.get("LSF_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            if shared_gpu:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize(dev)

    lib = _lib.load()
    N = args.size
    K, W = args.steps, args.warmup
    out = {}

    scaling = "weak"
    res = None
    if args.mode == "jacobi" and world > 1:
        from levelsetfortran_amd import distributed as lsd

        if args.global_size:
            G = args.global_size
            res = lsd.bench_decomposed((G, G, G), K, W, dev, arith=args.arith, dtype=args.dtype, shared_gpu=shared_gpu)
            scaling = "strong"
        else:
            res = lsd.bench_weak_scaling(N, K, W, dev, arith=args.arith, dtype=args.dtype, shared_gpu=shared_gpu)
        cells_total, seconds, prof, parallelism = res["cells_total"], res["seconds"], res["prof"], res["parallelism"]
        order = "jacobi"
        # the headline's own evidence (outside its timed region): same ranks, transport and decomposition on a grid that can be
        # compared as a whole with rank 0's single-domain sweep
        Gp = 64 if min(res["global_grid"]) < 256 else 256
        headline_parity = lsd.parity_decomposed((Gp, Gp, Gp), 8, dev, arith=args.arith, dtype=args.dtype, shared_gpu=shared_gpu, dims=res["dims"])
    else:
        order = args.mode
        nx = ny = nz = N - 1
        if N > 640 or world > 1:  # numpy temporaries (several fields per rank) do not belong in host memory: build it in HBM
            phi0, dx = fields.two_sphere_phi0_device((N, N, N), dev)
        else:
            phi0_np, dx = fields.two_sphere_phi0((N, N, N))
            phi0 = torch.from_numpy(phi0_np.reshape(-1, order="F")).to(dev)
            del phi0_np
        h = fields.reinit_step(dx)
        if f32:
            phi0 = phi0.to(torch.float32)
        phiS = phi0.clone()
        phi = phi0.clone()

        def run(order_, sweeps, profile=False, arith=None):
            lib.lsf_profile(1 if profile else 0)
            rep = lsf.reinit(phi, None, None, nx, ny, nz, sweeps - 1, dx, h, tol=0.0, order=order_, arith=arith or args.arith,
                             phiS=phiS)
            assert rep.count == sweeps, (rep.count, sweeps)
            return rep

        def timed(order_, arith=None, K_=None, W_=None, samples=1):
            """`samples` = 1: the headline (exactly K steps after W, once).  Secondary entries take two samples and report the faster
            one with both wall times listed (no secondary number is a single sample)."""
            K_, W_ = K_ or K, W if W_ is None else W_
            best, walls = None, []
            for _ in range(samples):
                phi.copy_(phi0)
                if W_ > 0:
                    run(order_, W_, arith=arith)
                barrier()
                t0 = time.perf_counter()
                run(order_, K_, profile=True, arith=arith)
                barrier()
                dt = time.perf_counter() - t0
                sw, bc, fin = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
                nl, ns = ctypes.c_longlong(), ctypes.c_int()
                lib.lsf_profile_get(ctypes.byref(sw), ctypes.byref(bc), ctypes.byref(fin), ctypes.byref(nl), ctypes.byref(ns))
                lib.lsf_profile(0)
                walls.append(dt / K_ * 1e3)
                pr = {"sweep_ms": sw.value, "bc_ms": bc.value, "finish_ms": fin.value, "launches": nl.value,
                      "sweeps": ns.value, "kernel": (lib.lsf_profile_kernel() or b"").decode()}
                if best is None or dt < best[0]:
                    best = (dt, pr)
            if samples > 1:
                best[1]["ms_per_step_samples"] = walls
            return best

        seconds, prof = timed(order)
        cells_total = float(nx - 1) * (ny - 1) * (nz - 1) * K * world
        parallelism = "1 GPU" if world == 1 else f"{world} independent replicas (exact Gauss-Seidel ordering does not shard)"

    if world > 1:
        t = torch.tensor([seconds], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seconds = float(t.item())
    # N > 1, exact ordering: the headline is N replicas.  The path that really shards -- the block-decomposed Jacobi
    # sweep with its halo exchange -- is measured afterwards in the same job and attached as "decomposed".  It runs
    # under a watchdog: whatever happens in there (an exception on one rank, a stuck collective), rank 0 still prints
    # the headline line first; the job then ends with a non-zero exit status.
    decomposed = None
    state = {"emitted": False}
    watchdog = None
    if (world > 1 or args.force_decomposed) and args.mode == "gs" and not args.no_decomposed:
        import threading

        def _bail():
            if rank == 0 and not state["emitted"] and state.get("line") is not None:
                line = dict(state["line"])
                line["decomposed"] = {"entries": state.get("entries", []), "error": f"not finished within {args.decomposed_timeout} s"}
                print(json.dumps(line), flush=True)
            os._exit(3)

        watchdog = threading.Timer(args.decomposed_timeout, _bail)
        watchdog.daemon = True
    state["run_decomposed"] = watchdog is not None

    src_now = kernel_sources_fingerprint()

    # the roofline "also against a measured device-copy bandwidth" -- a 1 GiB streaming copy (16 bytes per lane
    # and access) timed in THIS run on THIS device, bytes read + bytes written per second of the fastest of 5 passes
    peak_measured = None
    try:
        g_ = ctypes.c_double(0.0)
        if lib.lsf_copy_bandwidth(1 << 30, 5, ctypes.byref(g_)) == 0 and g_.value > 0:
            peak_measured = g_.value
    except Exception:  # noqa: BLE001
        peak_measured = None

    def roofline(prof_, cells_per_sweep, size=None):
        if not prof_ or not prof_.get("sweeps"):
            return None
        kernel = prof_["kernel"]  # the exact ordering picks box or skewed tiles by grid size
        per_sweep_s = prof_["sweep_ms"] * 1e-3 / prof_["sweeps"]
        ach = cells_per_sweep * bytes_per_cell / per_sweep_s / 1e9
        lps = prof_["launches"] / prof_["sweeps"]
        # measured HBM bytes and issue counters of THIS kernel instance at THIS size (profiles/ships.sh -> profiles/traffic.json),
        # attached only while the kernel sources are the ones the counters were collected on (the entry's stamp)
        traffic, issue, stale = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                ent = json.load(open(tpath)).get(kernel, {}).get(str(size or N))
                if isinstance(ent, dict) and ent.get("kernel_sources") != src_now:
                    stale = {"counters_collected_on_sources": ent.get("kernel_sources"), "sources_now": src_now, "file": ent.get("source")}
                    ent = None
                if isinstance(ent, dict):
                    traffic = ent.get("hbm_bytes_per_sweep")
                    if "valu_lane_insts_per_cell" in ent:
                        issue = {"valu_lane_insts_per_cell": ent["valu_lane_insts_per_cell"],
                                 "valu_active_over_wave_cycles": ent.get("valu_active_over_wave_cycles"),
                                 "valu_busy_share": ent.get("valu_busy_share"), "source": ent.get("source"),
                                 "note": "SQ_INSTS_VALU x 64 / cell updates: what the kernel ISSUES per cell (the cheapest form of the "
                                         "arithmetic found needs 259 fp64 operations, 313 in the in-place ordering); the busy share "
                                         "says how full the vector unit is -- of these instructions, redundant ones included"}
                elif ent:
                    traffic = ent
            except Exception:  # noqa: BLE001
                traffic = None
        # what the memory system really moved: counter bytes per sweep / the kernel's time in THIS run (fabric requests of the L2s;
        # Infinity-Cache hits are counted, MI355X_MICROARCH.md) -- against the vendor peak and against the copy measured above
        hbm_meas = traffic / per_sweep_s / 1e9 if traffic else None
        return {
            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "peak_measured": peak_measured, "frac_of_measured": (ach / peak_measured) if peak_measured else None,
            "hbm_measured_GBps": hbm_meas, "hbm_measured_frac_of_peak": (hbm_meas / HBM_PEAK_GBS) if hbm_meas else None,
            "hbm_measured_frac_of_peak_measured": (hbm_meas / peak_measured) if hbm_meas and peak_measured else None,
            "traffic": traffic / lps if traffic else None, "kernel": kernel, "launches_per_sweep": lps,
            "avg_launch_us": per_sweep_s / lps * 1e6,
            "algorithmic_bytes_per_launch": cells_per_sweep * bytes_per_cell / lps,
            "traffic_per_sweep": traffic, "algorithmic_bytes_per_sweep": cells_per_sweep * bytes_per_cell, "issue": issue,
            **({"counters_stale": stale} if stale else {}),
            "note": f"achieved = {bytes_per_cell:.0f} B x (N-2)^3 cells / HIP-event time of the sweep kernel launch(es) of one sweep; "
                    "traffic = measured HBM bytes per launch (per sweep / launches per sweep) from the rocprofv3 PMC "
                    "passes summarised in profiles/, null if not collected for this size; peak_measured = a 1 GiB streaming device "
                    "copy timed in this run (read + written bytes per second), frac_of_measured = achieved / that; hbm_measured_GBps "
                    "= traffic per sweep / this run's kernel time per sweep (what the memory system moved, re-reads included)",
        }

    per_gpu_cells_per_sweep = cells_total / K / world
    out = {
        "metric": f"cell-updates/s (WENO5 reinit, {N}^3 {'fp32' if f32 else 'fp64'})",
        "value": cells_total / seconds,
        "unit": "cell-updates/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": seconds / K * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": (f"WENO5 HJ reinit sweep (weno+Godunov+Euler, BC, RMS), global grid {res['global_grid']} {'fp32' if f32 else 'fp64'} "
                         f"split {res['dims']}, local block {res['local_block']} points, synthetic two-sphere phi0 (SURVEY.md 8d), "
                         f"HBM-resident") if res else
                        (f"WENO5 HJ reinit sweep (weno+Godunov+Euler, BC, RMS), {N}^3 {'fp32' if f32 else 'fp64'} per GPU, synthetic "
                         f"two-sphere phi0 (SURVEY.md 8d), HBM-resident"),
            "grid": res["global_grid"] if res else [N, N, N],
            "ordering": "exact Gauss-Seidel raster order of the reference (tiled hyperplane wavefront)" if order == "gs"
                        else "Jacobi (double-buffered; not reference-equal)",
            "arithmetic": args.arith,
            "parallelism": parallelism,
        },
        "roofline": roofline(prof, per_gpu_cells_per_sweep) or _job_roofline(cells_total / seconds, world, bytes_per_cell),
    }
    if world > 1:
        out["rccl_ranks"] = _rccl_ranks(dist)  # world size as the RCCL backend reports it
    if res is not None:
        out["parity"] = headline_parity
    if not f32 and args.arith == "fast":
        # how long the FAST arithmetic of `value` stays within north_star's 1e-10 RMS of the reference's field, per BASELINE
        # configuration (profiles/micro/fast_valid.py on the GPU; STRICT holds it for any number of sweeps)
        import glob

        fv = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fast_valid.json")))
        if fv:
            try:
                d_ = json.load(open(fv[-1]))
                out["fast_valid_sweeps"] = {
                    "source": os.path.relpath(fv[-1], ROOT), "tolerance_rms": d_.get("tolerance"),
                    **{k: {"within_tolerance_through_sweep": v["fast_within_1e-10_rms_through_sweep"],
                           "first_checkpoint_outside": v["first_checkpoint_outside"], "sweeps_run": v["sweeps_run"],
                           "stopped_at": v["stopped_at"]}
                       for k, v in d_.items() if isinstance(v, dict)}}
            except Exception as e:  # noqa: BLE001
                out["fast_valid_sweeps"] = {"error": repr(e)}
    if not f32 and args.arith == "fast" and order == "gs" and world == 1 and not args.no_secondary:
        # ... and measured in THIS run, for THIS field: the sweeps just timed (warm-up + steps, FAST) marched again in the
        # reference's own arithmetic from the same phi0, outside the timed region; RMS of the difference
        fast_field = phi.clone()
        phi.copy_(phi0)
        if W > 0:
            run("gs", W, arith="strict")
        run("gs", K, arith="strict")
        dlt = fast_field - phi
        rms_fs = float(torch.sqrt(torch.mean(dlt * dlt)).item())
        out.setdefault("fast_valid_sweeps", {})["this_run"] = {
            "sweeps": W + K, "rms_fast_minus_strict": rms_fs, "max_abs": float(dlt.abs().max().item()), "within_1e-10_rms": rms_fs <= 1.0e-10,
            "signs_equal": bool(torch.equal(fast_field < 0, phi < 0)),
            "note": "measured in this run: the field `value` was timed on, after its warm-up + steps sweeps, against the same sweeps in "
                    "LSF_ARITH_STRICT (the reference's bits); the per-configuration rows above are the committed table"}
        del fast_field, dlt
    if prof:
        out["step_breakdown_ms"] = {k: prof[k] / max(prof["sweeps"], 1) for k in ("sweep_ms", "bc_ms", "finish_ms")}
    out["roofline_fp32_valu" if f32 else "roofline_fp64_valu"] = {
        "bound": "fp32-valu (packed)" if f32 else "fp64-valu", "peak": FP32_VALU_PEAK_TFLOPS if f32 else FP64_VALU_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "note": ("the update is ~505 flop/cell as written in subs.f90 (SURVEY.md 8d): at 12 B/cell the vector unit, not "
                 "HBM, is the first bound; achieved = 505 x value; the peak needs every instruction to be a packed FMA")
                if f32 else
                ("the update is ~505 fp64 flop/cell as written in subs.f90 (SURVEY.md 8d): at 24 B/cell the fp64 vector "
                 "unit, not HBM, is the first bound; achieved = 505 x value"),
        "achieved": 505.0 * (cells_total / seconds) / world / 1e12,
    }
    if not f32:
        # the 505 are the reference's operations as written; what has to be computed is less, what is issued is more: see
        # roofline.issue (measured lane-instructions per cell) for the third number
        ops = 313.0 if order == "gs" else 259.0
        out["roofline_fp64_valu"]["useful_ops_per_cell"] = ops
        out["roofline_fp64_valu"]["achieved_useful_Tops"] = ops * (cells_total / seconds) / world / 1e12
        out["roofline_fp64_valu"]["note_useful"] = ("cheapest form of the arithmetic found: 259 fp64 operations per cell (FMA = 1), 313 in the in-place ordering "
                                                    "where the two sides of an interface cannot be shared (DESIGN.md 4.1, 4.2); against 39.3 T instructions/s")
    vkey = "roofline_fp32_valu" if f32 else "roofline_fp64_valu"
    out[vkey]["frac"] = out[vkey]["achieved"] / out[vkey]["peak"]

    if world == 1 and not args.no_secondary and not (args.mode == "jacobi" and world > 1):
        other = "jacobi" if order == "gs" else "gs"
        sec2, prof2 = timed(other, samples=2)
        cells2 = float(nx - 1) * (ny - 1) * (nz - 1) * K
        out[other] = {
            "value": cells2 / sec2, "unit": "cell-updates/s", "ms_per_step": sec2 / K * 1e3,
            "ms_per_step_samples": prof2.get("ms_per_step_samples"),
            "roofline": roofline(prof2, cells2 / K),
            "note": "Jacobi ordering: same per-cell arithmetic, double-buffered; its field differs from the reference's "
                    "Gauss-Seidel result (5.5e-5 RMS on cube40, SURVEY.md section 0)" if other == "jacobi"
                    else "exact Gauss-Seidel ordering of the reference",
        }

    if world == 1 and not args.no_secondary and not f32 and args.arith == "fast":
        # the same ordering(s) in the reference's own arithmetic (LSF_ARITH_STRICT: bit-identical results, the drop-in's
        # default): a shorter run, it is 2 x slower
        # (the exact ordering with the command's own sweep counts -- it is the number that carries the bit-for-bit guarantee;
        # the Jacobi ordering in a shorter run)
        st = {}
        for order_, KS, WS in (("gs", K, W), ("jacobi", min(K, 16), min(W, 8))):
            secs, profs = timed(order_, arith="strict", K_=KS, W_=WS, samples=2)
            cells_s = float(nx - 1) * (ny - 1) * (nz - 1) * KS
            st[order_] = {"value": cells_s / secs, "unit": "cell-updates/s", "ms_per_step": secs / KS * 1e3, "steps": KS,
                          "warmup": WS, "ms_per_step_samples": profs.get("ms_per_step_samples"), "roofline": roofline(profs, cells_s / KS)}
        st["note"] = ("every operation as subs.f90 writes it (no contraction, IEEE division and square root): the field is the "
                      "reference's bit for bit after any number of sweeps; `value` above is the FAST arithmetic (same "
                      "mathematics, ~1e-16 per sweep away: see fast_valid_sweeps for how long that stays inside 1e-10 RMS)")
        out["strict_arithmetic"] = st
        if order == "gs":
            # the number that carries the contract's parity on every BASELINE configuration (the drop-in's arithmetic), where it cannot
            # be overlooked
            out["value_strict"], out["ms_per_step_strict"] = st["gs"]["value"], st["gs"]["ms_per_step"]
            out["roofline_strict_frac"] = (st["gs"]["roofline"] or {}).get("frac")

    if world == 1 and not args.no_secondary and not f32 and order == "gs" and N == 512 and not args.no_sizes:
        # north_star's other two sizes, in the driver's own line: the exact ordering, 16 sweeps after 16, FAST and STRICT
        def at_size(G):
            p0, dxg = fields.two_sphere_phi0_device((G, G, G), dev)
            hg = fields.reinit_step(dxg)
            pS, p = p0.clone(), p0.clone()
            ent = {}
            for ar in ("fast", "strict"):
                best, walls = None, []
                for _ in range(2):  # two samples, the faster one reported, both listed
                    p.copy_(p0)
                    lib.lsf_profile(0)
                    lsf.reinit(p, None, None, G - 1, G - 1, G - 1, 15, dxg, hg, tol=0.0, order="gs", arith=ar, phiS=pS)
                    barrier()
                    t0 = time.perf_counter()
                    lib.lsf_profile(1)
                    rep = lsf.reinit(p, None, None, G - 1, G - 1, G - 1, 15, dxg, hg, tol=0.0, order="gs", arith=ar, phiS=pS)
                    barrier()
                    dt = time.perf_counter() - t0
                    assert rep.count == 16
                    sw, bc, fin = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
                    nl, ns = ctypes.c_longlong(), ctypes.c_int()
                    lib.lsf_profile_get(ctypes.byref(sw), ctypes.byref(bc), ctypes.byref(fin), ctypes.byref(nl), ctypes.byref(ns))
                    lib.lsf_profile(0)
                    pr = {"sweep_ms": sw.value, "bc_ms": bc.value, "finish_ms": fin.value, "launches": nl.value, "sweeps": ns.value,
                          "kernel": (lib.lsf_profile_kernel() or b"").decode()}
                    walls.append(dt / 16 * 1e3)
                    if best is None or dt < best[0]:
                        best = (dt, pr)
                dt, pr = best
                cells = float(G - 2) ** 3 * 16
                ent[ar] = {"value": cells / dt, "unit": "cell-updates/s", "ms_per_step": dt / 16 * 1e3, "steps": 16, "warmup": 16,
                           "ms_per_step_samples": walls, "roofline": roofline(pr, cells / 16, size=G)}
            del p0, pS, p
            torch.cuda.empty_cache()
            return ent

        out["sizes"] = {"note": "exact Gauss-Seidel ordering at north_star's other sizes, same run, same box: 16 sweeps after 16, FAST "
                                "and STRICT arithmetic, wall clock between barrier + synchronize pairs, the faster of two samples "
                                "(ms_per_step_samples lists both); `value` stays the 512^3 figure"}
        for G in (256, 1024):
            try:
                out["sizes"][f"{G}^3"] = at_size(G)
            except Exception as e:  # noqa: BLE001
                out["sizes"][f"{G}^3"] = {"error": repr(e)[:200]}
        lib.lsf_release_workspace()

    if world == 1 and not args.no_secondary:
        # min/max-flow sweep (set3d.f90:394-462) on the same grid: 16 B per grid point per iteration
        # (SURVEY.md 8d); input = exact two-sphere distance so that the narrow band is a thin shell
        del phi, phi0, phiS
        state["freed"] = True
        x, y, z, dxm = fields.grid_axes((N, N, N))
        dmin = None
        for c in ((-0.6, 0.0, 0.0), (0.6, 0.0, 0.0)):
            r = np.sqrt((x[:, None, None] - c[0]) ** 2 + (y[None, :, None] - c[1]) ** 2 + (z[None, None, :] - c[2]) ** 2) - 0.5
            dmin = r if dmin is None else np.minimum(dmin, r)
        sdf = torch.from_numpy(np.asfortranarray(dmin).reshape(-1, order="F")).to(dev)
        del dmin, r
        mm = {}
        KM, KM0 = 50, 10
        for order_ in ("gs", "jacobi"):
            times = {}
            for km in (KM0, KM):
                f = sdf.clone()
                nb = torch.zeros(f.numel(), dtype=torch.int32, device=dev)
                sb = torch.zeros_like(nb)
                lsf.narrowBand(nx, ny, nz, dxm, f, nb, sb)
                band = float((nb == 1).sum().item()) / float(nb.numel())
                lsf.minmaxFlow(f, nb, sb, nx, ny, nz, 2, dxm, 0.1 * h, tol=0.0, order=order_)
                barrier()
                t0 = time.perf_counter()
                lsf.minmaxFlow(f, nb, sb, nx, ny, nz, km, dxm, 0.1 * h, tol=0.0, order=order_)
                barrier()
                times[km] = time.perf_counter() - t0
                del f, nb, sb
            dtm = times[KM]
            slope = (times[KM] - times[KM0]) / (KM - KM0)
            # Bytes of ONE iteration on the band, from the band kernels' own arguments (csrc/lsf_minmax_band.hpp, compulsory bytes per
            # list cell and launch; on entry the list IS the band):
            #   start pass  k_minmax_band<0>   A 8 r, L 4 r, nb 6 x 4 r, isband 1 w, curv 8 w, down 3 x 8 w, X 8 w            = 77
            #   fix visit   k_minmax_band_fix  isband 1 r, nb 24 r, A 8 r, curv 8 r, down 24 r, L 4 r, X 8 r (exact ordering)  = 77
            #   RMS pass    k_minmax_band<2>   A 8 r, isband 1 r, X 8 r                                                         = 17
            # (neighbour gathers re-read A / X: no new bytes; further fix passes touch stamped chunks only and are not counted:
            # a lower bound of what moves, so the fraction printed cannot be flattered by re-reads)
            per_cell = (77.0 + 77.0 + 17.0) if order_ == "gs" else (77.0 + 17.0)
            list_cells = band * float(N) ** 3
            slope_pos = max(slope, 1e-12)
            band_gbps = list_cells * per_cell / slope_pos / 1e9
            mm[order_] = {"ms_per_iteration": dtm / KM * 1e3, "iterations_per_call": KM, "points_per_s": float(N) ** 3 * KM / dtm,
                          "ms_per_iteration_10": times[KM0] / KM0 * 1e3,
                          "band_cells": list_cells, "band_bytes_per_cell_and_iteration": per_cell,
                          "band_GBps": band_gbps, "frac_of_hbm_peak": band_gbps / HBM_PEAK_GBS,
                          "frac_of_peak_measured": (band_gbps / peak_measured) if peak_measured else None,
                          "dense_equivalent_GBps": 16.0 * float(N) ** 3 * KM / dtm / 1e9,
                          "ms_per_additional_iteration": slope * 1e3, "ms_per_call_overhead": (times[KM0] - KM0 * slope) * 1e3,
                          "band_fraction": band}
        mm["note"] = ("min/max-flow iteration at the same size (set3d.f90:394-462), narrow band = |phi| < 4.1 dx of an exact two-sphere "
                      "distance; gs = the reference's (+,+,+) raster order reproduced exactly, jacobi = double-buffered.  Default executor: "
                      "on the band only (list of band cells built once per call, compact arrays, the field written once at the end): the "
                      "cost follows band_fraction.  band_GBps = band_cells x band_bytes_per_cell_and_iteration (compulsory bytes of the "
                      "start pass, ONE fix visit -- exact ordering -- and the RMS pass, from the kernels' own arguments) / "
                      "ms_per_additional_iteration; frac_of_hbm_peak is that against 8 TB/s (a few small launches and grid barriers per "
                      "iteration: latency, not bandwidth, sets the time).  dense_equivalent_GBps = 16 B per GRID point per iteration, the "
                      "accounting of SURVEY.md 8d for an executor that streams the whole grid: a rate of work done, not of bytes moved "
                      "(no fraction is quoted for it).  ms_per_iteration = one call of iterations_per_call iterations, list build and "
                      "the final narrowBand pass included (ms_per_call_overhead), / iterations_per_call; ms_per_iteration_10 = the "
                      "same for a 10-iteration call (the figure of rounds 1-4)")
        out["minmax"] = mm

    failed = False
    if state["run_decomposed"]:
        from levelsetfortran_amd import distributed as lsd

        state["line"] = out
        state["entries"] = entries = []
        watchdog.start()
        try:
            if world > 1 and not state.get("freed"):
                # one GPU of this job, the single-domain Jacobi sweep of one --size^3 block: what the weak-scaling headline divides by
                secj, profj = timed("jacobi")
                tj = torch.tensor([secj], device=dev, dtype=torch.float64)
                dist.all_reduce(tj, op=dist.ReduceOp.MAX)
                cj = float(nx - 1) * (ny - 1) * (nz - 1) * K
                state["same_path_one_gpu"] = {"value": cj / float(tj.item()), "unit": "cell-updates/s", "ms_per_step": float(tj.item()) / K * 1e3,
                                              "note": "single-domain Jacobi sweep on one GPU (slowest rank), same steps / warm-up"}
            if not state.get("freed"):
                del phi, phi0, phiS
                state["freed"] = True
            torch.cuda.empty_cache()
            lib.lsf_release_workspace()

            def one(gpts, kind, note):
                r = lsd.bench_decomposed(gpts, K, W, dev, arith=args.arith, shared_gpu=shared_gpu)
                t = torch.tensor([r["seconds"]], device=dev, dtype=torch.float64)
                if world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sec = float(t.item())
                entries.append({"path": "one process per GPU, torch.distributed (RCCL)", "ordering": "jacobi", "dtype": "f64",
                                "scaling": kind, "global_grid": r["global_grid"], "dims": r["dims"], "local_block": r["local_block"],
                                "n_gpus": world, "value": r["cells_total"] / sec, "unit": "cell-updates/s",
                                "ms_per_step": sec / K * 1e3, "roofline": _job_roofline(r["cells_total"] / sec, world, BYTES_PER_CELL_UPDATE),
                                "rccl_ranks": _rccl_ranks(dist), "transport": (dist.get_backend() if dist.is_initialized() else "none"),
                                "parity": state.get("parity_td"), "note": note})
                torch.cuda.empty_cache()

            dims = lsd.default_dims(world)
            # evidence before numbers: the decomposed sweep of THIS job (same ranks, same transport, same decomposition) against rank 0's
            # single-domain sweep, on a grid small enough to compare as a whole; attached to every entry of the path (collective call)
            Gp = 64 if N < 128 else 256
            par = lsd.parity_decomposed((Gp, Gp, Gp), 8, dev, arith=args.arith, shared_gpu=shared_gpu, dims=dims,
                                        sabotage=os.environ.get("LSF_BENCH_PARITY_SABOTAGE") == "1")
            state["parity_td"] = par
            torch.cuda.empty_cache()
            one(tuple(d * N for d in dims), "weak", f"every rank owns a {N}^3-point block")
            for G in ((96,) if N < 128 else (256, 512, 1024)):  # north_star: fixed 256^3, 512^3, 1024^3 at 1, 2, 4, 8 GPUs
                if all(G // d >= 12 for d in dims):
                    one((G, G, G), "strong", "fixed global grid: compare with the same entry of the runs at other N")
            # the one-process C-ABI driver (include/lsf.h: lsf_multi_*): rank 0 drives 1, 2, 4, ... N devices, the other
            # ranks wait.  Needs every GPU of the job visible to rank 0 (torch.distributed.run does not hide them).
            if world > 1:
                sp = []
                # the other ranks wait on the HOST (a gloo group): an RCCL barrier would keep a polling kernel on the very
                # devices rank 0 is about to measure
                host_group = None if shared_gpu else dist.new_group(backend="gloo")
                torch.cuda.synchronize(dev)
                dist.barrier(group=host_group)
                if rank == 0 and not shared_gpu and torch.cuda.device_count() >= world:
                    try:
                        # appended to `entries` one by one (the peer-copy entries first): an RCCL entry that hangs on a node this code
                        # has never seen takes only itself into the watchdog's error record
                        Gs = 64 if N < 128 else 512
                        _single_process_entries(lib, world, Gs, K, W, args.arith, transports=("peer",), sink=entries)
                        entries.extend(_rccl_entries_in_a_child(world, Gs, K, W, args.arith))
                        # the ordering that IS reference-equal, sharded: z slabs of the exact Gauss-Seidel tile graph
                        entries.extend(_slab_entries_in_a_child(world, Gs, K, W, args.arith))
                    except Exception as e:  # noqa: BLE001
                        entries.append({"path": "one process, lsf_multi", "value": None, "error": repr(e)[:300]})
                        failed = True
                dist.barrier(group=host_group)
            decomposed = {"entries": entries,
                          "note": "Jacobi ordering (not reference-equal; bit-identical to the single-GPU Jacobi sweep), whole-job "
                                  "aggregates; same K sweeps after W warm-up sweeps, barrier + synchronize on both sides, max over "
                                  "ranks.  The driver's scaling efficiency over `value` describes replicas; halo-path scaling is the "
                                  "ratio of equal-grid \"strong\" entries across runs (or across n_gpus inside the lsf_multi block)."}
        except Exception as e:  # noqa: BLE001
            decomposed = {"entries": entries, "error": repr(e)[:300]}
            failed = True
    if decomposed is not None:
        # an entry whose in-run parity record says "not the single-domain field" makes the job fail -- after the line is printed
        bad = [e.get("path", "?") for e in decomposed.get("entries", []) if isinstance(e.get("parity"), dict) and not e["parity"].get("ok")]
        if bad:
            decomposed["parity_failed"] = bad
            failed = True
        out["decomposed"] = decomposed
        # N > 1: the headline is a path that COMMUNICATES -- the weak-scaling block-decomposed Jacobi sweep over RCCL (every rank a
        # --size^3 block, 3-cell halos per sweep) -- not N replicas, which scale linearly by construction and say nothing about the
        # node.  The replicas of the exact ordering stay in the line as "replicas_gs"; "same_path_one_gpu" is
        # the single-domain Jacobi sweep of one --size^3 block on one GPU of this job: efficiency = value / (N x that).
        weak = next((e for e in decomposed.get("entries", []) if e.get("scaling") == "weak" and e.get("ordering") == "jacobi"
                     and e.get("value") and "torch.distributed" in e.get("path", "")), None)
        if world > 1 and weak is not None:
            out["replicas_gs"] = {"value": out["value"], "unit": "cell-updates/s", "ms_per_step": out["ms_per_step"], "roofline": out["roofline"],
                                  "scaling": "weak", "parallelism": out["config"]["parallelism"],
                                  "note": "N independent replicas of the exact Gauss-Seidel ordering: linear by construction, no communication"}
            out["metric"] = (f"cell-updates/s (WENO5 reinit, {N}^3 fp64 per GPU); N > 1: block-decomposed Jacobi ordering, 3-cell halos "
                             "over RCCL, weak scaling (N = 1: the reference's exact ordering on one GPU)")
            # what was computed from the replicas' timing goes with the replicas; the VALU roofline of the headline is recomputed
            # from the headline's own value with the Jacobi ordering's operation count
            for k_ in ("step_breakdown_ms", "roofline_fp64_valu", "fast_valid_sweeps"):
                if k_ in out:
                    out["replicas_gs"][k_] = out.pop(k_)
            out["value"], out["ms_per_step"], out["roofline"] = weak["value"], weak["ms_per_step"], weak["roofline"]
            per_gpu = weak["value"] / world
            out["roofline_fp64_valu"] = {"bound": "fp64-valu", "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "achieved": 505.0 * per_gpu / 1e12, "frac": 505.0 * per_gpu / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                         "useful_ops_per_cell": 259.0, "achieved_useful_Tops": 259.0 * per_gpu / 1e12,
                                         "note": "per GPU, from the headline's own value: 505 fp64 flop per cell as written in subs.f90; the "
                                                 "Jacobi ordering needs 259 operations per cell in the cheapest form found (DESIGN.md 4.2)"}
            out["headline_path"] = "decomposed-jacobi-weak"
            out["parity"] = weak.get("parity")
            out["config"]["workload"] = (f"WENO5 HJ reinit sweep (weno+Godunov+Euler, BC, RMS), global grid {weak['global_grid']} fp64 split "
                                         f"{weak['dims']}, local block {weak['local_block']} points, synthetic two-sphere phi0 (SURVEY.md 8d), HBM-resident")
            out["config"]["grid"] = weak["global_grid"]
            out["config"]["ordering"] = "Jacobi (double-buffered; not reference-equal), block-decomposed, halos by torch.distributed (RCCL)"
            out["config"]["parallelism"] = f"{world} ranks, decomposition {weak['dims']}, one process per GPU"
            out["rccl_ranks"] = weak.get("rccl_ranks")
            if state.get("same_path_one_gpu"):
                out["same_path_one_gpu"] = state["same_path_one_gpu"]
        elif world > 1:
            out["headline_path"] = "replicas (no weak decomposed entry: see decomposed.error / entries)"
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N)
        print(json.dumps(out), flush=True)
        state["emitted"] = True
    if world > 1:
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0 and isinstance(out.get("parity"), dict) and not out["parity"].get("ok"):
        failed = True  # the headline's own parity record (mode jacobi, N > 1)
    if failed:
        sys.exit(3)


def _rccl_entries_in_a_child(world, G, K, W, arith, timeout=150.0):
    """the one-process driver with the RCCL transport (include/lsf.h: LSF_TRANSPORT_RCCL) on `world` devices, measured by a child
    process with a time limit: whatever happens in there ends up as an entry (a number, or an error), never as a hung job"""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rccl.json")
        cmd = [sys.executable, os.path.abspath(__file__), "--multi-worker", out, "--gpus", str(world), "--multi-grid", str(G),
               "--steps", str(K), "--warmup", str(W), "--arith", arith]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                 "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=timeout, env=env)
            return json.load(open(out))
        except Exception as e:  # noqa: BLE001
            return [{"path": "one process, lsf_multi (C ABI), halos by RCCL ncclSend / ncclRecv", "n_gpus": world, "transport": "rccl",
                     "value": None, "error": repr(e)[:300]}]


def _slab_entries_in_a_child(world, G, K, W, arith, timeout=120.0):
    """the reference's ordering over z slabs on 1, 2, 4 ... `world` devices, measured by a child process with a time limit (peer
    stores and cross-device flags have never run between two devices where this code was written)"""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "slabs.json")
        cmd = [sys.executable, os.path.abspath(__file__), "--slab-worker", out, "--gpus", str(world), "--multi-grid", str(G),
               "--steps", str(K), "--warmup", str(W), "--arith", arith]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                 "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=timeout, env=env)
            return json.load(open(out))
        except Exception as e:  # noqa: BLE001
            return [{"path": SLAB_PATH, "ordering": "gs", "n_gpus": world, "value": None, "error": repr(e)[:300]}]


SLAB_PATH = "one process, lsf_reinit_multi with LSF_ORDER_GS (C ABI): one dataflow launch per z slab, cut planes by peer stores"


def _slab_entries(world, G, K, W, arith, devices_of=None):
    """the reference's in-place ordering over z slabs (include/lsf.h: lsf_reinit_multi, LSF_ORDER_GS) on 1, 2, 4 ... `world`
    devices, fixed G^3 grid: `value` from the device time of the slabs' launches (lsf_slabs_info; the call also uploads, transposes
    and downloads the field: `call_s`).  devices_of(nd) -> device list (default 0 .. nd - 1; a one-GPU test names device 0 nd times)."""
    import numpy as np

    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import _lib, fields

    lib = _lib.load()
    out = []
    counts, nd = [], 1
    while nd <= world:
        counts.append(nd)
        nd *= 2
    if counts[-1] != world:
        counts.append(world)
    n = G - 1
    phi0, dx = fields.two_sphere_phi0((G, G, G))
    h = fields.reinit_step(dx)
    first = None
    for nd in counts:
        devs = list(devices_of(nd)) if devices_of else list(range(nd))
        try:
            a = phi0.copy(order="F")
            lsf.reinit_multi(a, n, n, n, max(min(W, 8), 1) - 1, dx, h, devs, tol=0.0, arith=arith, order="gs")  # warm-up: code, tables
            a = phi0.copy(order="F")
            t0 = time.perf_counter()
            rep = lsf.reinit_multi(a, n, n, n, K - 1, dx, h, devs, tol=0.0, arith=arith, order="gs")
            call_s = time.perf_counter() - t0
            assert rep.count == K, (rep.count, K)
            v = [ctypes.c_int(0) for _ in range(4)]
            ks = ctypes.c_double(0)
            _lib.check(lib.lsf_slabs_info(*[ctypes.byref(x) for x in v], ctypes.byref(ks)))
            val = float(n - 1) ** 3 * K / ks.value
            if first is None:
                first = a
            out.append({"path": SLAB_PATH, "ordering": "gs", "arith": arith, "dtype": "f64", "scaling": "strong", "global_grid": [G, G, G],
                        "n_gpus": nd, "devices": devs, "value": val, "unit": "cell-updates/s", "ms_per_step": ks.value / K * 1e3,
                        "roofline": _job_roofline(val, len(set(devs)), BYTES_PER_CELL_UPDATE), "call_s": call_s,
                        "blocks_per_slab": v[1].value, "finegrained": bool(v[2].value),
                        "equal_to_first_entry": bool(np.array_equal(a, first)),
                        "parity": _parity_one_process(devs, 64 if G < 128 else 256, 16, arith, order="gs"),
                        "note": "reference-equal ordering (bit-identical to lsf_reinit on one device); value = cell-updates of the K "
                                "sweeps / the longest of the slabs' launches (device events)"})
        except Exception as e:  # noqa: BLE001
            out.append({"path": SLAB_PATH, "ordering": "gs", "n_gpus": nd, "value": None, "error": repr(e)[:300]})
    return out


_PARITY_SINGLE = {}  # (G, sweeps, arith, order) -> (field, rms) of the single-device run the one-process entries are compared with


def _parity_one_process(devs, G, sweeps, arith, order="jacobi", transport="peer"):
    """In-run evidence for an entry of the one-process drivers (lsf_reinit_multi: blocks of the Jacobi ordering, z slabs of the exact
    ordering): `sweeps` sweeps of a G^3 two-sphere field on `devs` against the same sweeps by lsf_reinit on ONE device (the call site
    both stand in for: set3d.f90:308) -- the field by SHA-256, the RMS trace element by element (exact ordering: equal; Jacobi
    blocks: block sums added in rank order against one fixed-order sum, relative 1e-11).  Outside every timed region."""
    import hashlib

    import numpy as np

    import levelsetfortran_amd as lsf
    from levelsetfortran_amd import fields

    try:
        n = G - 1
        key = (G, sweeps, arith, order)
        if key not in _PARITY_SINGLE:
            phi0, dx = fields.two_sphere_phi0((G, G, G))
            a = phi0.copy(order="F")
            r1 = lsf.reinit(a, None, None, n, n, n, sweeps - 1, dx, fields.reinit_step(dx), tol=0.0, order=order, arith=arith)
            _PARITY_SINGLE.clear()  # one 134 MB field at a time
            _PARITY_SINGLE[key] = (a, list(r1.rms), phi0, dx)
        want, rms1, phi0, dx = _PARITY_SINGLE[key]
        got = phi0.copy(order="F")
        kw = {} if order == "gs" else {"transport": transport}
        r = lsf.reinit_multi(got, n, n, n, sweeps - 1, dx, fields.reinit_step(dx), list(devs), tol=0.0, arith=arith, order=order, **kw)
        sha = lambda f: hashlib.sha256(f.reshape(-1, order="F").tobytes()).hexdigest()  # noqa: E731
        field_ok = sha(got) == sha(want)
        rtol = 0.0 if order == "gs" else 1.0e-11
        worst = max((abs(x - y) / abs(y) if y else abs(x - y)) for x, y in zip(r.rms, rms1)) if r.rms else 0.0
        trace_ok = r.count == sweeps and len(r.rms) == len(rms1) and worst <= rtol
        return {"ok": bool(field_ok and trace_ok), "field_sha_equal": bool(field_ok), "rms_trace_equal": bool(trace_ok),
                "rms_trace_max_rel_diff": worst, "rms_trace_rtol": rtol, "grid": [G, G, G], "sweeps": sweeps, "devices": list(devs),
                "arith": arith, "ordering": order, "max_abs_field_diff": float(np.abs(got - want).max()),
                "against": "lsf_reinit on one device, same field and sweeps, outside the timed region"}
    except Exception as e:  # noqa: BLE001
        return {"ok": False, "error": repr(e)[:300]}


def _rccl_ranks(dist):
    """ranks of the job's RCCL communicator (torch.distributed backend "nccl" is RCCL on ROCm); 0 under gloo or without a group"""
    return dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else 0


def _job_roofline(cells_per_s, n_gpus, bytes_per_cell):
    """north_star: every multi-GPU number "as absolute numbers and as fraction of the HBM roofline" -- the job's algorithmic
    bytes per second against the sum of the HBM peaks of the GPUs it ran on (wall clock, halo exchange included)"""
    ach = cells_per_s * bytes_per_cell / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS * n_gpus, "unit": "GB/s", "frac": ach / (HBM_PEAK_GBS * n_gpus),
            "note": f"{bytes_per_cell:.0f} B x cell-updates/s of the whole job / ({n_gpus} x {HBM_PEAK_GBS:.0f} GB/s); wall clock of the "
                    "K sweeps, not a kernel time"}


def _single_process_entries(lib, world, G, K, W, arith, transports=("peer",), sink=None):
    """rank 0 only: the block-decomposed sweep through lsf_multi_* on 1, 2, 4, ... `world` devices, fixed G^3 grid, once per
    transport ("peer": peer copies; "rccl": ncclSend / ncclRecv behind the C ABI, needs a device per block)."""
    import numpy as np
    import torch

    from levelsetfortran_amd import _lib, fields

    out = sink if sink is not None else []  # (a caller's list: what is measured survives a watchdog bail-out of a later entry)
    nd = 1
    counts = []
    while nd <= world:
        counts.append(nd)
        nd *= 2
    if counts[-1] != world:
        counts.append(world)
    n = G - 1
    mode = _lib.LSF_ORDER_JACOBI | (_lib.LSF_ARITH_STRICT if arith == "strict" else _lib.LSF_ARITH_FAST)

    def measure(nd, tp):
        if tp == "rccl" and (nd == 1 or nd != counts[-1]):
            return  # one RCCL entry per job, on all of its devices (a single device has no neighbour, no message)
        devs = (ctypes.c_int * nd)(*range(nd))
        M = ctypes.c_void_p()
        _lib.check(lib.lsf_multi_create(n, n, n, devs, nd, None, 0, ctypes.byref(M)))
        try:
            _lib.check(lib.lsf_multi_configure(M, 8, _lib.LSF_TRANSPORT_RCCL if tp == "rccl" else _lib.LSF_TRANSPORT_PEER))
            dx = h = None
            for r in range(nd):
                g0, ext = (ctypes.c_int * 3)(), (ctypes.c_int * 3)()
                dv = ctypes.c_int(0)
                _lib.check(lib.lsf_multi_block(M, r, g0, ext, None, None, ctypes.byref(dv)))
                d = torch.device("cuda", dv.value)
                rng = tuple((int(a), int(a) + int(e)) for a, e in zip(g0, ext))
                blk, dx = fields.two_sphere_phi0_device((G, G, G), d, ranges=rng)
                torch.cuda.synchronize(d)
                _lib.check(lib.lsf_multi_upload_block(M, r, blk.data_ptr()))
                del blk
            h = fields.reinit_step(dx)
            done = ctypes.c_int(0)
            if W > 0:
                _lib.check(lib.lsf_multi_run(M, W - 1, dx, h, 0.0, mode, ctypes.byref(done), None, 0))
            t0 = time.perf_counter()
            _lib.check(lib.lsf_multi_run(M, K - 1, dx, h, 0.0, mode, ctypes.byref(done), None, 0))
            sec = time.perf_counter() - t0
            assert done.value == K, (done.value, K)
            rr, he, hc = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
            _lib.check(lib.lsf_multi_info(M, None, None, ctypes.byref(rr), None, ctypes.byref(he), ctypes.byref(hc), None, None))
            val = float(n - 1) ** 3 * K / sec
            out.append({"path": "one process, lsf_multi (C ABI): thread + 2 streams per device, halos by "
                                + ("RCCL ncclSend / ncclRecv" if tp == "rccl" else "peer copies"), "ordering": "jacobi",
                        "dtype": "f64", "scaling": "strong", "global_grid": [G, G, G], "n_gpus": nd, "transport": tp,
                        "rccl_ranks": rr.value, "value": val, "unit": "cell-updates/s", "ms_per_step": sec / K * 1e3,
                        "roofline": _job_roofline(val, nd, BYTES_PER_CELL_UPDATE),
                        "host_enqueue_ms_per_step": he.value / K * 1e3, "host_calls_ms_per_step": hc.value / K * 1e3,
                        "parity": None,  # filled in below, after the blocks of the timed run have been released
                        "note": "the call returns after the last sweep has finished on every device (the run includes its own "
                                "thread start-up and final synchronisation)"})
        finally:
            _lib.check(lib.lsf_multi_destroy(M))
        out[-1]["parity"] = _parity_one_process(list(range(nd)), 64 if G < 128 else 256, 8, arith, order="jacobi", transport=tp)

    for nd, tp in [(c, t) for t in transports for c in counts]:
        try:
            measure(nd, tp)
        except Exception as e:  # noqa: BLE001
            if tp != "rccl":
                raise
            # the RCCL transport has never run on more than one device where this code was written: report, do not fail the job
            out.append({"path": "one process, lsf_multi (C ABI), halos by RCCL ncclSend / ncclRecv", "n_gpus": nd, "transport": tp,
                        "value": None, "error": repr(e)[:300]})
    return out


if __name__ == "__main__":
    main()
