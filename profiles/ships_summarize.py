#!/usr/bin/env python3
"""Condense the PMC passes of profiles/ships.sh: per configuration the sweep kernel's INSTANCE (template arguments as rocprofv3
lists them), HBM bytes per sweep (FETCH_SIZE x 2: gfx950 counts a wide read's 128-byte request as 64 bytes,
MI355X_MICROARCH.md "HBM"; WRITE_SIZE as it is; both in KiB), and the issue counters: vector lane-instructions per cell update
(SQ_INSTS_VALU x 64 / cell updates) and the share of a wavefront's cycles its vector instructions are active.
  profiles/<tag>_ships.json   everything
  profiles/traffic.json       {instance: {N: {...}}}: what bench.py attaches to `roofline`
"""
import csv
import glob
import json
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
from bench import kernel_sources_fingerprint  # noqa: E402

SRC = kernel_sources_fingerprint()  # bench.py attaches the counters to a line only while the kernel sources are these


def norm(name):
    """'void lsf::k_reinit_gs_persist<16, 2, 2, 5, false>(lsf::GsArgs)' -> 'k_reinit_gs_persist<16,2,2,5,false>'"""
    m = re.search(r"(k_\w+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")).replace(" ", "") if m else name


def counters(d, want):
    tot, names = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = norm(r["Kernel_Name"])
            if want not in k:
                continue
            t = tot.setdefault(k, {})
            t[r["Counter_Name"]] = t.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            names[k] = names.get(k, 0) + 1
    return tot


def occupancy(d, want):
    """waves per SIMD the kernel runs at, from its dispatch record: registers (allocated in granules of 8, 512 per SIMD lane), LDS
    (160 KiB per CU) and workgroup size"""
    occ = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = norm(r["Kernel_Name"])
            if want not in k or k in occ:
                continue
            regs = (int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]) + 7) // 8 * 8
            wg_waves = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]) // 64)
            by_regs = min(8, 512 // max(regs, 8))
            lds = int(r["LDS_Block_Size"])
            by_lds = (163840 // lds) * wg_waves / 4.0 if lds else 8.0
            occ[k] = {"vgprs": regs, "lds_bytes": lds, "waves_per_simd": min(float(by_regs), by_lds, 8.0)}
    return occ


res, traffic = {}, {}
tp = os.path.join(here, "traffic.json")
if os.path.exists(tp):
    try:
        old = json.load(open(tp))
        traffic = {k: v for k, v in old.items() if any(isinstance(x, dict) for x in v.values())}  # drop the old name-only entries
    except Exception:  # noqa: BLE001
        traffic = {}
for line in open(os.path.join(out, "configs.txt")):
    N, arith, mode, sweeps = line.split()
    N, sweeps = int(N), int(sweeps)
    d = os.path.join(out, f"{N}_{arith}_{mode}")
    want = "k_reinit_gs_" if mode == "gs" else "k_reinit_jacobi"
    cells = float(N - 2) ** 3 * sweeps
    ent = {"size": N, "arithmetic": arith, "ordering": mode, "sweeps_counted": sweeps}
    fetch, write, sq = counters(os.path.join(d, "FETCH_SIZE"), want), counters(os.path.join(d, "WRITE_SIZE"), want), counters(os.path.join(d, "SQ"), want)
    for k in sorted(set(fetch) | set(write) | set(sq)):
        e = dict(ent)
        fb = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 / sweeps
        wb = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0 / sweeps
        e["fetch_bytes_per_sweep"], e["write_bytes_per_sweep"] = fb, wb
        e["hbm_bytes_per_sweep"] = fb + wb
        e["algorithmic_bytes_per_sweep"] = 24.0 * float(N - 2) ** 3
        e["traffic_over_algorithmic"] = (fb + wb) / e["algorithmic_bytes_per_sweep"]
        # the same from the L2s' request counters resolved by size (rocprofv3 on gfx950: TCC_EA0_RDREQ with _32B / _64B / _128B,
        # TCC_EA0_WRREQ with _64B): bytes really asked of the fabric; FETCH_SIZE x 2 above is exact only where every request is
        # a 128-byte one
        rq = counters(os.path.join(d, "RDREQ"), want).get(k, {})
        wq = counters(os.path.join(d, "WRREQ"), want).get(k, {})
        if rq.get("TCC_EA0_RDREQ_sum"):
            n_all, n32, n64, n128 = (rq.get(c, 0.0) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
            other = max(n_all - n32 - n64 - n128, 0.0)
            e["read_requests_per_sweep"] = {"all": n_all / sweeps, "32B": n32 / sweeps, "64B": n64 / sweeps, "128B": n128 / sweeps}
            e["fetch_bytes_per_sweep_by_request_size"] = (32.0 * n32 + 64.0 * n64 + 128.0 * n128 + 64.0 * other) / sweeps
        if wq.get("TCC_EA0_WRREQ_sum"):
            w_all, w64 = wq.get("TCC_EA0_WRREQ_sum", 0.0), wq.get("TCC_EA0_WRREQ_64B_sum", 0.0)
            e["write_requests_per_sweep"] = {"all": w_all / sweeps, "64B": w64 / sweeps}
            e["write_bytes_per_sweep_by_request_size"] = (64.0 * w64 + 32.0 * max(w_all - w64, 0.0)) / sweeps
            e["dram_requests_per_sweep"] = {"read": wq.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / sweeps, "write": wq.get("TCC_EA0_WRREQ_DRAM_sum", 0.0) / sweeps}
        if "fetch_bytes_per_sweep_by_request_size" in e and "write_bytes_per_sweep_by_request_size" in e:
            e["hbm_bytes_per_sweep_upper_bound"] = e["hbm_bytes_per_sweep"]
            e["hbm_bytes_per_sweep"] = e["fetch_bytes_per_sweep_by_request_size"] + e["write_bytes_per_sweep_by_request_size"]
            e["traffic_over_algorithmic_upper_bound"] = e["traffic_over_algorithmic"]
            e["traffic_over_algorithmic"] = e["hbm_bytes_per_sweep"] / e["algorithmic_bytes_per_sweep"]
            e["traffic_method"] = "TCC_EA0_RDREQ / _WRREQ by request size"
        s = sq.get(k, {})
        if s.get("SQ_WAVES"):
            e["sq"] = {c: v for c, v in s.items()}
            e["valu_lane_insts_per_cell"] = s.get("SQ_INSTS_VALU", 0.0) * 64.0 / cells
            e["valu_active_over_wave_cycles"] = s.get("SQ_ACTIVE_INST_VALU", 0.0) / max(s.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            e["wait_any_over_wave_cycles"] = s.get("SQ_WAIT_ANY", 0.0) / max(s.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            oc = occupancy(os.path.join(d, "SQ"), want).get(k)
            if oc and "k_reinit_gs_" in k: # (the tile kernels: LDS and their register bound fix the occupancy; the dispatch record of
                                           # the Jacobi kernels lists fewer registers than their waves-per-SIMD attribute allots)
                e.update(oc)
                # resident wavefronts per SIMD x the share of a wavefront's cycles in which it has a vector instruction in
                # flight: how full the SIMD's vector issue is (<= 1; the exact-ordering launch keeps its slots occupied)
                e["valu_busy_share"] = min(1.0, e["valu_active_over_wave_cycles"] * oc["waves_per_simd"])
        res[f"{N} {arith} {mode} {k}"] = e
        traffic.setdefault(k, {})[str(N)] = {q: e[q] for q in ("hbm_bytes_per_sweep", "traffic_over_algorithmic", "traffic_method", "traffic_over_algorithmic_upper_bound", "valu_lane_insts_per_cell",
                                                                 "valu_active_over_wave_cycles", "waves_per_simd", "valu_busy_share") if q in e}
        traffic[k][str(N)]["source"] = f"profiles/{tag}_ships.json"
        traffic[k][str(N)]["kernel_sources"] = SRC
json.dump(res, open(os.path.join(here, f"{tag}_ships.json"), "w"), indent=1)
json.dump(traffic, open(tp, "w"), indent=1)
for k, e in res.items():
    print(k, {q: (round(v, 3) if isinstance(v, float) else v) for q, v in e.items() if q != "sq"})
