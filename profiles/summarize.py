#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of profiles/collect.sh into the small committed summaries:
  profiles/<tag>_kernel_stats.csv      per-kernel calls / total / average / min / max (from --stats)
  profiles/<tag>_pmc_hbm.json          HBM bytes per sweep from FETCH_SIZE / WRITE_SIZE, raw and with the
                                       gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B
                                       requests as 64 B for wide coalesced reads -> x2 upper bound)
  (profiles/traffic.json is written by profiles/ships_summarize.py: keyed by kernel instance and size)
"""
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
SWEEPS = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0  # sweeps covered by the PMC passes (warm-up + timed)
here = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    g = glob.glob(os.path.join(out_dir, pattern), recursive=True)
    return g[0] if g else None


for sub, suffix in (("trace", ""), ("trace_f32", "_f32"), ("trace_s20", "_steps20"), ("trace_mm", "_minmax")):
    stats = one(f"{sub}/**/*kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats)))
        with open(os.path.join(here, f"{tag}_kernel_stats{suffix}.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])

res = {"unit": "bytes per sweep", "N": 512, "note": "sum over the sweep-kernel dispatches of one sweep; counters are in KiB "
       "(rocprofv3 FETCH_SIZE/WRITE_SIZE); fetch_corrected = 2 x fetch_raw (gfx950 wide-read correction, upper bound)"}
traffic = {}
def gs_kernel_name():
    # the exact ordering launches k_reinit_gs_persist (dataflow, default), k_reinit_gs_skew or k_reinit_gs_box (slots)
    f = one("pmc_FETCH_SIZE_gs/**/*counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            if "k_reinit_gs_" in r["Kernel_Name"]:
                for name in ("k_reinit_gs_persist", "k_reinit_gs_skew", "k_reinit_gs_box"):
                    if name in r["Kernel_Name"]:
                        return name
    return "k_reinit_gs_box"


def jacobi_kernel_name(mode="jacobi"):
    f = one(f"pmc_FETCH_SIZE_{mode}/**/*counter_collection.csv")
    sh, plain = ("k_reinit_jacobi_f32_sh", "k_reinit_jacobi_f32") if mode == "f32" else ("k_reinit_jacobi_sh", "k_reinit_jacobi")
    if f:
        for r in csv.DictReader(open(f)):
            if sh in r["Kernel_Name"]:
                return sh
    return plain


for mode, kern in (("gs", gs_kernel_name()), ("jacobi", jacobi_kernel_name()), ("f32", jacobi_kernel_name("f32")), ("mm", "k_minmax_")):  # every kernel of a min/max iteration: band executor since round 5 (k_minmax_band*), dense k_minmax_fp<*> before
    entry = {}
    bytes_per_cell = 12.0 if mode == "f32" else 24.0
    sweeps = 17.0 if mode == "mm" else SWEEPS  # mm_time.py: 1 + 16 iterations
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = one(f"pmc_{ctr}_{mode}/**/*counter_collection.csv")
        if not f:
            continue
        tot, ndisp = 0.0, 0
        for r in csv.DictReader(open(f)):
            hit = kern in r["Kernel_Name"] and (mode == "f32" or "_f32" not in r["Kernel_Name"])
            if hit and r["Counter_Name"] == ctr:
                tot += float(r["Counter_Value"])
                ndisp += 1
        entry[ctr] = {"sum_KiB": tot, "dispatches": ndisp}
    if "FETCH_SIZE" in entry and "WRITE_SIZE" in entry:
        fr = entry["FETCH_SIZE"]["sum_KiB"] * 1024 / sweeps
        wr = entry["WRITE_SIZE"]["sum_KiB"] * 1024 / sweeps
        entry["per_sweep"] = {"fetch_raw": fr, "fetch_corrected": 2 * fr, "write": wr, "hbm_raw": fr + wr,
                              "hbm_corrected": 2 * fr + wr, "algorithmic": 16.0 * 512 ** 3 if mode == "mm" else bytes_per_cell * 510 ** 3, "sweeps_covered": sweeps}
        traffic[kern] = {"512": 2 * fr + wr}
    res[kern] = entry
# calibration of the counters on a kernel with a known byte count in the same access width (8 B per lane):
# k_narrowband reads n doubles and writes 2n int32 -> 8n bytes each way
cal = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = one(f"pmc_{ctr}_cal/**/*counter_collection.csv")
    if f:
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_narrowband" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        if vals:
            cal[ctr] = {"measured_bytes": vals[-1] * 1024, "true_bytes": 8.0 * 512 ** 3, "ratio": vals[-1] * 1024 / (8.0 * 512 ** 3)}
res["calibration_k_narrowband_512"] = cal
# the same for 4 B per lane: k_pack<float> over a 512^3 field moves 4n bytes each way
cal32 = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = one(f"pmc_{ctr}_cal/**/*counter_collection.csv")
    if f:
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_pack" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        if vals:
            cal32[ctr] = {"measured_bytes": vals[-1] * 1024, "true_bytes": 4.0 * 512 ** 3, "ratio": vals[-1] * 1024 / (4.0 * 512 ** 3)}
res["calibration_k_pack_f32_512"] = cal32
if "FETCH_SIZE" in cal:
    # use the measured ratio instead of the blanket x2 for the committed traffic numbers
    for kern in list(traffic):
        e = res[kern]["per_sweep"]
        cc = cal32 if ("_f32" in kern and "FETCH_SIZE" in cal32) else cal
        e["fetch_calibrated"] = e["fetch_raw"] / cc["FETCH_SIZE"]["ratio"]
        e["hbm_calibrated"] = e["fetch_calibrated"] + e["write"] / cc.get("WRITE_SIZE", {"ratio": 1.0})["ratio"]
        traffic[kern] = {"512": e["hbm_calibrated"]}
json.dump(res, open(os.path.join(here, f"{tag}_pmc_hbm.json"), "w"), indent=1)
# (profiles/traffic.json -- what bench.py attaches to `roofline` -- is written by profiles/ships_summarize.py since round 4: keyed by
# kernel INSTANCE and size; the per-name numbers above stay in <tag>_pmc_hbm.json)
for name in ("bench_under_rocprof.json", "bench_f32_under_rocprof.json", "bench_s20_under_rocprof.json"):
    b = os.path.join(out_dir, name)
    if os.path.exists(b):
        lines = [l for l in open(b) if l.startswith("{")]
        if lines:
            open(os.path.join(here, f"{tag}_{name}"), "w").write(lines[-1])
print(json.dumps(res)[:1500])
