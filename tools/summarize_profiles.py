#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 csv output) into profiles/<tag>_*.{csv,json}.
  python tools/summarize_profiles.py <tag> [kernel-name substring, default dm_beam_]
With a trace-only directory (no pmc_* passes) the summary lists every kernel of the run (calls, average, total)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
MATCH = sys.argv[2] if len(sys.argv) > 2 else "dm_beam_"
src = "gpurun_out/prof_%s" % tag
os.makedirs("profiles", exist_ok=True)
out = {"tag": tag}
for f in glob.glob(src + "/trace/*kernel_stats.csv"):
    shutil.copy(f, "profiles/%s_kernel_stats.csv" % tag)
    best = None
    for r in csv.DictReader(open(f)):        # the dominant beam kernel: dm_beam_w_kernel<E, KQ> or dm_beam_kernel<E, KQ, SPLIT>
        if MATCH in r["Name"] and (best is None or float(r["Percentage"]) > float(best["Percentage"])):
            best = r
    out["kernels"] = [{"kernel": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
                       "pct": float(r["Percentage"])} for r in csv.DictReader(open(f)) if float(r["Percentage"]) >= 0.01][:60]
    if best is not None:
        r = best
        out["kernel_trace"] = {"kernel": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                               "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
pmc = {}
for d in sorted(glob.glob(src + "/pmc_*")):
    for f in glob.glob(d + "/*counter_collection.csv"):
        acc = collections.defaultdict(float)
        n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].split("(")[0].strip() == out.get("kernel_trace", {}).get("kernel", "dm_beam_kernel").split("(")[0].strip() or \
               ("kernel_trace" not in out and MATCH in r["Kernel_Name"]):
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
        for k in acc:
            pmc[k] = acc[k] / n[k]
out["pmc_per_launch"] = pmc
for f in glob.glob(src + "/bench_trace.log"):
    for line in open(f):
        if line.startswith("{"):
            try:
                out.setdefault("bench_lines_under_profiler", []).append(json.loads(line))
            except ValueError:
                pass
if "FETCH_SIZE" in pmc:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE is in KiB-like units of 64-B requests and reads exactly 1/2 of a
    # wide (16 B/lane) coalesced stream on gfx950 -> double the read side; WRITE_SIZE is uncalibrated.
    fetch = pmc["FETCH_SIZE"] * 1024.0
    write = pmc.get("WRITE_SIZE", 0.0) * 1024.0
    out["hbm_traffic_per_launch_bytes"] = {"fetch_raw": fetch, "fetch_corrected_x2": 2 * fetch, "write_raw": write,
                                           "total_corrected": 2 * fetch + write}
if "SQ_VALU_MFMA_BUSY_CYCLES" in pmc and "GRBM_GUI_ACTIVE" in pmc:
    simd_cycles = pmc["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0     # GRBM summed over 8 XCDs; 256 CUs x 4 SIMDs
    out["mfma_pipe_utilisation"] = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
json.dump(out, open("profiles/%s_summary.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
