#!/usr/bin/env python3
"""MFMA utilisation and effective shader clock per kernel family from hardware counters of the bench command:
one rocprofv3 pass with --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv.

    clock       = GRBM_GUI_ACTIVE / 8 / kernel duration        (the counter sums the 8 XCDs' graphics-active cycles)
    utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)          (busy SIMD-cycles over available ones;
                  SQ_VALU_MFMA_BUSY_CYCLES = 32 x the number of 32x32x16 MFMAs, summed over all SIMDs)
    so TFLOP/s = utilisation x clock / 2.4 GHz x 2516.6.  GRBM's clock is what rocm-smi shows (1.9 - 2.3 GHz under these kernels);
    s_memtime inside the same kernels ticks 0.7 - 0.8x as often (tools/gemm_timeline.py: 1.3 - 1.7 GHz) - the power management
    delivers fewer shader cycles than the PLL frequency, so utilisation in DELIVERED cycles is higher by that factor.

    python tools/mfma_util.py <pmc dir> --arch sdxl --batch 8 > profiles/<name>.json
"""
import argparse
import collections
import csv
import glob
import importlib.util
import json
import os

spec = importlib.util.spec_from_file_location("hbm_traffic_family", os.path.join(os.path.dirname(os.path.abspath(__file__)), "hbm_traffic.py"))
src = open(spec.origin).read().split("def collect")[0]          # family() and kernels_sha() without that tool's argument parser
ns = {"__file__": spec.origin}
exec(compile(src, spec.origin, "exec"), ns)
family, kernels_sha = ns["family"], ns["kernels_sha"]

ap = argparse.ArgumentParser()
ap.add_argument("pmc_dir")
ap.add_argument("--arch", required=True)
ap.add_argument("--batch", type=int, required=True)
a = ap.parse_args()
rows = collections.defaultdict(dict)                             # dispatch id -> {counter: value, start, end, kernel}
for f in glob.glob(a.pmc_dir + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = rows[(f, r["Dispatch_Id"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["k"], d["t0"], d["t1"] = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for d in rows.values():
    fam = family(d["k"])
    if not fam or "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    x = agg[fam]
    x[0] += 1; x[1] += d["SQ_VALU_MFMA_BUSY_CYCLES"]; x[2] += d["GRBM_GUI_ACTIVE"]; x[3] += (d["t1"] - d["t0"]) * 1e-9
out = {"arch": a.arch, "per_gpu_batch": a.batch, "kernels_sha": kernels_sha(),
       "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; clock = active / 8 XCDs / duration; utilisation = busy / (1024 SIMDs x active / 8)",
       "families": {}}
for fam, (n, busy, active, secs) in sorted(agg.items()):
    out["families"][fam] = {"launches_sampled": n, "mfma_busy_cycles_per_launch": round(busy / n), "gui_active_cycles_per_launch": round(active / n),
                            "mfma_utilisation": round(busy / (active / 8.0 * 1024.0), 4) if active else None,
                            "effective_clock_ghz": round(active / 8.0 / secs / 1e9, 3) if secs else None,
                            "tflops_from_counters": round(busy / 32.0 * 32768.0 / secs / 1e12, 1) if secs else None,
                            "avg_launch_us": round(secs / n * 1e6, 2)}
print(json.dumps(out, indent=1))
