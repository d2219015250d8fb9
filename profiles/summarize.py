#!/usr/bin/env python3
"""Condenses a profiles/run_profile.sh output directory into a text summary (per-kernel durations from the
kernel trace, per-kernel counter totals from the PMC passes)."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "*", "*_kernel_stats.csv")):
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:90]:90s} calls={r['Calls']:>4s} avg_ms={float(r['AverageNs']) / 1e6:10.4f} total_ms={float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):6.2f}%")
for d in ("sq1", "sq2", "tcc", "fetch", "write"):
    for f in glob.glob(os.path.join(out, d, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "rocclr" in k or "at::native" in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
        print(f"== PMC pass {d}: per-dispatch averages ==")
        for k, v in agg.items():
            n = len(calls[k])
            print(f"{k[:70]:70s} dispatches={n} " + " ".join(f"{a}={b / n:.4g}" for a, b in sorted(v.items())))
