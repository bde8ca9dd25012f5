#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs under gpurun_out/pmc_*: per-trade instruction counts of the pricing kernel."""
import collections, csv, glob, re, sys
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
for f in sorted(glob.glob("gpurun_out/pmc_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "price_" in r["Kernel_Name"]:
            m = re.search(r"(price_\w+<[^>]*>)", r["Kernel_Name"])
            agg[((m.group(1) if m else r["Kernel_Name"])[-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{k:62s} {c:24s} {sum(v)/len(v):16.0f}  per trade {sum(v)/len(v)/n:10.1f}")
