#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs under gpurun_out/pmc_*: per-trade instruction counts of the pricing kernel."""
import collections, csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _identity
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
_b = _identity.from_log("gpurun_out/pmc_a.log")       # the build the counted runs loaded (bench.py's `build` object)
print(f"# sources {_b.get('source_sha256')}  lib {_b.get('lib_sha256')}  summarised at {_identity.git_head()}")
for f in sorted(glob.glob("gpurun_out/pmc_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "price_" in r["Kernel_Name"]:
            m = re.search(r"(price_\w+<[^>]*>)", r["Kernel_Name"])
            agg[((m.group(1) if m else r["Kernel_Name"])[-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{k:62s} {c:24s} {sum(v)/len(v):16.0f}  per trade {sum(v)/len(v)/n:10.1f}")
