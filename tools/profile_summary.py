#!/usr/bin/env python3
"""Turn gpurun_out/prof_TAG/{trace,fetch,write} into profiles/rNN_TAG_* summaries (kernel stats CSV + traffic JSON).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are collected in
separate passes, are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of a coalesced streaming read, so
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-byte-per-lane stores (the gamma rows)."""
import csv, glob, json, os, shutil, sys
tag, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r01")
base = f"gpurun_out/prof_{tag}"
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(f"{base}/trace/*/*kernel_stats.csv")
shutil.copy(stats, f"profiles/{rnd}_{tag}_kernel_stats.csv")
out = {"tag": tag}
for name in ("fetch", "write"):
    vals = {}
    for r in csv.DictReader(open(newest(f"{base}/{name}/*/*counter_collection.csv"))):
        if "price_" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in vals.items():
        out[k + "_KiB_per_launch"] = sum(v) / len(v)
for r in csv.DictReader(open(stats)):
    if "price_" in r["Name"] and "kernel" not in out:
        out["kernel"] = r["Name"].split("(")[0]
        out["avg_ns"] = float(r["AverageNs"]); out["calls"] = int(r["Calls"])
        out["min_ns"] = float(r["MinNs"]); out["max_ns"] = float(r["MaxNs"])
# the timed region alone: the last `steps` launches of the kernel in the trace (the all-launch average of the stats file
# also holds the ~90 warm-up launches, the first of them on a cold clock)
try:
    trace = newest(f"{base}/trace/*/*kernel_trace.csv")
    rows = [r for r in csv.DictReader(open(trace)) if "price_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps = 20
    try:
        import re as _re
        m_ = _re.search(r'"steps": ([0-9]+)', open(f"{base}.trace.log").read())
        if m_:
            steps = int(m_.group(1))
    except OSError:
        pass
    last = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[-steps:]]
    out["timed_region_launches"] = len(last)
    out["avg_ns_timed_region"] = sum(last) / len(last)
except (ValueError, OSError, KeyError):
    pass
# bench.py's own HIP-event kernel time inside the profiled (kernel-trace) run: must agree with rocprofv3's average
import re
try:
    m = re.search(r'"kernel_ms": ([0-9.]+)', open(f"{base}.trace.log").read())
    if m:
        out["bench_kernel_ms_in_profiled_run"] = float(m.group(1))
except OSError:
    pass
rd = 2.0 * out.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024
wr = out.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
out.update(read_bytes_corrected=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr,
           note="read = 2 x FETCH_SIZE x 1024 (gfx950 half-count correction), write = WRITE_SIZE x 1024")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _identity
_identity.stamp(out, f"{base}.trace.log")       # which build was profiled, which commit the summary is filed under
json.dump(out, open(f"profiles/{rnd}_{tag}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
